"""Dry run of the sharded path AT THE BENCH SIZE on one GPU: n ranks as host threads over the in-process simulator communicator
(device-to-device copies instead of RCCL), default partition policy of bench.py (chunks >= 4 M DoFs, rank-group tier >= 1 M DoFs,
replicated below).  Timings mean nothing here (the ranks share one GPU and exchange around host barriers); what it shows before a
real multi-GPU run: setup time and memory per rank, the level layout, halo sizes, and that the solve takes the single-GPU iteration
count.   python3 tools/sim_full_scale.py [geometry nref degree n_ranks [mg_type]]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dealii_multigrid_amd as m

geo, L, p, n_ranks = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("quadrant", 8, 4, 8)
mg_type = sys.argv[5] if len(sys.argv) > 5 else "HMG-global"
os.environ["MGAMD_CHEB_KEY_INIT"] = "1"
sim = m.SimGroup(n_ranks)
out, err = [None] * n_ranks, [None] * n_ranks


def rank_main(rk):
    try:
        t0 = time.time()
        ctx = m.Context(0)
        h = m.DistributedHierarchy(ctx, sim.comm(rk), geo, L, p, coarse_solver="amg" if mg_type != "PMG" else "cg_with_chebyshev", mg_type=mg_type)
        ctx.synchronize()
        t1 = time.time()
        b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
        h.fine_operator.rhs(b)
        it, res = m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
        ctx.synchronize()
        info = h.dofs[-1].info
        out[rk] = dict(setup_s=t1 - t0, solve_s=time.time() - t1, it=it, res=res, n_dofs=h.n_dofs, n_local=h.n_local, layout=h.layout(),
                       level_dofs=h.global_level_dofs(ctx), peers=int(info.n_peers), halo_send=int(info.n_halo_send),
                       local_per_level=[d.n_dofs for d in h.dofs], fused=[t.n_fused_bricks() for t in h.transfers[1:]])
    except BaseException as e:  # noqa
        err[rk] = e


th = [threading.Thread(target=rank_main, args=(r,)) for r in range(n_ranks)]
t0 = time.time()
for t in th:
    t.start()
for t in th:
    t.join()
for e in err:
    if e is not None:
        raise e
print(f"{geo} L={L} p={p} {mg_type} on {n_ranks} simulated ranks: wall {time.time() - t0:.1f} s")
print("  level layout (pieces per level, coarse -> fine):", out[0]["layout"])
print("  global DoFs per level:", out[0]["level_dofs"])
for rk, o in enumerate(out):
    print(f"  rank {rk}: setup {o['setup_s']:.1f} s  solve {o['solve_s']:.2f} s  CG iterations {o['it']} (residual {o['res']:.3e})  "
          f"local DoFs {o['n_local']}  peers {o['peers']}  halo entries {o['halo_send']}  fused bricks {o['fused']}")
print("  local DoFs per level, rank 0:", out[0]["local_per_level"])
assert len({o["it"] for o in out}) == 1 and all(o["n_dofs"] == out[0]["n_dofs"] for o in out)
