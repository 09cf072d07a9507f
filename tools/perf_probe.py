"""Quick V-cycle timing probe (development tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dealii_multigrid_amd as m

def algorithmic_bytes(ndofs_per_level, k=3, word=8):
    # bytes per V-cycle = s * [ (10k+3) * sum_{l>=1} N_l + 2 * sum_{l<L} N_l ]
    N = ndofs_per_level
    return word * ((10 * k + 3) * sum(N[1:]) + 2 * sum(N[:-1]))

ctx = m.Context(0)
cases = [tuple(c.split(":")) for c in sys.argv[1:]] or [("hypercube", "7", "1"), ("quadrant", "6", "4")]
for geo, L, p in cases:
    L, p = int(L), int(p)
    t0 = time.time()
    h = m.Hierarchy(ctx, geo, L, p, "HMG-global", coarse_solver="amg", max_brick=int(os.environ.get("MGAMD_MAX_BRICK", "-1")))
    ctx.synchronize()
    t1 = time.time()
    n = h.n_dofs
    b, z = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    ms_e = h.mg.time_vcycles(z, b, 10, False)
    ms_g = h.mg.time_vcycles(z, b, 10, True)
    N = [d.n_dofs for d in h.dofs]
    by = algorithmic_bytes(N)
    print(f"{geo} L={L} p={p}: n_dofs={n} setup {t1-t0:.1f}s  eager {ms_e:.3f} ms  graph {ms_g:.3f} ms  "
          f"-> {n/ms_g*1e3:.3e} DoF/s, algorithmic {by/n:.1f} B/DoF -> {by/ms_g*1e-9:.3f} TB/s = {by/ms_g*1e-9/8.0*100:.1f}% of 8 TB/s", flush=True)
    # per-level x per-stage breakdown of the SAME cycle: HIP events on the stream, no host synchronisation
    reps = 5
    h.mg.stage_timing(True)
    for _ in range(reps):
        h.mg.vmult(z, b)
    ms = h.mg.stage_times() / reps
    h.mg.stage_timing(False)
    nl = len(N)
    print(f"   stage times [ms per cycle, HIP events, sum = {ms.sum():.3f}]   to_mg {ms[7].sum():.3f}  to_global {ms[8].sum():.3f}")
    print("   level  n_dofs      pre    resid   restr   coarse  prol    post")
    for lv in range(nl - 1, -1, -1):
        row = [ms[s_, lv] for s_ in (0, 1, 2, 3, 4, 6)]
        if sum(row) > 0:
            print(f"   {lv:3d} {N[lv]:10d} " + " ".join(f"{v:7.3f}" for v in row))
    it, res = m.solve_cg(h.fine_operator, h.mg, z, b, 1e-4)
    print(f"   CG iterations {it}, residual {res:.3e}", flush=True)
