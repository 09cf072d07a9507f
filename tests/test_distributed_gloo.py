"""world_size-2 test over torch.distributed (gloo, CPU): every process builds ITS rank's level tables and halo plan,
and the partial sums of the shared DoFs travel through real point-to-point messages (dist.send/recv) laid out by the
plan -- the same pack / exchange / combine protocol the GPU path runs over RCCL.  Checked against the global
right-hand side and the global DoF count."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    import dealii_multigrid_amd as m

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    geo, L, p = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])  # p: one level of a PMG hierarchy lives on the finest mesh too
    trias = m.create_geometric_coarsening_sequence(m.Triangulation(geo, L))
    part = m.Partition(trias, world)
    lvl = len(trias) - 1
    d = m.DoFs(trias[lvl], p, 0, part, lvl, rank)
    plan = d.halo_plan()
    info = d.info
    # global DoF count: every DoF owned exactly once
    n_owned = torch.tensor([float(info.n_interior + info.n_tail_owned + info.n_dirichlet_owned + info.n_hanging_owned)], dtype=torch.float64)
    dist.all_reduce(n_owned)
    full = m.DoFs(trias[lvl], p)
    assert int(n_owned.item()) == full.n_dofs, (n_owned, full.n_dofs)
    # partial right-hand side, then the halo exchange with real messages
    b = d.rhs_constant()
    tail = b[info.n_interior:info.n_interior + info.n_tail]
    send = torch.from_numpy(np.ascontiguousarray(tail[plan["pack_idx"]]))
    recv = torch.zeros_like(send)
    reqs = []
    for j, q in enumerate(plan["peers"]):
        lo, hi = int(plan["peer_offset"][j]), int(plan["peer_offset"][j + 1])
        reqs.append(dist.isend(send[lo:hi].clone(), int(q)))
        reqs.append(dist.irecv(recv[lo:hi], int(q)))
    for r in reqs:
        r.wait()
    rv = recv.numpy()
    new = []
    for i, ti in enumerate(plan["sh_tail"]):
        acc = 0.0
        for e in range(plan["sh_ptr"][i], plan["sh_ptr"][i + 1]):
            s = plan["sh_src"][e]
            acc += tail[ti] if s < 0 else rv[s]
        new.append(acc)
    tail[plan["sh_tail"]] = new
    ks = lambda keys: [tuple(int(v) for v in k) for k in keys]
    gref = dict(zip(ks(full.keys()), full.rhs_constant()))
    ref = np.array([gref[k] for k in ks(d.keys())])
    assert np.abs(b - ref).max() < 1e-15, np.abs(b - ref).max()
    assert len(plan["peers"]) == world - 1 and len(plan["pack_idx"]) > 0
    dist.barrier()
    if rank == 0:
        print("GLOO_OK", world, full.n_dofs, len(plan["pack_idx"]))
    dist.destroy_process_group()
    """
)


import pytest


@pytest.mark.parametrize("geo,L,p,port", [("quadrant", 4, 2, 29517), ("annulus", 5, 4, 29518), ("annulus", 5, 1, 29519)])
def test_halo_exchange_over_gloo_world_size_2(tmp_path, geo, L, p, port):
    """octant p=2 (HMG-global finest level) and the p=4 / p=1 levels of the PMG annulus hierarchy (BASELINE configs[4]): all
    levels of a PMG hierarchy share the finest mesh's partition"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script), ROOT, geo, str(L), str(p)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "GLOO_OK 2" in out.stdout


WORKER_TIERS = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    import dealii_multigrid_amd as m

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    geo, L, p, group = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    trias = m.create_geometric_coarsening_sequence(m.Triangulation(geo, L))
    nl = len(trias)
    part = m.Partition(trias, world, 2.0, min_root_cells=trias[nl - 1].n_cells, group=group, min_sub_root_cells=0)
    assert part.group == group and part.sub_root_level < part.root_level == nl - 1
    lvl = part.root_level - 1                      # a level of the rank-group tier: n_parts pieces, each on `group` ranks
    n_parts, my_part, my_index = world // group, rank // group, rank % group
    d = m.DoFs(trias[lvl], p, 0, part, lvl, rank)  # the tables of MY PART (the same on every member of my group)
    plan, info = d.halo_plan(), d.info
    # every DoF of the level owned by exactly one part: sum over ONE member per part = sum over all ranks / group
    n_owned = torch.tensor([float(info.n_interior + info.n_tail_owned + info.n_dirichlet_owned + info.n_hanging_owned)], dtype=torch.float64)
    dist.all_reduce(n_owned)
    full = m.DoFs(trias[lvl], p)
    assert n_owned.item() / group == full.n_dofs, (n_owned, group, full.n_dofs)
    # halo exchange between the parts with real messages: part q is reached through the member of its group at MY position
    # (SubsetComm::exchange, csrc/comm.hpp)
    b = d.rhs_constant()
    tail = b[info.n_interior:info.n_interior + info.n_tail]
    send = torch.from_numpy(np.ascontiguousarray(tail[plan["pack_idx"]]))
    recv = torch.zeros_like(send)
    reqs = []
    for j, q in enumerate(plan["peers"]):
        assert 0 <= q < n_parts and q != my_part
        lo, hi = int(plan["peer_offset"][j]), int(plan["peer_offset"][j + 1])
        peer_rank = int(q) * group + my_index
        reqs.append(dist.isend(send[lo:hi].clone(), peer_rank))
        reqs.append(dist.irecv(recv[lo:hi], peer_rank))
    for r in reqs:
        r.wait()
    rv = recv.numpy()
    new = []
    for i, ti in enumerate(plan["sh_tail"]):
        acc = 0.0
        for e in range(plan["sh_ptr"][i], plan["sh_ptr"][i + 1]):
            s = plan["sh_src"][e]
            acc += tail[ti] if s < 0 else rv[s]
        new.append(acc)
    tail[plan["sh_tail"]] = new
    ks = lambda keys: [tuple(int(v) for v in k) for k in keys]
    gref = dict(zip(ks(full.keys()), full.rhs_constant()))
    ref = np.array([gref[k] for k in ks(d.keys())])
    assert np.abs(b - ref).max() < 1e-15, np.abs(b - ref).max()
    # the members of a group end up with the same vector (they hold the same part and received from counterparts that do)
    mine = torch.from_numpy(b.copy())
    sizes = torch.tensor([mine.numel()], dtype=torch.int64)
    all_sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    assert all(int(all_sizes[my_part * group + k].item()) == mine.numel() for k in range(group))
    # group sum by recursive doubling (SubsetComm::replica_sum): every member ends with the sum of the members' vectors
    v = torch.full((8,), float(rank + 1), dtype=torch.float64)
    bit = 1
    while bit < group:
        partner = rank ^ bit
        other = torch.zeros_like(v)
        ops = [dist.P2POp(dist.isend, v.clone(), partner), dist.P2POp(dist.irecv, other, partner)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        v = v + other
        bit <<= 1
    assert float(v[0]) == sum(my_part * group + k + 1 for k in range(group))
    dist.barrier()
    if rank == 0:
        print("GLOO_TIERS_OK", world, group, full.n_dofs, len(plan["pack_idx"]))
    dist.destroy_process_group()
    """
)


@pytest.mark.parametrize("geo,L,p,group,port", [("quadrant", 5, 2, 2, 29527), ("annulus", 6, 1, 2, 29528)])
def test_rank_group_tier_over_gloo_world_size_4(tmp_path, geo, L, p, group, port):
    """Partition tiers on 4 CPU processes: a level cut into world / group parts, each part's tables identical on the members of its
    group, the halo exchange between parts through the group member at the same position, the group sum by recursive doubling --
    the protocol of SubsetComm (csrc/comm.hpp) with real point-to-point messages"""
    script = tmp_path / "worker_tiers.py"
    script.write_text(WORKER_TIERS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script), ROOT, geo, str(L), str(p), str(group)], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "GLOO_TIERS_OK 4" in out.stdout
