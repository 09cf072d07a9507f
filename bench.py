#!/usr/bin/env python3
"""bench.py -- DoF/s per V-cycle of the global-coarsening multigrid preconditioner (BASELINE.json metric).

A "step" is one application of PreconditionMG::vmult (copy_to_mg + V-cycle + copy_from_mg,
ref:multigrid_throughput.cc:1132-1133) on the finest level of the workload, with the right-hand side
already resident in HBM.  Workloads (config.workload):

  octant_p4 (default)  BASELINE.json configs[2]: 3D octant (GeometryType "quadrant"), global coarsening, p = 4, FP64,
                       SmootherDegree 3 -- the configuration the metric is quoted on.  At N = 1 the line also carries, under
                       "also", the octant p = 1 number north_star asks for and p = 1 on a uniform mesh that fills the GPU
                       (BASELINE.json configs[1]).
  pmg_annulus          BASELINE.json configs[4]: polynomial global coarsening p = 4 -> 2 -> 1 on the annulus; the p = 1
                       coarse level is solved by CG preconditioned with its Chebyshev smoother (ref:multigrid_throughput.cc:922-944).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU.  The SAME global problem is sharded by spatial domain
   decomposition: every rank owns a Morton chunk of the octree, shared DoFs are exchanged with grouped RCCL
   send/recv over xGMI, the replicated coarse levels take one RCCL all-reduce -> "scaling": "strong",
   value = global n_dofs / max-over-ranks time; see DESIGN.md section 7.  `--mode replicas` runs N independent copies.
   A sharded run that fails on any rank ends the job with a non-zero exit code: there is no in-process fallback.)

Prints ONE JSON line on rank 0.  Synthetic data (f == 1, zero Dirichlet), no dataset.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
WORKLOADS = {
    # name: (geometry, NRefGlobal default, degree, Type, coarse solver, BASELINE.json config)
    "octant_p4": ("quadrant", 8, 4, "HMG-global", "amg", "configs[2]"),
    # (coarse solver "amg" with CoarseSolverNCycles 2: the reference's defaults, ref:scripts/default.json:11,14 -- here the library's
    # own smoothed-aggregation AMG)
    "pmg_annulus": ("annulus", 8, 4, "PMG", "amg", "configs[4]"),
}


def algorithmic_bytes_per_vcycle(n_dofs_per_level, k=3, word=8):
    """s * [ (10k+3) * sum_{l>=1} N_l + 2 * sum_{l<L} N_l ]  (the reference algorithm's compulsory
    vector traffic with x_1 of the zero-start smoother stored; the implementation recomputes x_1 = D^-1 b / theta inside
    the first two operator passes and moves 5 words per DoF and level fewer).  Level 0 belongs to the
    coarse solver (PMG: the p = 1 level), as in SURVEY's table."""
    N = n_dofs_per_level
    return word * ((10 * k + 3) * sum(N[1:]) + 2 * sum(N[:-1]))


def run_workload(m, ctx, geometry, n_ref, degree, mg_type, coarse, steps, warmup, barrier, sync, profile, comm=None, details=False,
                 number_type=None, diagnostics=False, coarse_cycles=1, subset_group=None):
    t0 = time.time()
    number_type = m.F64 if number_type is None else number_type
    word = 8 if number_type == m.F64 else 4
    if comm is None:
        h = m.Hierarchy(ctx, geometry, n_ref, degree, mg_type, smoother_degree=3, coarse_solver=coarse, number_type=number_type,
                        coarse_n_cycles=coarse_cycles)
    else:
        h = m.DistributedHierarchy(ctx, comm, geometry, n_ref, degree, mg_type=mg_type, smoother_degree=3, coarse_solver=coarse,
                                   number_type=number_type, coarse_n_cycles=coarse_cycles, subset_group=subset_group)
    # PreconditionMG::vmult acts on the OUTER vectors, which are double whatever MGNumberType is (ref:multigrid_throughput.cc:
    # 2430-2433 run<3, 1, double, MGNumber>): with float levels copy_to_mg / copy_from_mg cast
    if number_type == m.F64:
        b, z = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
        h.fine_operator.rhs(b)
    else:
        import numpy as np

        b, z = m.Vector(ctx, h.dofs[-1].n_dofs), m.Vector(ctx, h.dofs[-1].n_dofs)
        b.from_host(np.asarray(h.dofs[-1].rhs_constant()))
    ctx.synchronize()
    setup_s = time.time() - t0
    diag = rccl_diagnostics(m, ctx, comm, h) if (diagnostics and comm is not None) else None
    for _ in range(max(warmup, 1)):
        h.mg.vmult(z, b)
    if profile:
        # the dominant kernel symbol = the lattice kernel of the brick size with the most work on the finest level;
        # all of its launches (on every level that has such bricks) are timed, like rocprofv3's per-symbol average
        brick = max(h.dofs[-1].groups(), key=lambda g: g[1] * (degree * g[0] + 1) ** 3)[0]
        ctx.kernel_profile(True, brick)
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.mg.vmult(z, b)
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = None
    if profile:
        prof = ctx.kernel_profile_read() + (ctx.kernel_profile_bytes_moved(),)
        ctx.kernel_profile(False)
    if comm is None:
        N = [d.n_dofs for d in h.dofs]
    else:  # global level sizes: owned DoFs summed over the ranks (replicated levels are complete on every rank)
        N = h.global_level_dofs(ctx)
    res = dict(n_dofs=h.n_dofs, n_cells=h.trias[-1].n_cells, n_levels=len(N), level_dofs=N, elapsed=elapsed, setup_s=setup_s,
               bytes_per_vcycle=algorithmic_bytes_per_vcycle(N, word=word), groups=h.dofs[-1].groups(), prof=prof,
               coarse_solver=h.mg.coarse_solver_used(),
               fused_transfer_bricks=sum(t.n_fused_bricks() for t in h.transfers[1:] if t is not None))
    if comm is not None:
        # pieces every level is cut into: n_ranks, n_ranks / group on the subset tier (each part held by a group of ranks), 1 = replicated
        res["level_layout"], res["subset_group"] = h.layout(), h.partition.group
    if diag is not None:
        res["rccl_diagnostics"] = diag
    if comm is not None:
        info = h.dofs[-1].info
        res["halo"] = dict(root_level=h.partition.root_level, peers=info.n_peers, halo_send_entries=info.n_halo_send, n_local=h.n_local)
    if details and comm is None:
        # (a) the same cycle with the tabulated coarse levels switched off
        lc = h.mg.set_collapse(False)
        res["collapse_level"] = lc
        res["collapse_level_dofs"] = N[lc] if lc else 0
        res["ms_no_collapse"] = h.mg.time_vcycles(z, b, max(steps // 2, 3), False)
        h.mg.set_collapse(True)
        # (b) per-level x per-stage times of the unchanged cycle (HIP events, no host synchronisation)
        reps = max(steps // 4, 3)
        h.mg.stage_timing(True)
        for _ in range(reps):
            h.mg.vmult(z, b)
        ms = h.mg.stage_times() / reps
        h.mg.stage_timing(False)
        L = len(N) - 1
        k = 3
        res["stage_ms_finest"] = {name: float(ms[s, L]) for s, name in ((0, "pre"), (1, "residual"), (2, "restrict"), (4, "prolongate"), (6, "post"))}
        res["stage_ms_total"] = float(ms.sum())
        res["stage_ms_per_level"] = [float(ms[:, l].sum()) for l in range(len(N))]
        # post-smoothing on the finest level: k operator passes moving 4 + 5 (k - 1) words per DoF, ALL kernels of the passes
        # (bricks, small slots, tail) between the two events
        words = 4 + 5 * (k - 1)
        res["pass_level"] = dict(stage="post-smoothing on the finest level: 3 Chebyshev operator passes, all kernels (bricks + small slots + tail)",
                                 words_per_dof=words, ms=float(ms[6, L]), algorithmic_bytes=8.0 * words * N[L],
                                 achieved_GBps=8.0 * words * N[L] / (ms[6, L] * 1e-3) / 1e9)
    # reference protocol for context: CG solve to reltol 1e-4 (ref:multigrid_throughput.cc:1238-1254); the reference's
    # own headline column throughput = n_dofs * n_iterations / time (ref:multigrid_throughput.cc:1282)
    if number_type != m.F64 and comm is not None:
        return res
    x = h.fine_operator.initialize_dof_vector() if number_type == m.F64 else m.Vector(ctx, h.dofs[-1].n_dofs)
    m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)  # warm-up
    sync()
    t0 = time.perf_counter()
    it, r = m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    sync()
    res["cg_iterations"], res["cg_time_s"] = it, time.perf_counter() - t0
    res["cg_throughput"] = h.n_dofs * it / res["cg_time_s"]
    return res


def rccl_diagnostics(m, ctx, comm, h, reps=20):
    """First contact of the sharded path with real RCCL, made a diagnosis instead of one number: per rank the halo plan of the
    finest level, a CHECKED exchange (ones on every DoF: after compress(add) a shared entry holds the number of ranks that share
    it, known from the halo plan), and the times of one halo exchange (pack + grouped ncclSend/ncclRecv + combine) and of one
    scalar all-reduce, host-synchronised averages over `reps` calls."""
    import numpy as np

    op, d = h.fine_operator, h.dofs[-1]
    info = d.info
    out = dict(rank=comm.rank, n_local_dofs=d.n_dofs, peers=int(info.n_peers), halo_send_entries=int(info.n_halo_send))
    s = comm.allreduce_sum(ctx, float(comm.rank + 1))
    out["allreduce_ok"] = bool(abs(s - comm.n_ranks * (comm.n_ranks + 1) / 2) < 1e-12)
    v = op.initialize_dof_vector()
    v.set(1.0)
    op.exchange_add_tail(v)
    ctx.synchronize()
    plan = d.halo_plan() if info.n_peers else None
    if plan is not None:
        got = v.to_host()[info.n_interior + np.asarray(plan["sh_tail"], dtype=np.int64)]
        want = np.diff(np.asarray(plan["sh_ptr"], dtype=np.int64)).astype(float)
        out["exchange_ok"] = bool(np.array_equal(got, want))
        out["shared_dofs"] = int(len(want))
    else:
        out["exchange_ok"] = True
        out["shared_dofs"] = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        op.exchange_add_tail(v)
    ctx.synchronize()
    out["exchange_us"] = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        comm.allreduce_sum(ctx, 1.0)
    out["allreduce_us"] = (time.perf_counter() - t0) / reps * 1e6
    return out


def pmc_traffic(n_ref, B):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 --pmc summary of this workload
    (profiles/*_pmc_traffic_octant<nref>_p4.json, made by tools/pmc_summary.py: FETCH_SIZE x2 + WRITE_SIZE, separate
    passes), averaged over the kernel's launches like `achieved`.  It is NOT measured in this run: the source file and the
    commit that last touched it are returned with it; None when no profile of this workload is committed."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_traffic_octant{n_ref}_p4.json")))
    if not files:
        return None, None
    rows = [r for r in json.load(open(files[-1]))["kernels"]
            if f"lattice_apply_kernel<double, 4, {B}, 2" in r["kernel"] or f"lattice_apply_persistent_kernel<double, 4, {B}, 2" in r["kernel"]]
    for r in rows:  # tools/pmc_vcycle.py (round 2) counts launches per V-cycle, tools/pmc_summary.py (round 1) per run
        r.setdefault("launches", r.get("launches_per_cycle", 0))
    n = sum(r["launches"] for r in rows)
    src = os.path.relpath(files[-1], ROOT)
    try:
        src += " @" + subprocess.check_output(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", files[-1]], text=True,
                                              stderr=subprocess.DEVNULL).strip()
    except Exception:
        pass
    return (sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n if n else None), src


def host_cpu_topology():
    """(usable cores for an OpenMP team, description): physical cores inside this process's affinity mask, capped by the
    cgroup CPU quota (the GPU boxes of this pool hand a 1-GPU job a share of the host)."""
    aff = os.sched_getaffinity(0)
    cores, sockets, cur = set(), set(), {}
    try:
        for line in list(open("/proc/cpuinfo")) + [""]:
            if ":" in line:
                k, v = line.split(":", 1)
                cur[k.strip()] = v.strip()
            elif cur:
                cpu = int(cur.get("processor", -1))
                key = (cur.get("physical id", "0"), cur.get("core id", str(cpu)))
                sockets.add(key[0])
                if cpu in aff:
                    cores.add(key)
                cur = {}
    except Exception:
        pass
    n = len(cores) or len(aff)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    use = min(n, quota) if quota else n
    desc = (f"host: {os.cpu_count()} logical CPUs, {max(len(sockets), 1)} socket(s); this job may use {len(aff)} logical / {n} physical "
            f"cores" + (f", cgroup quota {quota} CPUs" if quota else "") + f"; OpenMP team {use}")
    return use, desc


def cpu_baseline(m, geometry, n_ref, degree, headline_nref, max_seconds=25.0):
    """host-CPU baseline: the C++/OpenMP oracle ("port": deal.II cannot be built here; a scalar-per-cell restatement, well below
    what deal.II's cell-batch SIMD does) on a bounded sample of the SAME workload: by default the headline configuration itself
    (NRefGlobal 8, 137 M DoFs, a few V-cycles), every core this job may use; the library is rebuilt on this host when it was
    compiled for another CPU (oracle/cpu_oracle.py)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_oracle

    cores, desc = host_cpu_topology()
    cpu_oracle.set_num_threads(cores)
    fine = m.Triangulation(geometry, n_ref)
    trias = m.create_geometric_coarsening_sequence(fine)
    dofs = [m.DoFs(t, degree) for t in trias]
    levels, transfers, mg = cpu_oracle.build_from_dofs(dofs, m.transfer_tables, coarse="direct")
    b = dofs[-1].rhs_constant()
    t1 = mg.time_vcycles(b, 1)
    n = int(max(1, min(20, max_seconds / max(t1, 1e-6) / 2)))
    t = mg.time_vcycles(b, n)
    out = dict(value=dofs[-1].n_dofs / t, unit="DoF/s", cores=cpu_oracle.num_threads(), kind="port",
               sample=f"{n} V-cycles of {geometry} NRefGlobal={n_ref} p={degree} ({dofs[-1].n_dofs} DoFs), C++/OpenMP oracle "
                      f"(host restatement of the reference's CPU path, not deal.II), {t*1e3:.1f} ms/cycle",
               config_matches_headline=bool(n_ref == headline_nref), built_for_this_host=cpu_oracle.built_for_this_host(),
               host=desc, host_cpus=os.cpu_count())
    # the single-core figure next to it, on a coarser octant (a 1-core cycle of the sample takes too long)
    del levels, transfers, mg
    dofs1 = dofs[:-1] if n_ref <= 7 else dofs[:-2]
    levels, transfers, mg = cpu_oracle.build_from_dofs(dofs1, m.transfer_tables, coarse="direct")
    cpu_oracle.set_num_threads(1)
    b1 = dofs1[-1].rhs_constant()
    t_one = mg.time_vcycles(b1, 2)
    cpu_oracle.set_num_threads(cores)
    out["value_1core"] = dofs1[-1].n_dofs / t_one
    out["sample_1core"] = f"2 V-cycles of {geometry} NRefGlobal={n_ref - (1 if n_ref <= 7 else 2)} p={degree} ({dofs1[-1].n_dofs} DoFs), 1 thread"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="octant_p4")
    ap.add_argument("--nref", type=int, default=None, help="NRefGlobal of the primary workload (default: octant_p4 8 -- 9 = 1.1 G DoFs is the "
                    "weak-scaling companion for 8 GPUs; pmg_annulus: 9 on one GPU (149 M DoFs), 8 sharded)")
    ap.add_argument("--nref-p1", type=int, default=9, help="NRefGlobal of the secondary octant p=1 workload")
    ap.add_argument("--nref-uniform-p1", type=int, default=9, help="NRefGlobal of the uniform-mesh p=1 workload (135 M DoFs)")
    ap.add_argument("--cpu-nref", type=int, default=8, help="NRefGlobal of the CPU-baseline sample (default: the headline configuration)")
    ap.add_argument("--no-float", action="store_true", help="skip the MGNumberType float figure (also_float)")
    ap.add_argument("--coarse", default=None, help="CoarseGridSolverType override (pmg_annulus: cg_with_chebyshev | cg | amg | cg_with_amg | "
                    "gmg_vcycle = the geometric stand-in of rounds 1-2)")
    ap.add_argument("--coarse-cycles", type=int, default=None, help="CoarseSolverNCycles (amg, cg_with_amg, gmg_vcycle); default 2 for "
                    "pmg_annulus with amg (the reference's default.json), else 1")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded", help="N > 1: domain decomposition (default) or replicas")
    ap.add_argument("--subset-group", type=int, default=None, help="N > 1: ranks per part on the subset tier of the partition (levels of 1-4 M "
                    "DoFs are cut into N / group parts; default 4 from 8 ranks on, 2 from 4 on; 1 = replicate those levels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()
    geometry, nref_default, degree, mg_type, coarse, cfg_name = WORKLOADS[args.workload]
    coarse = args.coarse or coarse
    if args.coarse_cycles is None:
        args.coarse_cycles = 2 if (args.workload == "pmg_annulus" and coarse == "amg") else 1

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.workload == "pmg_annulus" and world == 1:
        nref_default = 9  # fills one GPU (149 M DoFs); NRefGlobal 8 (18.6 M) is the size quoted for 8 GPUs and is printed beside it
    nref = args.nref if args.nref is not None else nref_default
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    # MGAMD_BENCH_FORCE_SHARDED=1 (development): run the N > 1 code path -- RCCL communicator, DistributedHierarchy, diagnostics,
    # gathers -- with ONE rank, which is all a one-GPU box can execute of it (RCCL refuses two ranks on one device)
    force_sharded = world == 1 and os.environ.get("MGAMD_BENCH_FORCE_SHARDED") == "1"
    if world > 1 or force_sharded:
        import torch.distributed as dist

        if force_sharded:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        torch.cuda.synchronize()

    import dealii_multigrid_amd as m

    ctx = m.Context(local_rank)
    mode = "single" if (world == 1 and not force_sharded) else args.mode
    comm = None
    try:
        if mode == "sharded":
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(m.Communicator.rccl_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            comm = m.Communicator.rccl(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))
        prim = run_workload(m, ctx, geometry, nref, degree, mg_type, coarse, args.steps, args.warmup, barrier, sync, profile=True, comm=comm,
                            details=world == 1, diagnostics=True, coarse_cycles=args.coarse_cycles, subset_group=args.subset_group)
    except Exception as e:  # noqa: BLE001
        # no fallback, no relabelled metric: the other ranks may be blocked inside a collective, so leave hard with a
        # non-zero status and let the launcher tear the job down
        print(f"bench.py: rank {rank}/{world} failed ({mode}): {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        os._exit(3)
    elapsed = prim["elapsed"]
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    # sharded: ONE global problem; replicas: N copies of it
    value = (world if mode == "replicas" else 1) * prim["n_dofs"] / (elapsed / args.steps)

    collapsed = ""
    if prim.get("collapse_level"):
        collapsed = (f"; levels 0..{prim['collapse_level']} (<= {prim['collapse_level_dofs']} DoFs) applied as ONE tabulated dense matrix "
                     f"(result-equivalent, see ms_per_step_no_collapse)")
    out = {
        "metric": f"DoF/s per V-cycle, 3D {'octant' if geometry == 'quadrant' else geometry} p={degree}" + (" PMG" if mg_type == "PMG" else ""),
        "value": value,
        "unit": "DoF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak" if mode == "replicas" else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"3D {'octant (GeometryType quadrant)' if geometry == 'quadrant' else geometry} {mg_type} p={degree} FP64, "
                        f"NRefGlobal={nref}, SmootherDegree=3, coarse solver {prim['coarse_solver']}"
                        + (f" x{args.coarse_cycles} cycles" if args.coarse_cycles > 1 else "") + ", f=1, zero Dirichlet "
                        f"(BASELINE.json {cfg_name})" + collapsed,
            "n_dofs": prim["n_dofs"], "n_cells": prim["n_cells"], "n_levels": prim["n_levels"], "level_dofs": prim["level_dofs"],
            "parallelism": "1 GPU" if world == 1 else (
                f"domain decomposition over {world} GPUs: RCCL send/recv halo exchange, rank-group tier for the mid-size levels (level_layout), "
                f"all-reduce onto the replicated coarse levels"
                if mode == "sharded" else f"replicas x{world} (no data-path collective)"),
            "cg_iterations_reltol_1e-4": prim["cg_iterations"],
            "cg_throughput_dofs_x_iterations_per_s": prim["cg_throughput"],
            "chebyshev_start_vector": "deal.II's (i mod 11) - mean on the local numbering" if world == 1 or mode == "replicas" else
                                      "sharded run: numbering-independent hash of the geometric DoF key (mod 11) - mean; a 1-GPU run uses "
                                      "the index-based vector, so eigenvalue estimates (and borderline CG counts) can differ slightly",
        },
    }
    if "ms_no_collapse" in prim:
        out["ms_per_step_no_collapse"] = prim["ms_no_collapse"]
    out["config"]["fused_transfer_bricks"] = prim["fused_transfer_bricks"]
    if "level_layout" in prim:
        out["config"]["level_layout"] = prim["level_layout"]  # pieces per level, coarse -> fine (tiers)
        out["config"]["subset_group"] = prim["subset_group"]
    if "halo" in prim:
        out["config"]["halo_rank0"] = prim["halo"]
    if "rccl_diagnostics" in prim:
        # every rank's view (halo plan sizes, checked exchange, exchange / all-reduce times), gathered on rank 0
        gathered = [None] * world
        dist.all_gather_object(gathered, prim["rccl_diagnostics"])
        out["rccl_diagnostics"] = gathered
        if not all(g["exchange_ok"] and g["allreduce_ok"] for g in gathered):
            print(f"bench.py: RCCL self-check failed: {gathered}", file=sys.stderr, flush=True)
            os._exit(3)
    # whole-V-cycle roofline figure (against the aggregate HBM bandwidth of the GPUs used) and the dominant kernel's
    vcycle_gbs = (world if mode == "replicas" else 1) * prim["bytes_per_vcycle"] / (elapsed / args.steps) / 1e9 / world
    out["vcycle_algorithmic_GBps_per_gpu"] = vcycle_gbs
    out["vcycle_frac_of_hbm_peak"] = vcycle_gbs / HBM_PEAK_GBS
    out["vcycle_bytes_model"] = ("SURVEY 8(d): 8 B x [(10k+3) sum_{l>=1} N_l + 2 sum_{l<L} N_l], k=3 (the reference algorithm's compulsory "
                                 "traffic; this implementation never stores x_1 of the zero-start smoother and moves 5 words per DoF and "
                                 "level fewer, so this fraction is the model's bytes over time, not achieved bandwidth)")
    if prim["prof"] and prim["prof"][1] > 0:
        ms, n, by, by_moved = prim["prof"]
        achieved = by / (ms * 1e-3) / 1e9
        B = max(prim["groups"], key=lambda g: g[1] * (degree * g[0] + 1) ** 3)[0]
        traffic, traffic_src = pmc_traffic(nref, B) if args.workload == "octant_p4" else (None, None)
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": traffic_src,
                           "kernel": f"lattice_apply_persistent_kernel<double,{degree},{B},MODE_CHEB> (the Chebyshev passes that read x and x_old from memory; all "
                                     f"launches of this symbol in the timed region).  Algorithmic bytes per brick (SURVEY 8(d) per-unit figure): 5 (4 "
                                     f"without x_old) words (x, x_old, b, D^-1, out) x {(degree * B - 1) ** 3} slot-interior DoFs + 2 words (gathered x, "
                                     f"partial sum) x {(degree * B) ** 3 - (degree * B - 1) ** 3} shell DoFs whose epilogue tail_kernel finishes.  The kernel "
                                     f"itself evaluates the interior D^-1 in closed form instead of reading it: `achieved_moved` counts 4 (3) words",
                           "launches": n, "avg_launch_us": ms / n * 1e3, "algorithmic_bytes_per_launch": by / n,
                           "achieved_moved": by_moved / (ms * 1e-3) / 1e9, "moved_bytes_per_launch": by_moved / n}
        if "pass_level" in prim:
            pl = dict(prim["pass_level"])
            pl["frac"] = pl["achieved_GBps"] / HBM_PEAK_GBS
            out["roofline"]["pass_level"] = pl
    for k in ("stage_ms_finest", "stage_ms_total", "stage_ms_per_level"):
        if k in prim:
            out[k] = prim[k]
    if not args.no_secondary and args.workload == "octant_p4" and (rank == 0 or mode == "sharded"):
        # octant p=1: on one GPU (rank 0), or sharded like the primary workload when that ran sharded
        sharded2 = mode == "sharded"
        try:
            sec = run_workload(m, ctx, "quadrant", args.nref_p1, 1, "HMG-global", "amg", args.steps, args.warmup,
                               barrier if sharded2 else (lambda: None), sync, profile=False, comm=comm if sharded2 else None, details=world == 1,
                               subset_group=args.subset_group)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}/{world} failed in the p=1 workload: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            os._exit(3)
        t = sec["elapsed"]
        if sharded2:
            tt = torch.tensor([t], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        t /= args.steps
        out["also"] = {"metric": "DoF/s per V-cycle, 3D octant p=1", "value": sec["n_dofs"] / t, "unit": "DoF/s",
                       "n_gpus": world if sharded2 else 1, "ms_per_step": t * 1e3, "n_dofs": sec["n_dofs"], "NRefGlobal": args.nref_p1,
                       "vcycle_frac_of_hbm_peak": sec["bytes_per_vcycle"] / t / 1e9 / HBM_PEAK_GBS / (world if sharded2 else 1),
                       "cg_iterations_reltol_1e-4": sec["cg_iterations"]}
        for k in ("ms_no_collapse", "collapse_level", "collapse_level_dofs", "stage_ms_per_level"):
            if k in sec:
                out["also"][k] = sec[k]
        if world == 1:
            # BASELINE.json configs[1]: p = 1 on a uniformly refined mesh that fills the GPU (135 M DoFs)
            uni = run_workload(m, ctx, "hypercube", args.nref_uniform_p1, 1, "HMG-global", "amg", max(args.steps // 2, 3), 2, lambda: None, sync,
                               profile=False)
            tu = uni["elapsed"] / max(args.steps // 2, 3)
            out["also_uniform_p1"] = {"metric": "DoF/s per V-cycle, 3D uniform cube p=1 (BASELINE.json configs[1])", "value": uni["n_dofs"] / tu,
                                      "unit": "DoF/s", "n_gpus": 1, "ms_per_step": tu * 1e3, "n_dofs": uni["n_dofs"],
                                      "NRefGlobal": args.nref_uniform_p1, "vcycle_frac_of_hbm_peak": uni["bytes_per_vcycle"] / tu / 1e9 / HBM_PEAK_GBS,
                                      "cg_iterations_reltol_1e-4": uni["cg_iterations"]}
    if world == 1 and args.workload == "octant_p4" and not args.no_float:
        # MGNumberType "float" is the reference's default (ref:scripts/default.json:16, ref:multigrid_throughput.cc:2430-2433): the
        # same hierarchy with FP32 level vectors under the FP64 outer vectors (copy_to_mg / copy_from_mg cast); 4-byte words in
        # the byte model.  Its own line: a different dtype, never mixed into `value`.
        flt = run_workload(m, ctx, geometry, nref, degree, mg_type, coarse, max(args.steps // 2, 3), 2, lambda: None, sync, profile=False,
                           number_type=m.F32)
        tf = flt["elapsed"] / max(args.steps // 2, 3)
        out["also_float"] = {"metric": f"DoF/s per V-cycle, 3D octant p={degree}, MGNumberType float (FP32 levels under FP64 CG vectors)",
                             "value": flt["n_dofs"] / tf, "unit": "DoF/s", "n_gpus": 1, "ms_per_step": tf * 1e3, "dtype": "f32",
                             "n_dofs": flt["n_dofs"], "NRefGlobal": nref, "bytes_model": "SURVEY 8(d) with 4-byte words",
                             "vcycle_frac_of_hbm_peak": flt["bytes_per_vcycle"] / tf / 1e9 / HBM_PEAK_GBS,
                             "fused_transfer_bricks": flt["fused_transfer_bricks"], "cg_iterations_reltol_1e-4": flt.get("cg_iterations"),
                             "note": "same kernels as FP64 (persistent 17^3 bricks, fused transfers) on float level vectors with float constants"}
    if world == 1 and args.workload == "pmg_annulus" and nref != 8 and not args.no_secondary:
        # the size BASELINE.json configs[4] shards over 8 GPUs, on one GPU for reference
        sm = run_workload(m, ctx, geometry, 8, degree, mg_type, coarse, args.steps, args.warmup, lambda: None, sync, profile=False,
                          coarse_cycles=args.coarse_cycles)
        ts = sm["elapsed"] / args.steps
        out["also_sharded_size"] = {"metric": out["metric"], "value": sm["n_dofs"] / ts, "unit": "DoF/s", "n_gpus": 1, "ms_per_step": ts * 1e3,
                                    "n_dofs": sm["n_dofs"], "NRefGlobal": 8, "coarse_solver": sm["coarse_solver"],
                                    "cg_iterations_reltol_1e-4": sm["cg_iterations"]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(m, "quadrant", args.cpu_nref, 4, nref if args.workload == "octant_p4" else -1)
    barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
