#!/usr/bin/env python3
"""bench.py -- DoF/s per V-cycle of the global-coarsening multigrid preconditioner (BASELINE.json metric).

A "step" is one application of PreconditionMG::vmult (copy_to_mg + V-cycle + copy_from_mg,
ref:multigrid_throughput.cc:1132-1133) on the finest level of the workload, with the right-hand side
already resident in HBM.  Workload (config.workload): the configuration the metric is quoted on,
BASELINE.json configs[2]: 3D octant (GeometryType "quadrant"), global coarsening, p = 4, FP64,
SmootherDegree 3 -- synthetic data (f == 1, zero Dirichlet), no dataset.  The p = 1 octant number that
north_star also asks for is reported in the same line under "also".

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU.  The SAME global problem is sharded by spatial domain
   decomposition: every rank owns a Morton chunk of the octree, shared DoFs are exchanged with grouped RCCL
   send/recv over xGMI, the replicated coarse levels take one RCCL all-reduce -> "scaling": "strong",
   value = global n_dofs / max-over-ranks time; see DESIGN.md section 7.  `--mode replicas` runs N independent copies.)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_vcycle(n_dofs_per_level, k=3, word=8):
    """s * [ (10k+3) * sum_{l>=1} N_l + 2 * sum_{l<L} N_l ]  (the reference algorithm's compulsory
    vector traffic with x_1 of the zero-start smoother stored; the implementation recomputes x_1 = D^-1 b / theta inside
    the first two operator passes and moves 5 words per DoF and level fewer)."""
    N = n_dofs_per_level
    return word * ((10 * k + 3) * sum(N[1:]) + 2 * sum(N[:-1]))


def run_workload(m, ctx, geometry, n_ref, degree, steps, warmup, barrier, sync, profile, comm=None):
    t0 = time.time()
    if comm is None:
        h = m.Hierarchy(ctx, geometry, n_ref, degree, "HMG-global", smoother_degree=3, coarse_solver="amg", number_type=m.F64)
    else:
        h = m.DistributedHierarchy(ctx, comm, geometry, n_ref, degree, smoother_degree=3, coarse_solver="amg", number_type=m.F64)
    b, z = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    ctx.synchronize()
    setup_s = time.time() - t0
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
        prof = ctx.kernel_profile_read()
        ctx.kernel_profile(False)
    if comm is None:
        N = [d.n_dofs for d in h.dofs]
    else:  # global level sizes: owned DoFs summed over the ranks (replicated levels are complete on every rank)
        N = [int(round(comm.allreduce_sum(ctx, float(op.n_owned())))) if l >= h.partition.root_level else h.dofs[l].n_dofs
             for l, op in enumerate(h.operators)]
    res = dict(n_dofs=h.n_dofs, n_cells=h.trias[-1].n_cells, n_levels=len(N), level_dofs=N, elapsed=elapsed, setup_s=setup_s,
               bytes_per_vcycle=algorithmic_bytes_per_vcycle(N), groups=h.dofs[-1].groups(), prof=prof)
    if comm is not None:
        info = h.dofs[-1].info
        res["halo"] = dict(root_level=h.partition.root_level, peers=info.n_peers, halo_send_entries=info.n_halo_send, n_local=h.n_local)
    # reference protocol for context: CG solve to reltol 1e-4 (ref:multigrid_throughput.cc:1238-1254)
    x = h.fine_operator.initialize_dof_vector()
    t0 = time.perf_counter()
    it, r = m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    res["cg_iterations"], res["cg_time_s"] = it, time.perf_counter() - t0
    return res


def pmc_traffic(n_ref, B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this workload
    (profiles/*_pmc_traffic_octant<nref>_p4.json, made by tools/pmc_summary.py: FETCH_SIZE x2 + WRITE_SIZE, separate
    passes), averaged over the kernel's launches like `achieved`; None when no profile of this workload is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_traffic_octant{n_ref}_p4.json")))
    if not files:
        return None
    rows = [r for r in json.load(open(files[-1]))["kernels"] if f"lattice_apply_kernel<double, 4, {B}, 2>" in r["kernel"]]
    n = sum(r["launches"] for r in rows)
    return sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n if n else None


def cpu_baseline(m, geometry, n_ref, degree, max_seconds=25.0):
    """host-CPU baseline: the C++/OpenMP oracle ("port": deal.II cannot be built here) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_oracle

    fine = m.Triangulation(geometry, n_ref)
    trias = m.create_geometric_coarsening_sequence(fine)
    dofs = [m.DoFs(t, degree) for t in trias]
    levels, transfers, mg = cpu_oracle.build_from_dofs(dofs, m.transfer_tables, coarse="direct")
    b = dofs[-1].rhs_constant()
    t1 = mg.time_vcycles(b, 1)
    n = int(max(1, min(20, max_seconds / max(t1, 1e-6) / 2)))
    t = mg.time_vcycles(b, n)
    cores = cpu_oracle.num_threads()
    cpu_oracle.set_num_threads(1)  # the single-core figure next to it
    t_one = mg.time_vcycles(b, int(max(1, min(3, 5.0 / max(t * cores, 1e-6)))))
    cpu_oracle.set_num_threads(cores)
    return dict(value=dofs[-1].n_dofs / t, unit="DoF/s", cores=cores, kind="port",
                sample=f"{n} V-cycles of {geometry} NRefGlobal={n_ref} p={degree} ({dofs[-1].n_dofs} DoFs), C++/OpenMP oracle, "
                       f"{t*1e3:.1f} ms/cycle",
                value_1core=dofs[-1].n_dofs / t_one, host_cpus=os.cpu_count())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nref", type=int, default=8, help="NRefGlobal of the primary workload (octant p=4)")
    ap.add_argument("--nref-p1", type=int, default=9, help="NRefGlobal of the secondary octant p=1 workload")
    ap.add_argument("--cpu-nref", type=int, default=6, help="NRefGlobal of the CPU-baseline sample")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded", help="N > 1: domain decomposition (default) or replicas")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        torch.cuda.synchronize()

    import dealii_multigrid_amd as m

    ctx = m.Context(local_rank)
    mode, note = ("single", None) if world == 1 else (args.mode, None)
    prim = None
    if mode == "sharded":
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(m.Communicator.rccl_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            comm = m.Communicator.rccl(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))
            prim = run_workload(m, ctx, "quadrant", args.nref, 4, args.steps, args.warmup, barrier, sync, profile=True, comm=comm)
            ok = 1.0
        except Exception as e:  # noqa: BLE001 -- reported in the JSON line, never silent
            ok, note = 0.0, f"sharded run failed on rank {rank}: {type(e).__name__}: {e}"
            print(note, file=sys.stderr, flush=True)
        flag = torch.tensor([ok], device="cuda", dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 0.5:
            mode, prim = "replicas", None
            note = note or "sharded run failed on another rank"
    if prim is None:
        prim = run_workload(m, ctx, "quadrant", args.nref, 4, args.steps, args.warmup, barrier, sync, profile=True)
    elapsed = prim["elapsed"]
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    # sharded: ONE global problem; replicas: N copies of it
    value = (world if mode == "replicas" else 1) * prim["n_dofs"] / (elapsed / args.steps)

    out = {
        "metric": "DoF/s per V-cycle, 3D octant p=4",
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
            "workload": f"3D octant (GeometryType quadrant) HMG-global p=4 FP64, NRefGlobal={args.nref}, SmootherDegree=3, "
                        f"coarse solver direct, f=1, zero Dirichlet (BASELINE.json configs[2])",
            "n_dofs": prim["n_dofs"], "n_cells": prim["n_cells"], "n_levels": prim["n_levels"],
            "parallelism": "1 GPU" if world == 1 else (
                f"domain decomposition over {world} GPUs: RCCL send/recv halo exchange + all-reduce onto replicated coarse levels"
                if mode == "sharded" else f"replicas x{world} (no data-path collective)"),
            "cg_iterations_reltol_1e-4": prim["cg_iterations"],
        },
    }
    if note:
        out["config"]["note"] = note
    if "halo" in prim:
        out["config"]["halo_rank0"] = prim["halo"]
    # whole-V-cycle roofline figure (against the aggregate HBM bandwidth of the GPUs used) and the dominant kernel's
    vcycle_gbs = (world if mode == "replicas" else 1) * prim["bytes_per_vcycle"] / (elapsed / args.steps) / 1e9 / world
    out["vcycle_algorithmic_GBps_per_gpu"] = vcycle_gbs
    out["vcycle_frac_of_hbm_peak"] = vcycle_gbs / HBM_PEAK_GBS
    out["vcycle_bytes_model"] = "SURVEY 8(d): 8 B x [(10k+3) sum_{l>=1} N_l + 2 sum_{l<L} N_l], k=3"
    if prim["prof"] and prim["prof"][1] > 0:
        ms, n, by = prim["prof"]
        achieved = by / (ms * 1e-3) / 1e9
        B = max(prim["groups"], key=lambda g: g[1] * (4 * g[0] + 1) ** 3)[0]
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": pmc_traffic(args.nref, B),
                           "kernel": f"lattice_apply_kernel<double,4,{B},MODE_CHEB> (the Chebyshev passes that read x and x_old from memory: "
                                     f"5 words/DoF with x_old, 4 without; all launches of this symbol in the timed region)",
                           "launches": n, "avg_launch_us": ms / n * 1e3, "algorithmic_bytes_per_launch": by / n}
    if not args.no_secondary and (rank == 0 or mode == "sharded"):
        # octant p=1: on one GPU (rank 0), or sharded like the primary workload when that ran sharded
        sharded2 = mode == "sharded"
        sec = run_workload(m, ctx, "quadrant", args.nref_p1, 1, args.steps, args.warmup, barrier if sharded2 else (lambda: None), sync,
                           profile=False, comm=comm if sharded2 else None)
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(m, "quadrant", args.cpu_nref, 4)
    barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
