"""Run N V-cycles of one workload between two marker kernels, for `rocprofv3 --kernel-trace --output-format csv`:
tools/vcycle_table.py then keeps only the dispatches between the markers, so that the per-kernel sums add up to the
V-cycle time (no setup, no eigenvalue estimation, no collapse tabulation in the statistics).

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 tools/vcycle_trace.py quadrant 8 4 [cycles]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dealii_multigrid_amd as m

MARKER_N = 77777  # vec_set on a vector of this length = the marker dispatch

geo, L, p = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cycles = int(sys.argv[4]) if len(sys.argv) > 4 else 5
mg_type = sys.argv[5] if len(sys.argv) > 5 else "HMG-global"
coarse = sys.argv[6] if len(sys.argv) > 6 else ("amg" if mg_type == "HMG-global" else "cg_with_chebyshev")
coarse_cycles = int(sys.argv[7]) if len(sys.argv) > 7 else 1
number_type = m.F32 if os.environ.get("MGAMD_TRACE_FLOAT") else m.F64  # MGNumberType float: FP32 levels under FP64 outer vectors
ctx = m.Context(0)
h = m.Hierarchy(ctx, geo, L, p, mg_type, coarse_solver=coarse, coarse_n_cycles=coarse_cycles, number_type=number_type)
if number_type == m.F64:
    b, z = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
else:
    import numpy as np

    b, z = m.Vector(ctx, h.dofs[-1].n_dofs), m.Vector(ctx, h.dofs[-1].n_dofs)
    b.from_host(np.asarray(h.dofs[-1].rhs_constant()))
marker = m.Vector(ctx, MARKER_N)
# calibration dispatches for the PMC summaries (tools/pmc_vcycle.py): y = s y + a x on 32 M doubles reads 2 words and
# writes 1 word per entry
ca, cb = m.Vector(ctx, 1 << 25), m.Vector(ctx, 1 << 25)
ca.set(1.0), cb.set(2.0)
for _ in range(3):
    ca.sadd(0.5, 0.25, cb)
for _ in range(3):
    h.mg.vmult(z, b)
ctx.synchronize()
marker.set(1.0)
for _ in range(cycles):
    h.mg.vmult(z, b)
marker.set(2.0)
ctx.synchronize()
ms = h.mg.time_vcycles(z, b, cycles, False)
print(f"{geo} L={L} p={p} {mg_type}: n_dofs={h.n_dofs} levels={[d.n_dofs for d in h.dofs]} cycles={cycles} eager {ms:.3f} ms/cycle", flush=True)
