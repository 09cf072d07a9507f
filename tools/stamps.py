"""In-kernel phase timeline of the dominant lattice kernel (development tool; MGAMD_STAMPS must be set)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dealii_multigrid_amd as m

geo, L, p = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mode = int(os.environ["MGAMD_STAMPS"])
ctx = m.Context(0)
d = m.DoFs(m.Triangulation(geo, L), p, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
op = m.Operator(ctx, d)
n = d.n_dofs
x, y, b = (op.initialize_dof_vector() for _ in range(3))
x.from_host(np.random.default_rng(0).standard_normal(n)); b.from_host(np.random.default_rng(1).standard_normal(n))
if mode == 0:
    for _ in range(3): op.vmult(y, x)
elif mode in (4, 5):  # zero-start passes
    ch = m.PreconditionChebyshev(op, 3, 20.0, 2)
    for _ in range(3): ch.vmult(y, b)
else:
    ch = m.PreconditionChebyshev(op, 3, 20.0, 2)
    for _ in range(3): ch.step(y, b)
ctx.synchronize()
buf = np.zeros(8 * 70000, np.uint64); cnt = C.c_uint64()
m._chk(m._lib.mgamd_level_op_debug_stamps(op._h, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), C.byref(cnt)))
st = buf[: cnt.value].reshape(-1, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
tick = 0.01  # us per tick (100 MHz)
dur = (st[:, 4] - st[:, 0]) * tick
print(f"{geo} L={L} p={p} mode={mode}: {len(st)} workgroups, kernel span {(st[:,4].max()-t0)*tick:.1f} us")
print(f"  per-WG total   : median {np.median(dur):6.2f} us  p10 {np.percentile(dur,10):6.2f}  p90 {np.percentile(dur,90):6.2f}")
for k, name in enumerate(["gather (issue+wait+LDS)", "sweeps (+epilogue operand issue)", "interior epilogue", "shell atomics + drain"]):
    ph = (st[:, k + 1] - st[:, k]) * tick
    print(f"  {name:34s}: median {np.median(ph):6.2f} us  p10 {np.percentile(ph,10):6.2f}  p90 {np.percentile(ph,90):6.2f}")
if (st[:, 5] > 0).all() and (st[:, 6] > 0).all():  # the one-shot body also stamps the constraint passes
    for name, a, b in [("  of which: interpolation passes", 1, 5), ("            sweeps", 5, 6), ("            transposed passes", 6, 2)]:
        ph = (st[:, b] - st[:, a]) * tick
        print(f"  {name:34s}: median {np.median(ph):6.2f} us  p10 {np.percentile(ph,10):6.2f}  p90 {np.percentile(ph,90):6.2f}")
starts = np.sort(st[:, 0] - t0) * tick
print("  WG start times (us) quantiles:", [round(float(np.percentile(starts, q)), 1) for q in (0, 5, 25, 50, 75, 95, 100)])
# concurrency: average number of WGs alive
ends = (st[:, 4] - t0) * tick
span = ends.max()
print(f"  avg concurrent WGs: {dur.sum()/span:.1f}  (256 CUs)")
