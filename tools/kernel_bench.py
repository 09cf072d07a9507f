"""Micro-benchmark of the level operator kernels on the finest level (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dealii_multigrid_amd as m

geo, L, p = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = 20
ctx = m.Context(0)
t = m.Triangulation(geo, L)
d = m.DoFs(t, p, int(os.environ.get("MGAMD_MAX_BRICK", "0")))
op = m.Operator(ctx, d)
n = d.n_dofs
x, y, b = (op.initialize_dof_vector() for _ in range(3))
x.from_host(np.random.default_rng(0).standard_normal(n))
b.from_host(np.random.default_rng(1).standard_normal(n))
def timeit(fn):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
tv = timeit(lambda: op.vmult(y, x))
ch = m.PreconditionChebyshev(op, 3, 20.0, 2)
ts = timeit(lambda: ch.step(y, b))   # 3 operator passes (4,5,5 words) + maybe a copy
tz = timeit(lambda: ch.vmult(y, b))  # 1 vector pass + 2 operator passes
print(f"ABLATE={os.environ.get('MGAMD_ABLATE','0'):>3} {geo} L={L} p={p} n={n} groups={d.groups()} I/T={d.info.n_interior}/{d.info.n_tail}: "
      f"vmult {tv:8.1f} us ({2*8*n/tv*1e-6:6.3f} TB/s)   cheb.step {ts:8.1f} us ({14*8*n/ts*1e-6:6.3f} TB/s)   cheb.vmult {tz:8.1f} us", flush=True)
