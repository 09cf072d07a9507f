import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dealii_multigrid_amd as m
ctx = m.Context(0)
for geo, L, p in [("quadrant", 8, 4), ("quadrant", 9, 1), ("hypercube", 9, 1)]:
    h = m.Hierarchy(ctx, geo, L, p, "HMG-global", coarse_solver="amg", number_type=m.F32)
    n = h.n_dofs
    b, z = m.Vector(ctx, n), m.Vector(ctx, n)   # double in/out, float levels
    h.fine_operator.rhs(b) if hasattr(h.fine_operator, "rhs") else None
    for _ in range(3): h.mg.vmult(z, b)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): h.mg.vmult(z, b)
    ctx.synchronize(); t = (time.perf_counter() - t0) / 20
    x = m.Vector(ctx, n)
    it, r = m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    print(f"{geo} L={L} p={p} FP32 levels under FP64 CG: n={n} V-cycle {t*1e3:.3f} ms -> {n/t:.3e} DoF/s, CG iterations {it}")
    del h
