"""One-off wide sweep: GPU V-cycle / CG solve against the C++ CPU oracle over geometries, degrees and sizes, with the
production slot policy (max_brick=-1).  Development aid; the test-suite holds the pinned subset."""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import numpy as np
import dealii_multigrid_amd as m
import cpu_oracle

def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
ctx = m.Context(0)
cases = []
for p in (1, 2, 3, 4):
    cases += [("quadrant", {1: 7, 2: 6, 3: 5, 4: 5}[p], p, "HMG-global"), ("annulus", {1: 7, 2: 6, 3: 6, 4: 5}[p], p, "HMG-global"),
              ("circle", {1: 6, 2: 5, 3: 5, 4: 4}[p], p, "HMG-global"), ("hypercube", {1: 6, 2: 5, 3: 4, 4: 4}[p], p, "HMG-global")]
cases += [("quadrant", 5, 4, "HPMG"), ("annulus", 6, 4, "PMG"), ("quadrant", 6, 3, "HMG-global"), ("quadrant", 6, 4, "HMG-global")]
if len(sys.argv) > 1 and sys.argv[1] == "large":  # bricks and constrained families on every geometry
    cases = [("annulus", 7, 4, "HMG-global"), ("circle", 6, 4, "HMG-global"), ("quadrant", 7, 2, "HMG-global"), ("annulus", 8, 2, "HMG-global"),
             ("hypercube", 5, 4, "HMG-global"), ("circle", 7, 3, "HMG-global"), ("quadrant", 8, 1, "HMG-global"), ("annulus", 8, 1, "HMG-global"),
             ("annulus", 7, 4, "PMG"), ("quadrant", 6, 4, "HPMG")]
worst = 0.0
for geo, L, p, typ in cases:
    t0 = time.time()
    coarse = "amg" if typ == "HMG-global" else "cg_with_chebyshev"
    h = m.Hierarchy(ctx, geo, L, p, typ, coarse_solver=coarse)
    levels, transfers, mg = cpu_oracle.build_from_dofs(h.dofs, m.transfer_tables, coarse=coarse)
    n = h.n_dofs
    r = np.random.default_rng(3).standard_normal(n)
    vr, vz = m.Vector(ctx, n).from_host(r), m.Vector(ctx, n)
    h.mg.vmult(vz, vr)
    ev = rel(vz.to_host(), mg.vcycle(r))
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    it, res = m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    xs, its, ress = mg.solve_cg(b.to_host(), 1e-4)
    es = rel(x.to_host(), xs)
    tol = 1e-10 if coarse == "amg" else 1e-5
    flag = "ok" if (it == its and ev < tol and es < tol) else "MISMATCH"
    worst = max(worst, ev if coarse == "amg" else 0.0)
    print(f"{flag:8s} {geo:9s} L={L} p={p} {typ:10s} n={n:8d} groups={h.dofs[-1].groups()} its gpu/cpu {it}/{its} vcycle err {ev:.2e} solve err {es:.2e} ({time.time()-t0:.1f}s)", flush=True)
    del h
print("worst V-cycle error (direct coarse):", worst)
