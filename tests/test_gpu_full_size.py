"""The bench workloads AT FULL SIZE are checked runs (round 3): octant p = 4 NRefGlobal 8 (137 M DoFs, BASELINE.json configs[2],
the bench line) and the uniform cube p = 1 NRefGlobal 9 (135 M DoFs, configs[1]).  No oracle reaches these sizes, so the checks
are the size-independent properties of the preconditioned solver:
  * the V-cycle (same Chebyshev smoother before and after, exact coarse solve) is a SYMMETRIC linear operator: u.Mv = v.Mu to
    1e-10 and M(a u + b v) = a Mu + b Mv to rounding -- any misplaced transfer weight, ownership flag or tail contribution of the
    brick / fused-transfer kernels breaks one of the two;
  * constrained rows of the level operator are identity rows (ref:include/operator.h:170-172);
  * repeated application gives the same vector (no state left between cycles);
  * CG to reltol 1e-4 takes the iteration count the CPU oracle confirms one refinement level down
    (tests/test_gpu_vs_cpu_oracle.py: octant p=4 L=7, octant p=1 L=9), h-independent for this multigrid.
Inputs are seeded random vectors on the free DoFs; inner products and vector updates are device operations (mgamd_vec_dot,
mgamd_vec_sadd), so only the inputs cross PCIe."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]


def _random_free(mgamd, ctx, n, first_constrained, seed):
    x = np.random.default_rng(seed).standard_normal(n)
    x[first_constrained:] = 0.0
    return mgamd.Vector(ctx, n).from_host(x)


# the annulus (configs[4]'s mesh, here with h-multigrid at p = 4) is the case whose 17-point bricks border constrained cells: bricks
# that a hanging-node constraint reaches must stay out of the fused transfers (transfer_tables.hpp; an ownership plan without that
# rule lost their residual contributions at NRefGlobal 8 only -- CG then stalls)
@pytest.mark.parametrize("geo,L,p,expected_iterations,min_dofs,min_fused",
                         [("quadrant", 8, 4, 3, 130_000_000, 20000), ("hypercube", 9, 1, 4, 130_000_000, 20000), ("annulus", 8, 4, 4, 18_000_000, 300)])
def test_bench_workload_at_full_size(mgamd, ctx, geo, L, p, expected_iterations, min_dofs, min_fused):
    h = mgamd.Hierarchy(ctx, geo, L, p, "HMG-global", coarse_solver="amg")
    info = h.dofs[-1].info
    n, first_c = h.n_dofs, info.n_interior + info.n_tail
    assert n > min_dofs
    assert sum(t.n_fused_bricks() for t in h.transfers[1:]) > min_fused  # the fused-transfer kernels are what runs
    u, v = _random_free(mgamd, ctx, n, first_c, 1), _random_free(mgamd, ctx, n, first_c, 2)
    zu, zv, zw, w = (mgamd.Vector(ctx, n) for _ in range(4))
    h.mg.vmult(zu, u)
    h.mg.vmult(zv, v)
    # symmetry
    a, b = v.dot(zu), u.dot(zv)
    assert abs(a - b) <= 1e-10 * max(abs(a), abs(b)), (a, b)
    assert u.dot(zu) > 0 and v.dot(zv) > 0  # positive on these vectors
    # linearity: M(2u - 3v) = 2Mu - 3Mv
    w.copy_from(u)
    w.sadd(2.0, -3.0, v)
    h.mg.vmult(zw, w)
    zw.sadd(1.0, -2.0, zu)
    zw.sadd(1.0, 3.0, zv)
    assert np.sqrt(zw.dot(zw)) <= 1e-12 * (2 * np.sqrt(zu.dot(zu)) + 3 * np.sqrt(zv.dot(zv)))
    # the same vector again
    h.mg.vmult(zw, u)
    zw.sadd(1.0, -1.0, zu)
    assert np.sqrt(zw.dot(zw)) <= 1e-13 * np.sqrt(zu.dot(zu))
    # identity rows of the operator on the constrained DoFs
    x = np.random.default_rng(3).standard_normal(n)
    src, dst = mgamd.Vector(ctx, n).from_host(x), mgamd.Vector(ctx, n)
    h.fine_operator.vmult(dst, src)
    assert np.array_equal(dst.to_host()[first_c:], x[first_c:])
    del src, dst, x
    # the solve of the bench protocol
    rhs, sol = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(rhs)
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, sol, rhs, 1e-4)
    assert it == expected_iterations
    # true residual of the returned solution
    h.fine_operator.vmult(zw, sol)
    zw.sadd(-1.0, 1.0, rhs)
    assert np.sqrt(zw.dot(zw)) <= 1.05e-4 * np.sqrt(rhs.dot(rhs))
