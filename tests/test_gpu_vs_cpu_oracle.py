"""GPU path against the C++ CPU oracle at sizes the numpy oracle cannot reach (10^5 - 2 10^7 DoFs):
iteration counts equal, CG iterates agree to <= 1e-10 relative (north_star: stated FP64 tolerance)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("geo,L,p,typ", [("quadrant", 5, 4, "HMG-global"), ("hypercube", 5, 1, "HMG-global"), ("quadrant", 6, 1, "HMG-global"),
                                         ("annulus", 6, 2, "HMG-global"), ("annulus", 5, 4, "PMG"), ("hypercube", 4, 3, "HMG-global"),
                                         # more 17^3-lattice bricks than resident workgroups (512): the persistent kernels walk
                                         # several bricks per workgroup (3375 bricks at p = 4; 3375 plain + 721 constrained at p = 1)
                                         ("quadrant", 7, 4, "HMG-global"), ("quadrant", 9, 1, "HMG-global")])
def test_solve_matches_cpu_oracle(mgamd, ctx, geo, L, p, typ):
    import cpu_oracle

    coarse = "amg" if typ == "HMG-global" else "cg_with_chebyshev"
    h = mgamd.Hierarchy(ctx, geo, L, p, typ, coarse_solver=coarse, max_brick=0)
    levels, transfers, mg = cpu_oracle.build_from_dofs(h.dofs, mgamd.transfer_tables, coarse=coarse)
    n = h.n_dofs
    rng = np.random.default_rng(11)
    # single operator application, every level
    for l, op in enumerate(h.operators):
        x = rng.standard_normal(h.dofs[l].n_dofs)
        src, dst = op.initialize_dof_vector().from_host(x), op.initialize_dof_vector()
        op.vmult(dst, src)
        assert rel_err(dst.to_host(), levels[l].vmult(x)) < 1e-13
        assert h.smoothers[l].eigenvalue_estimates()[1] == pytest.approx(mg.max_eigenvalue(l), rel=1e-9)
    # one V-cycle
    r = rng.standard_normal(n)
    vr, vz = mgamd.Vector(ctx, n).from_host(r), mgamd.Vector(ctx, n)
    h.mg.vmult(vz, vr)
    tol_v = 1e-11 if coarse == "amg" else 1e-5
    assert rel_err(vz.to_host(), mg.vcycle(r)) < tol_v
    # full solve
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    xs, its, ress = mg.solve_cg(b.to_host(), 1e-4)
    assert it == its
    assert rel_err(x.to_host(), xs) < (1e-10 if coarse == "amg" else 1e-5)


@pytest.mark.parametrize("geo,L,p,typ", [("quadrant", 5, 4, "HMG-global"), ("annulus", 6, 2, "HMG-global"), ("quadrant", 6, 1, "HMG-global"),
                                         ("annulus", 5, 4, "PMG")])
def test_solve_matches_cpu_oracle_on_the_oracles_own_tables(mgamd, ctx, geo, L, p, typ):
    """The same comparison with NOTHING of the product's host setup on the checker's side: mesh, coarsening sequence, DoF
    classification, constraint masks, gather lists and transfer patches come from oracle/own_tables.py (pure Python on
    mgoracle's octree).  Only the DoF LABELS are the product's (its geometric key list per level), so that deal.II's
    numbering-dependent Chebyshev start vector (i mod 11) - mean is the same vector on both sides and the tolerances can stay at
    rounding level.  321 k DoFs at p = 4: 27 bricks of 17^3 lattice points with fused transfers next to constrained families
    and hanging single cells."""
    import cpu_oracle

    coarse = "amg" if typ == "HMG-global" else "cg_with_chebyshev"
    h = mgamd.Hierarchy(ctx, geo, L, p, typ, coarse_solver=coarse, max_brick=0)
    tabs, levels, transfers, mg = cpu_oracle.build_from_own_tables(geo, L, p, typ, coarse=coarse, numbering_keys=[d.keys() for d in h.dofs])
    for d, t in zip(h.dofs, tabs):
        assert t.n == d.n_dofs and t.first_constrained == d.info.n_interior + d.info.n_tail
        # (hanging nodes ON the boundary: own_tables files them under Dirichlet, the product under hanging -- constrained either way)
        assert int(t.dirichlet.sum()) >= d.info.n_dirichlet and t.n - t.first_constrained == d.info.n_dirichlet + d.info.n_hanging
    rng = np.random.default_rng(17)
    for l, op in enumerate(h.operators):
        x = rng.standard_normal(tabs[l].n)
        src, dst = op.initialize_dof_vector().from_host(x), op.initialize_dof_vector()
        op.vmult(dst, src)
        assert rel_err(dst.to_host(), levels[l].vmult(x)) < 1e-13
        assert h.smoothers[l].eigenvalue_estimates()[1] == pytest.approx(mg.max_eigenvalue(l), rel=1e-9)
    n = h.n_dofs
    r = rng.standard_normal(n)
    vr, vz = mgamd.Vector(ctx, n).from_host(r), mgamd.Vector(ctx, n)
    h.mg.vmult(vz, vr)
    assert rel_err(vz.to_host(), mg.vcycle(r)) < (1e-11 if coarse == "amg" else 1e-5)
    b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
    h.fine_operator.rhs(b)
    bo = tabs[-1].rhs_constant()
    bo[tabs[-1].first_constrained:] = 0.0
    assert rel_err(b.to_host(), bo) < 1e-13
    it, res = mgamd.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
    xs, its, ress = mg.solve_cg(bo, 1e-4)
    assert it == its
    assert rel_err(x.to_host(), xs) < (1e-10 if coarse == "amg" else 1e-5)
