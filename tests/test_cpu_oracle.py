"""The C++/OpenMP CPU oracle (quadrature-based, reference formulation) against the numpy/scipy assembled oracle."""
import numpy as np
import pytest

from conftest import rel_err


@pytest.fixture(scope="module")
def cpu_oracle():
    import cpu_oracle

    return cpu_oracle


@pytest.mark.parametrize("geo,L,p,typ", [("quadrant", 3, 1, "HMG-global"), ("quadrant", 3, 4, "HMG-global"), ("annulus", 5, 2, "HMG-global"),
                                         ("quadrant", 3, 4, "PMG"), ("hypercube", 3, 2, "HMG-global"), ("quadrant", 3, 3, "HMG-global")])
def test_cpu_oracle_matches_numpy_oracle(mgamd, oracle, cpu_oracle, geo, L, p, typ):
    fine = mgamd.Triangulation(geo, L)
    if typ == "PMG":
        degs = mgamd.create_polynomial_coarsening_sequence(p)
        trias = [fine] * len(degs)
    else:
        trias = mgamd.create_geometric_coarsening_sequence(fine)
        degs = [p] * len(trias)
    dofs = [mgamd.DoFs(t, d) for t, d in zip(trias, degs)]
    levels, transfers, mg = cpu_oracle.build_from_dofs(dofs, mgamd.transfer_tables)
    ol, P = oracle.build_hierarchy(geo, L, p, typ, numbering_keys=[d.keys() for d in dofs])
    omg = oracle.Multigrid(ol, P, 3)
    rng = np.random.default_rng(0)
    for l in range(len(dofs)):
        x = rng.standard_normal(dofs[l].n_dofs)
        assert rel_err(levels[l].vmult(x), ol[l].A @ x) < 1e-13
        assert rel_err(levels[l].inverse_diagonal(), ol[l].inv_diag) < 1e-12
        assert mg.max_eigenvalue(l) == pytest.approx(omg.sm[l].max_ev, rel=1e-10)
        if l > 0:
            xc = rng.standard_normal(dofs[l - 1].n_dofs)
            assert rel_err(transfers[l].prolongate_and_add(x, xc), x + P[l] @ xc) < 1e-13
            assert rel_err(transfers[l].restrict_and_add(xc, x), xc + P[l].T @ x) < 1e-13
    r = rng.standard_normal(dofs[-1].n_dofs)
    assert rel_err(mg.vcycle(r), omg.vcycle(r)) < 1e-12
    xs, it, res = mg.solve_cg(ol[-1].rhs_constant)
    xr, itr, hist = oracle.pcg(ol[-1].A, ol[-1].rhs_constant, omg.vcycle, 1e-4)
    assert it == itr and rel_err(xs, xr) < 1e-11


@pytest.mark.parametrize("geo,L,p", [("quadrant", 3, 2), ("annulus", 4, 1), ("quadrant", 3, 4), ("annulus", 5, 2)])
def test_own_tables_feed_the_cpp_oracle_independently_of_the_product(geo, L, p):
    """oracle/own_tables.py (mesh, numbering, masks, gather lists, transfer patches in pure Python on mgoracle's octree) +
    the C++ oracle's quadrature kernel against the numpy oracle's assembled C^T K C, P, Chebyshev and V-cycle: two
    formulations, no product code on either side."""
    import cpu_oracle
    import mgoracle as o

    tabs, levels, transfers, mg = cpu_oracle.build_from_own_tables(geo, L, p)
    keys = [np.array(t.keys) for t in tabs]
    lvls, P = o.build_hierarchy(geo, L, p, "HMG-global", numbering_keys=keys)
    rng = np.random.default_rng(2)
    for l, (cl, lv) in enumerate(zip(levels, lvls)):
        x = rng.standard_normal(lv.n)
        assert np.linalg.norm(cl.vmult(x) - lv.A @ x) <= 1e-13 * np.linalg.norm(lv.A @ x)
        assert np.abs(cl.inverse_diagonal() - lv.inv_diag).max() <= 1e-12 * np.abs(lv.inv_diag).max()
        if l > 0:
            xc, xf = rng.standard_normal(lvls[l - 1].n), rng.standard_normal(lv.n)
            assert np.linalg.norm(transfers[l].prolongate_and_add(xf, xc) - (xf + P[l] @ xc)) <= 1e-13 * np.linalg.norm(xf)
            assert np.linalg.norm(transfers[l].restrict_and_add(xc, xf) - (xc + P[l].T @ xf)) <= 1e-13 * np.linalg.norm(xc + P[l].T @ xf)
    omg = o.Multigrid(lvls, P, 3, coarse="direct")
    r = rng.standard_normal(lvls[-1].n)
    zr = omg.vcycle(r)
    assert np.linalg.norm(mg.vcycle(r) - zr) <= 1e-11 * np.linalg.norm(zr)
    b = tabs[-1].rhs_constant()
    b[tabs[-1].first_constrained:] = 0.0
    assert np.abs(b - lvls[-1].rhs_constant).max() <= 1e-14
    xs, its, _ = mg.solve_cg(b, 1e-4)
    xref, itref, _ = o.pcg(lvls[-1].A, lvls[-1].rhs_constant, omg.vcycle, 1e-4)
    assert its == itref and np.linalg.norm(xs - xref) <= 1e-10 * np.linalg.norm(xref)
