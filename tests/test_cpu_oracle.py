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
