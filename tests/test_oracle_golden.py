"""The oracle reproduces its committed golden vectors (tests/golden/oracle_solve.json) and the product's host setup
reproduces the fixture's mesh / DoF counts."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_solve.json")))


@pytest.mark.parametrize("fx", FIX, ids=lambda f: f"{f['geometry']}-L{f['n_ref_global']}-p{f['degree']}-{f['type']}")
def test_oracle_matches_fixture(oracle, mgamd, fx):
    r = oracle.solve(fx["geometry"], fx["n_ref_global"], fx["degree"], fx["type"])
    assert r["n_iterations"] == fx["n_iterations"]
    assert np.allclose(r["history"], fx["residual_history"], rtol=1e-9)
    assert np.linalg.norm(r["x"]) == pytest.approx(fx["solution_l2"], rel=1e-10)
    assert np.allclose([s.max_ev for s in r["mg"].sm], fx["max_eigenvalue_estimates"], rtol=1e-10)
    # product host setup: same level sizes
    fine = mgamd.Triangulation(fx["geometry"], fx["n_ref_global"])
    if fx["type"] == "PMG":
        degs = mgamd.create_polynomial_coarsening_sequence(fx["degree"])
        trias = [fine] * len(degs)
    else:
        trias = mgamd.create_geometric_coarsening_sequence(fine)
        degs = [fx["degree"]] * len(trias)
    assert [t.n_cells for t in trias] == fx["n_cells"]
    dofs = [mgamd.DoFs(t, d) for t, d in zip(trias, degs)]
    assert [d.n_dofs for d in dofs] == fx["n_dofs"]
    assert [d.info.n_dirichlet + d.info.n_hanging for d in dofs] == fx["n_constrained"]
    assert dofs[-1].rhs_constant().sum() == pytest.approx(fx["rhs_sum"], rel=1e-12)


LS_FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_local_smoothing.json")))


@pytest.mark.parametrize("fx", LS_FIX, ids=lambda f: f"{f['geometry']}-L{f['n_ref_global']}-p{f['degree']}-{f['type']}")
def test_local_smoothing_oracle_matches_fixture(mgamd, fx):
    """oracle/ls_oracle.py reproduces its committed vectors, and the product's host tables the level sizes / edge counts"""
    import ls_oracle

    geo, L, p = fx["geometry"], fx["n_ref_global"], fx["degree"]
    s = ls_oracle.LocalSmoothing(geo, L, p) if fx["type"] == "HMG-local" else ls_oracle.PolynomialOverLocalSmoothing(geo, L, p)
    x, it, hist = s.solve(1e-4)
    assert it == fx["n_iterations"] and np.allclose(hist, fx["residual_history"], rtol=1e-9)
    assert np.linalg.norm(x) == pytest.approx(fx["solution_l2"], rel=1e-10)
    ls = s if fx["type"] == "HMG-local" else s.ls
    assert np.allclose([sm.max_ev for sm in ls.sm], fx["max_eigenvalue_estimates"], rtol=1e-10)
    fine = mgamd.Triangulation(geo, L)
    p_ls = p if fx["type"] == "HMG-local" else 1
    act = mgamd.DoFs(fine, p_ls, 0)
    for l in range(fine.n_levels):
        d = mgamd.DoFs(fine.level_mesh(l), p_ls, 0, local_smoothing_level=True)
        assert (d.n_dofs, d.info.n_edge) == (fx["level_n_dofs"][l], fx["level_n_edge"][l])
        assert len(mgamd.ls_copy_indices(act, d, l)[0]) == fx["level_n_copied"][l]
