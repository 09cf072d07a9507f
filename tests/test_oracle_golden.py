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
