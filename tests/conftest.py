import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test that loops (a solver that stops converging) must fail, not sit on the GPU box until the box is reclaimed
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(420, method="thread"))


@pytest.fixture(scope="session")
def mgamd():
    import dealii_multigrid_amd as m

    return m


@pytest.fixture(scope="session")
def oracle():
    import mgoracle

    return mgoracle


@pytest.fixture(scope="session")
def ctx(mgamd):
    return mgamd.Context(0)


def oracle_level(oracle, dofs, geometry, n_ref, degree, mesh=None):
    """oracle Level numbered like the product (matched through the geometric DoF keys)."""
    if mesh is None:
        mesh = oracle.create_mesh(geometry, n_ref)
    return oracle.Level(mesh, degree, numbering_keys=dofs.keys())


def rel_err(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
