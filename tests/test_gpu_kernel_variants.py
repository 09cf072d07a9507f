"""The 17^3-lattice kernels exist in two forms (persistent workgroups with a software pipeline = default; one workgroup per
brick = MGAMD_NO_PERSISTENT=1), and tail_kernel reads D^-1 either through one-byte codes (default) or from the vector
(MGAMD_NO_DINV_CODES=1).  Round 3 added: level transfers fused into the brick kernel (default) or as separate kernels
(MGAMD_NO_FUSED_TRANSFER=1) and wave-scoped single cells (default) or workgroup-scoped ones (MGAMD_NO_CELL_WAVES=1).  Every other
GPU test runs the defaults; here the alternatives are checked against them on meshes with several 17^3 bricks (the switches are
read when the library is loaded, hence child processes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("quadrant", 7, 4), ("hypercube", 8, 1), ("hypercube", 4, 2)]  # the first two: several bricks per persistent workgroup


def run(geo, L, p, out, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    subprocess.run([sys.executable, os.path.join(HERE, "_vcycle_dump.py"), geo, str(L), str(p), out], check=True, env=env, timeout=600)
    return np.load(out)


@pytest.mark.parametrize("geo,L,p", CASES)
def test_alternative_kernel_paths_agree(tmp_path, geo, L, p):
    ref = run(geo, L, p, str(tmp_path / "default.npz"), {})
    assert any(g[1] > 1 and p * g[0] + 1 == 17 for g in ref["groups"]), "the case must contain several 17^3-lattice bricks"
    alt = run(geo, L, p, str(tmp_path / "alt.npz"), {"MGAMD_NO_PERSISTENT": "1", "MGAMD_NO_DINV_CODES": "1"})
    for key in ("ax", "step", "vcycle"):
        a, b = ref[key], alt[key]
        assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(a), key


@pytest.mark.parametrize("geo,L,p", [("quadrant", 6, 4), ("annulus", 6, 2), ("quadrant", 5, 3)])
def test_round3_kernel_paths_agree(tmp_path, geo, L, p):
    """hanging-node meshes with many single cells (and, at p = 4, 17^3 bricks with fused transfers): separate transfer kernels
    and workgroup-scoped cell kernels give the defaults' results to rounding"""
    ref = run(geo, L, p, str(tmp_path / "default.npz"), {})
    assert any(g[0] == 1 and g[1] > 8 for g in ref["groups"]), "the case must contain single-cell slots"
    alt = run(geo, L, p, str(tmp_path / "alt.npz"), {"MGAMD_NO_FUSED_TRANSFER": "1", "MGAMD_NO_CELL_WAVES": "1"})
    for key in ("ax", "step", "vcycle"):
        a, b = ref[key], alt[key]
        assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(a), key


@pytest.mark.parametrize("geo,L,p", [("quadrant", 6, 4)])
def test_constrained_rim_bricks_agree(tmp_path, geo, L, p):
    """MGAMD_MAX_CONSTRAINED_BRICK=4 (development switch): the 4^3 bricks next to coarser cells stay 17-point
    lattices with whole hanging faces / edges (persistent CONSTR kernel, hanging-node passes node by node over the faces:
    brick_face_passes) instead of eight families each -- same operator, smoother step and V-cycle to rounding.  The slot
    decomposition changes the DoF numbering, so inputs and results are matched through the geometric DoF keys."""
    def run_keys(out, env_extra):
        env = dict(os.environ, MGAMD_CHEB_KEY_INIT="1")  # numbering-independent Chebyshev start vector
        env.update(env_extra)
        subprocess.run([sys.executable, os.path.join(HERE, "_vcycle_dump.py"), geo, str(L), str(p), out, "keys"], check=True, env=env, timeout=600)
        return np.load(out)

    ref = run_keys(str(tmp_path / "default.npz"), {})
    alt = run_keys(str(tmp_path / "alt.npz"), {"MGAMD_MAX_CONSTRAINED_BRICK": "4"})
    assert sum(1 for g in alt["groups"] if g[0] == 4 and g[1] > 0) == 2 > sum(1 for g in ref["groups"] if g[0] == 4 and g[1] > 0)
    order = lambda d: np.lexsort(d["keys"].T[::-1])
    ro, ao = order(ref), order(alt)
    assert np.array_equal(ref["keys"][ro], alt["keys"][ao])
    for key in ("ax", "step", "vcycle"):
        a, b = ref[key][ro], alt[key][ao]
        assert np.linalg.norm(a - b) <= 1e-11 * np.linalg.norm(a), key
