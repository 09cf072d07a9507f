"""Ownership plan of the fused level transfers (csrc/transfer_tables.hpp), checked on the host: a DoF owned by a fused 17-point
brick must not be referenced by any cell outside the fused set -- directly or through a hanging-node constraint -- because that
cell's residual contribution would reach a row of t that no un-fused patch restricts (and that cell would gather an x that the
prolongation has not corrected yet).  tools/fuse_plan_check.cpp walks every cell of every level of the hierarchy; the annulus at
NRefGlobal 8, p = 4 is the case that broke the first plan (bricks bordering constrained cells)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fuse_plan") / "fuse_plan_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "dealii_multigrid_amd", "csrc"),
                    os.path.join(ROOT, "tools", "fuse_plan_check.cpp"), "-o", exe], check=True, timeout=300)
    return exe


@pytest.mark.parametrize("geo,L,p", [("annulus", 8, 4), ("quadrant", 6, 4), ("annulus", 8, 2), ("quadrant", 7, 1), ("annulus", 9, 1)])
def test_fused_bricks_own_only_what_fused_bricks_touch(checker, geo, L, p):
    r = subprocess.run([checker, geo, str(L), str(p)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "fused bricks" in l]
    assert lines and any(int(l.split(":")[1].split()[0]) > 0 for l in lines), r.stdout  # the case has fused bricks at all
