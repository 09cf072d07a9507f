"""BASELINE.json configs[0]: the JSON-driven harness binary (dealii_multigrid_amd/bin/multigrid_throughput, built on the
deal.II-named C++ layer csrc/mgamd.hpp) run the way a user of the reference runs `multigrid_throughput input_*.json`
(ref:multigrid_throughput.cc:2437-2470): table columns in the reference's order, mesh/DoF/level counts of SURVEY App. B,
CG iteration counts of the oracle, and the reference's error behaviour (message + exit code 1)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "dealii_multigrid_amd", "bin", "multigrid_throughput")
GOLDEN = os.path.join(ROOT, "tests", "golden")
# ref:multigrid_throughput.cc:2328-2335 (dim..n_dofs), :1488 (sub_comm_size), :1278-1283 (n_levels..throughput),
# :1381-1394 (stage times), :1396-1401 (transfer times); this project's additions come after them
REFERENCE_COLUMNS = ["dim", "n_cells", "n_cells_hn", "n_cells_n", "degree", "n_ref_global", "n_ref_local", "n_dofs", "sub_comm_size",
                     "n_levels", "n_iterations", "time", "time_cg", "throughput", "time_pre", "time_residuum", "time_res", "time_cs",
                     "time_pro", "time_edge_pro", "time_post", "time_to_mg", "time_to_global"]


def run_harness(*files):
    r = subprocess.run([BIN, *files], capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout, r.stderr


def final_table(stdout):
    lines = stdout.rstrip().split("\n")
    start = max(i for i, l in enumerate(lines) if l.startswith("dim "))
    header = lines[start].split()
    return header, [dict(zip(header, l.split())) for l in lines[start + 1:] if l.strip()]


def oracle_iterations(oracle, geo, L, p, mg_type="HMG-global"):
    levels, P = oracle.build_hierarchy(geo, L, p, mg_type)
    mg = oracle.Multigrid(levels, P, 3, coarse="direct")
    return oracle.pcg(levels[-1].A, levels[-1].rhs_constant, mg.vcycle, 1e-4)[1]


def test_reference_generated_inputs(oracle):
    """input_0001.json / input_0003.json were written by the reference's own scripts/small-scaling.py (tests/golden/README.md):
    octant, NRefGlobal 3, p = 1 and p = 4, MGNumberType float, CoarseGridSolverType amg (one-cell coarse level: exact)."""
    assert os.path.exists(BIN), "build the harness with `make` (__graft_entry__.build())"
    rc, out, err = run_harness(os.path.join(GOLDEN, "input_0001.json"), os.path.join(GOLDEN, "input_0003.json"))
    assert rc == 0, err
    header, rows = final_table(out)
    assert header[:len(REFERENCE_COLUMNS)] == REFERENCE_COLUMNS
    # Verbosity true: the partition statistics of ref:include/mg_tools.h:316-317,377,443,491, then this project's additions
    assert header[len(REFERENCE_COLUMNS):] == ["workload_eff", "workload_path_max", "vertical_eff", "horizontal_eff", "mem_total",
                                               "dofs_per_s_per_vcycle", "coarse_solver"]
    assert len(rows) == 2
    r1, r4 = rows
    assert (int(r1["dim"]), int(r1["n_cells"]), int(r1["n_cells_hn"]), int(r1["n_dofs"]), int(r1["n_levels"])) == (3, 120, 37, 223, 4)
    assert (int(r4["n_cells"]), int(r4["degree"]), int(r4["n_dofs"]), int(r4["n_levels"])) == (120, 4, 9295, 4)
    assert int(r1["n_cells_n"]) == 120 - 37 and int(r1["sub_comm_size"]) == 1 and int(r1["n_ref_global"]) == 3
    # FP32 levels under the FP64 CG: iteration counts of the FP64 oracle +- 1 (as in test_float_levels_mixed_precision)
    assert abs(int(r1["n_iterations"]) - oracle_iterations(oracle, "quadrant", 3, 1)) <= 1
    assert abs(int(r4["n_iterations"]) - oracle_iterations(oracle, "quadrant", 3, 4)) <= 1
    for r in rows:
        t, its, n = float(r["time"]), int(r["n_iterations"]), int(r["n_dofs"])
        assert t > 0 and float(r["throughput"]) == pytest.approx(n * its / t, rel=2e-3)  # ref:multigrid_throughput.cc:1282
        stages = sum(float(r[c]) for c in REFERENCE_COLUMNS[14:])
        assert 0 < stages and float(r["time_cg"]) == pytest.approx((t - its * stages) / its, abs=2e-3 * t)
        assert float(r["time_edge_pro"]) < 0.2 * stages  # no edge matrices in global coarsening
        assert float(r["dofs_per_s_per_vcycle"]) == pytest.approx(n / stages, rel=2e-3)
        assert r["coarse_solver"] == "direct"
    assert float(r1["workload_eff"]) == 1.0 and float(r1["vertical_eff"]) == 1.0 and float(r1["horizontal_eff"]) == 1.0  # one rank
    assert float(r1["workload_path_max"]) == 1 + 8 + 15 + 120 and float(r1["mem_total"]) > 0
    assert "cells" in out and "dofs" in out  # Verbosity: the per-level table (ref:multigrid_throughput.cc:1644-1655)


def test_double_levels_match_oracle_iteration_counts(oracle, tmp_path):
    cases = [("quadrant", 3, 4, "HMG-global"), ("annulus", 5, 2, "PMG"), ("quadrant", 3, 4, "HPMG")]
    files = []
    base = json.load(open(os.path.join(GOLDEN, "input_0003.json")))
    for i, (geo, L, p, typ) in enumerate(cases):
        cfg = dict(base, Type=typ, GeometryType=geo, NRefGlobal=L, Degree=p, MGNumberType="double", Verbosity=False)
        files.append(str(tmp_path / f"in{i}.json"))
        json.dump(cfg, open(files[-1], "w"))
    rc, out, err = run_harness(*files)
    assert rc == 0, err
    header, rows = final_table(out)
    for r, (geo, L, p, typ) in zip(rows, cases):
        assert int(r["n_iterations"]) == oracle_iterations(oracle, geo, L, p, typ)
        assert r["coarse_solver"] == "direct"


def test_amg_on_a_large_coarse_level(oracle, tmp_path):
    """PMG with the reference's default CoarseGridSolverType "amg": the p = 1 coarse level of annulus L = 6 has 9,763 DoFs: the
    library's smoothed-aggregation AMG runs (coarse_solver column "amg").  With CoarseGridSolverType "gmg_vcycle" (this project's
    extension: the geometric stand-in of rounds 1-2) and one cycle the preconditioner is the HPMG V-cycle: iteration counts
    equal to the oracle's HPMG hierarchy; the algebraic solver needs at most one iteration more."""
    base = json.load(open(os.path.join(GOLDEN, "input_0003.json")))
    its = {}
    for coarse in ("amg", "gmg_vcycle"):
        cfg = dict(base, Type="PMG", GeometryType="annulus", NRefGlobal=6, Degree=2, MGNumberType="double", CoarseSolverNCycles=1,
                   CoarseGridSolverType=coarse)
        f = str(tmp_path / f"pmg_{coarse}.json")
        json.dump(cfg, open(f, "w"))
        rc, out, err = run_harness(f)
        assert rc == 0, err
        header, rows = final_table(out)
        assert rows[0]["coarse_solver"] == coarse and int(rows[0]["n_dofs"]) == 71509 and int(rows[0]["n_levels"]) == 2
        its[coarse] = int(rows[0]["n_iterations"])
    assert its["gmg_vcycle"] == oracle_iterations(oracle, "annulus", 6, 2, "HPMG")
    assert its["amg"] <= its["gmg_vcycle"] + 1


def test_local_smoothing_inputs(tmp_path):
    """input_0000.json / input_0002.json of scripts/small-scaling.py are the `HMG-local` siblings of the golden inputs: same
    mesh and DoF counts, n_levels = refinement levels, iteration counts of the local-smoothing oracle (float levels: +-1),
    and a non-zero edge-prolongation column (ref:multigrid_throughput.cc:1189-1190,1391)."""
    import ls_oracle

    files = []
    for name, p in (("input_0001.json", 1), ("input_0003.json", 4)):
        cfg = dict(json.load(open(os.path.join(GOLDEN, name))), Type="HMG-local")
        files.append(str(tmp_path / f"ls{p}.json"))
        json.dump(cfg, open(files[-1], "w"))
    rc, out, err = run_harness(*files)
    assert rc == 0, err
    header, rows = final_table(out)
    assert header[:len(REFERENCE_COLUMNS)] == REFERENCE_COLUMNS
    for r, p, n in zip(rows, (1, 4), (223, 9295)):
        assert (int(r["n_cells"]), int(r["n_dofs"]), int(r["n_levels"])) == (120, n, 4)
        assert abs(int(r["n_iterations"]) - ls_oracle.LocalSmoothing("quadrant", 3, p).solve(1e-4)[1]) <= 1
        assert float(r["time_edge_pro"]) > 0 and r["coarse_solver"] == "direct"


def test_hpmg_local_input(tmp_path):
    """`HPMG-local`: p-levels 1 -> 2 -> 4 on the active mesh over one local-smoothing V-cycle at p = 1"""
    import ls_oracle

    cfg = dict(json.load(open(os.path.join(GOLDEN, "input_0003.json"))), Type="HPMG-local", MGNumberType="double")
    f = str(tmp_path / "hpls.json")
    json.dump(cfg, open(f, "w"))
    rc, out, err = run_harness(f)
    assert rc == 0, err
    header, rows = final_table(out)
    r = rows[0]
    assert (int(r["n_dofs"]), int(r["n_levels"]), r["coarse_solver"]) == (9295, 3, "gmg_vcycle")
    assert int(r["n_iterations"]) == ls_oracle.PolynomialOverLocalSmoothing("quadrant", 3, 4).solve(1e-4)[1]


@pytest.mark.parametrize("key,value,message", [("Type", "AMGPETSc", "not implemented"), ("Type", "AMG", "not implemented"),
                                               ("GeometryType", "torus", "not implemented"), ("MGNumberType", "half", "not implemented"),
                                               ("CoarseGridSolverType", "lu", "not implemented")])
def test_error_behaviour(tmp_path, key, value, message):
    """AssertThrow -> message on stderr, exit code 1 (ref:multigrid_throughput.cc:2444-2468)"""
    cfg = dict(json.load(open(os.path.join(GOLDEN, "input_0001.json"))), **{key: value})
    f = str(tmp_path / "bad.json")
    json.dump(cfg, open(f, "w"))
    rc, out, err = run_harness(f)
    assert rc == 1 and message in err and "Aborting!" in err


def test_no_arguments():
    rc, out, err = run_harness()
    assert rc == 1 and "No .json parameter files" in out


@pytest.mark.parametrize("mg_type,coarse", [("HMG-global", "amg"), ("PMG", "cg_with_chebyshev"), ("PMG", "amg"), ("HPMG", "amg")])
def test_sharded_mode_with_one_rank_over_rccl(tmp_path, mg_type, coarse):
    """One process per GPU (ref:multigrid_throughput.cc:2403-2442 is an MPI program): MGAMD_HARNESS_SHARDED=1 runs the sharded code
    path -- RCCL communicator from the id file, Partition (two tiers), DoFHandler(partition, mesh, rank, degree),
    Operator::reinit(..., comm), the geometric stand-in on a sharded AMG coarse level -- with the one rank a one-GPU box can run (RCCL
    refuses two ranks on one device): same table as the plain run."""
    base = json.load(open(os.path.join(GOLDEN, "input_0003.json")))
    cfg = dict(base, Type=mg_type, GeometryType="annulus", NRefGlobal=6, Degree=2, MGNumberType="double", CoarseSolverNCycles=1,
               CoarseGridSolverType=coarse)
    f = str(tmp_path / "sharded.json")
    json.dump(cfg, open(f, "w"))
    rc, out, err = run_harness(f)
    assert rc == 0, err
    _, plain = final_table(out)
    env = dict(os.environ, MGAMD_HARNESS_SHARDED="1", MGAMD_RCCL_ID_FILE=str(tmp_path / "rccl_id"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([BIN, f], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    _, sharded = final_table(r.stdout)
    for key in ("n_cells", "n_dofs", "n_levels", "n_iterations", "sub_comm_size", "coarse_solver"):
        assert sharded[0][key] == plain[0][key], key
