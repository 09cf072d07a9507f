#!/usr/bin/env python3
"""Generates tests/golden/oracle_solve.json from the numpy/scipy oracle (oracle/mgoracle.py), in the oracle's own DoF
numbering.  The reference has no recorded outputs (parity unpinned); these vectors pin the oracle against regressions and
give the iteration counts the GPU path and the C++ CPU oracle must reproduce."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mgoracle as mo

CASES = [("quadrant", 3, 1, "HMG-global"), ("quadrant", 3, 4, "HMG-global"), ("quadrant", 4, 1, "HMG-global"), ("hypercube", 3, 1, "HMG-global"),
         ("hypercube", 2, 4, "HMG-global"), ("annulus", 5, 2, "HMG-global"), ("quadrant", 3, 4, "PMG"), ("quadrant", 3, 2, "HMG-global")]
out = []
for geo, L, p, typ in CASES:
    r = mo.solve(geo, L, p, typ)
    lv = r["levels"]
    out.append({
        "geometry": geo, "n_ref_global": L, "degree": p, "type": typ,
        "n_cells": [len(l.cells) for l in lv], "n_dofs": [l.n for l in lv],
        "n_constrained": [int(l.constrained.sum()) for l in lv],
        "n_iterations": r["n_iterations"], "residual_history": [float(h) for h in r["history"]],
        "max_eigenvalue_estimates": [float(s.max_ev) for s in r["mg"].sm],
        "solution_l2": float(np.linalg.norm(r["x"])), "solution_sum": float(r["x"].sum()), "rhs_sum": float(lv[-1].rhs_constant.sum()),
    })
    print(out[-1]["geometry"], L, p, typ, out[-1]["n_iterations"], out[-1]["n_dofs"])
json.dump(out, open(os.path.join(HERE, "oracle_solve.json"), "w"), indent=1)

# local smoothing (oracle/ls_oracle.py): HMG-local and HPMG-local
import ls_oracle as lo

LS_CASES = [("quadrant", 3, 1, "HMG-local"), ("quadrant", 4, 2, "HMG-local"), ("annulus", 5, 1, "HMG-local"), ("quadrant", 3, 4, "HPMG-local")]
ls_out = []
for geo, L, p, typ in LS_CASES:
    s = lo.LocalSmoothing(geo, L, p) if typ == "HMG-local" else lo.PolynomialOverLocalSmoothing(geo, L, p)
    x, it, hist = s.solve(1e-4)
    ls = s if typ == "HMG-local" else s.ls
    ls_out.append({
        "geometry": geo, "n_ref_global": L, "degree": p, "type": typ,
        "level_n_dofs": [int(v.n) for v in ls.levels], "level_n_edge": [int(v.edge.sum()) for v in ls.levels],
        "level_n_copied": [int(len(g)) for g, _ in ls.copy],
        "n_iterations": it, "residual_history": [float(h) for h in hist],
        "max_eigenvalue_estimates": [float(sm.max_ev) for sm in ls.sm],
        "solution_l2": float(np.linalg.norm(x)),
    })
    print(geo, L, p, typ, it, ls_out[-1]["level_n_dofs"])
json.dump(ls_out, open(os.path.join(HERE, "oracle_local_smoothing.json"), "w"), indent=1)
