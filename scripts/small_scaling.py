#!/usr/bin/env python3
"""Generator of the JSON input files of the small-scaling experiment: counterpart of the reference's
scripts/small-scaling.py + scripts/default.json (ref:scripts/small-scaling.py:7-55, ref:scripts/default.json:1-17).
Writes input_0000.json ... byte-identical to the reference's output (json.dump(..., indent=4, separators=(',', ': '))).

    python scripts/small_scaling.py <quadrant|annulus> [PartitionerName] [--max-ref N] [--out DIR]
"""
import argparse
import json
import os
from collections import OrderedDict

# ref:scripts/default.json (values keep their JSON types: several numbers are strings there)
DEFAULT = OrderedDict([
    ("Type", "HMG-global"), ("GeometryType", "quadrant"), ("NRefGlobal", "7"), ("NRefLocal", "0"), ("Degree", "3"),
    ("Paraview", False), ("Verbosity", True), ("PartitionerName", "CellWeightPolicy-2.0"), ("MinLevel", "0"), ("MinNCells", "0"),
    ("CoarseGridSolverType", "amg"), ("SmootherDegree", 3), ("CoarseSolverNCycles", 2), ("RelativeTolerance", 1e-4),
    ("MGNumberType", "float"),
])


def run_instance(out, counter, geometry_type, n_refinements, k, solver, partitioner, overrides=None):
    datastore = OrderedDict(DEFAULT)
    datastore["Type"] = solver
    datastore["GeometryType"] = geometry_type
    datastore["NRefGlobal"] = n_refinements
    datastore["Degree"] = k
    if partitioner != "":
        datastore["PartitionerName"] = partitioner
    if overrides:
        datastore.update(overrides)
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "input_%s.json" % str(counter).zfill(4)), "w") as f:
        json.dump(datastore, f, indent=4, separators=(",", ": "))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("geometry_type", choices=["quadrant", "annulus"])
    ap.add_argument("partitioner", nargs="?", default="")
    ap.add_argument("--max-ref", type=int, default=20, help="exclusive upper bound of NRefGlobal (reference: 20)")
    ap.add_argument("--out", default=".")
    ap.add_argument("--mg-number-type", default=None, help="override MGNumberType (reference default: float)")
    args = ap.parse_args()
    min_ref = 3 if args.geometry_type == "quadrant" else 5
    solvers = ["HMG-local", "HMG-global"] if args.partitioner == "" else ["HMG-global"]
    overrides = {"MGNumberType": args.mg_number_type} if args.mg_number_type else None
    counter = 0
    for n_refinements in range(min_ref, args.max_ref):
        for k in [1, 4]:
            for solver in solvers:
                run_instance(args.out, counter, args.geometry_type, n_refinements, k, solver, args.partitioner, overrides)
                counter += 1


if __name__ == "__main__":
    main()
