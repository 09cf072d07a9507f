#!/bin/bash
# BASELINE.json configs[4] (PMG p = 4 -> 2 -> 1 on the annulus) bench lines for profiles/: the single-GPU size (NRefGlobal 9) with the
# reference's default coarse solver, and the size quoted for 8 GPUs (NRefGlobal 8) with every coarse solver.
#   tools/profile_pmg.sh <tag>   -> gpurun_out/<tag>_bench_pmg_annulus*.json
set -o pipefail
tag=${1:-rXX}; O=$PWD/gpurun_out
timeout -k 10 500 python3 bench.py --workload pmg_annulus --no-cpu-baseline > $O/${tag}_bench_pmg_annulus9_p4_amg.json 2> $O/${tag}_pmg9.err || exit 1
for c in amg gmg_vcycle cg_with_chebyshev cg_with_amg; do
  timeout -k 10 200 python3 bench.py --workload pmg_annulus --nref 8 --coarse $c --no-cpu-baseline --no-secondary > $O/${tag}_bench_pmg_annulus8_p4_$c.json 2> $O/${tag}_pmg8_$c.err || exit 1
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${tag}_bench_pmg_annulus*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  %.3e DoF/s" % d["value"], "CG its", d["config"]["cg_iterations_reltol_1e-4"], d["config"]["workload"][:110])
PY
