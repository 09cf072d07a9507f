mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t7.log 2>&1 || tail -40 gpurun_out/r2_t7.log
tail -3 gpurun_out/r2_t7.log
timeout -k 10 300 python bench.py --workload pmg_annulus --no-cpu-baseline > gpurun_out/r2_bench_pmg2.json 2> gpurun_out/r2_bench_pmg2.err || tail -5 gpurun_out/r2_bench_pmg2.err
timeout -k 10 300 python bench.py --workload pmg_annulus --coarse amg --no-cpu-baseline > gpurun_out/r2_bench_pmg3.json 2> gpurun_out/r2_bench_pmg3.err || tail -5 gpurun_out/r2_bench_pmg3.err
python - <<'PY'
import json
for f in ("gpurun_out/r2_bench_pmg2.json","gpurun_out/r2_bench_pmg3.json"):
    d=json.load(open(f)); print(f, d["ms_per_step"], d["value"], d["config"]["workload"][:90], d["config"]["cg_iterations_reltol_1e-4"], d.get("stage_ms_per_level"))
PY
