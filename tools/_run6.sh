mkdir -p gpurun_out
(time timeout -k 10 900 python bench.py > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err) 2>&1 | tail -3
tail -3 gpurun_out/r2_bench_a.err
timeout -k 10 300 python bench.py --workload pmg_annulus --no-cpu-baseline > gpurun_out/r2_bench_pmg.json 2> gpurun_out/r2_bench_pmg.err || tail -5 gpurun_out/r2_bench_pmg.err
cat gpurun_out/r2_bench_a.json gpurun_out/r2_bench_pmg.json
