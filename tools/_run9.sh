mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_local_smoothing.py tests/test_harness_gpu.py -m gpu -x -q > gpurun_out/r2_t9.log 2>&1 || tail -60 gpurun_out/r2_t9.log
tail -3 gpurun_out/r2_t9.log
python scripts/small_scaling.py quadrant --out gpurun_out/inputs_q > /dev/null
ls gpurun_out/inputs_q | wc -l
timeout -k 10 600 ./dealii_multigrid_amd/bin/multigrid_throughput gpurun_out/inputs_q/input_000{0,1,2,3,4,5,6,7}.json gpurun_out/inputs_q/input_001{0,1,2,3,4,5}.json > gpurun_out/r2_harness_small_scaling.log 2>&1; echo rc=$?
tail -20 gpurun_out/r2_harness_small_scaling.log | cut -c1-250
