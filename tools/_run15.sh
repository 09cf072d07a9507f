mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_local_smoothing.py tests/test_harness_gpu.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2_t15.log 2>&1 || tail -60 gpurun_out/r2_t15.log
tail -3 gpurun_out/r2_t15.log
timeout -k 10 200 python tools/perf_probe.py quadrant:8:4 > gpurun_out/r2_p84_g.log 2>&1 && head -5 gpurun_out/r2_p84_g.log
