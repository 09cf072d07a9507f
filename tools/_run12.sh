mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t12.log 2>&1 || tail -60 gpurun_out/r2_t12.log
tail -3 gpurun_out/r2_t12.log
timeout -k 10 200 python tools/perf_probe.py quadrant:8:4 > gpurun_out/r2_p84_f.log 2>&1 && \
timeout -k 10 200 python tools/perf_probe.py quadrant:9:1 > gpurun_out/r2_p91_f.log 2>&1 && \
timeout -k 10 300 python tools/perf_probe.py hypercube:9:1 > gpurun_out/r2_h91_f.log 2>&1
head -3 gpurun_out/r2_p84_f.log gpurun_out/r2_p91_f.log gpurun_out/r2_h91_f.log
