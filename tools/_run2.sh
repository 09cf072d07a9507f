set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2_t2.log 2>&1 || tail -30 gpurun_out/r2_t2.log
tail -3 gpurun_out/r2_t2.log
MGAMD_NO_PIPELINE=1 timeout -k 10 200 python tools/perf_probe.py quadrant:8:4 > gpurun_out/r2_p84_seg.log 2>&1 && \
MGAMD_NO_PIPELINE=1 timeout -k 10 200 python tools/perf_probe.py quadrant:9:1 > gpurun_out/r2_p91_seg.log 2>&1 && \
MGAMD_NO_PIPELINE=1 timeout -k 10 200 python tools/perf_probe.py hypercube:9:1 > gpurun_out/r2_h91_seg.log 2>&1
head -14 gpurun_out/r2_p84_seg.log gpurun_out/r2_p91_seg.log gpurun_out/r2_h91_seg.log
