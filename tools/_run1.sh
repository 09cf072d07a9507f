set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pipelined or stage_timing" > gpurun_out/r2_t1.log 2>&1 || tail -30 gpurun_out/r2_t1.log
tail -3 gpurun_out/r2_t1.log
timeout -k 10 200 python tools/perf_probe.py quadrant:8:4 > gpurun_out/r2_p84_pipe.log 2>&1 && \
MGAMD_NO_PIPELINE=1 timeout -k 10 200 python tools/perf_probe.py quadrant:8:4 > gpurun_out/r2_p84_nopipe.log 2>&1 && \
timeout -k 10 200 python tools/perf_probe.py quadrant:9:1 > gpurun_out/r2_p91_pipe.log 2>&1 && \
MGAMD_NO_PIPELINE=1 timeout -k 10 200 python tools/perf_probe.py quadrant:9:1 > gpurun_out/r2_p91_nopipe.log 2>&1 && \
for mode in 2 4 0; do MGAMD_LIBRARY=$PWD/dealii_multigrid_amd/lib_debug/libmgamd.so MGAMD_NO_PIPELINE=1 MGAMD_STAMPS=$mode timeout -k 10 200 python tools/stamps.py quadrant 8 4 > gpurun_out/r2_stamps84_m$mode.log 2>&1 || exit 1; done
head -14 gpurun_out/r2_p84_pipe.log gpurun_out/r2_p84_nopipe.log gpurun_out/r2_p91_pipe.log gpurun_out/r2_p91_nopipe.log; cat gpurun_out/r2_stamps84_m*.log
