mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t10.log 2>&1 || tail -60 gpurun_out/r2_t10.log
tail -3 gpurun_out/r2_t10.log
for v in new old; do
  if [ $v = old ]; then export MGAMD_MAX_CONSTRAINED_BRICK=2; fi
  timeout -k 10 200 python tools/perf_probe.py quadrant:8:4 > gpurun_out/r2_p84_rim_$v.log 2>&1 && \
  timeout -k 10 200 python tools/perf_probe.py quadrant:9:1 > gpurun_out/r2_p91_rim_$v.log 2>&1 || exit 1
  head -12 gpurun_out/r2_p84_rim_$v.log gpurun_out/r2_p91_rim_$v.log
done
