R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export MGAMD_MAX_CONSTRAINED_BRICK=2; fi
  rm -rf $R/gpurun_out/trace_$v
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$v -o t -- python3 $R/tools/vcycle_trace.py quadrant 8 4 3 > $R/gpurun_out/r2_trace_rim_$v.log 2>&1 || { tail -5 $R/gpurun_out/r2_trace_rim_$v.log; exit 1; }
  f=$(find $R/gpurun_out/trace_$v -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/vcycle_table.py $f 3 $R/gpurun_out/r2_vcycle_kernels_rim_$v.csv | head -24
  rm -rf $R/gpurun_out/trace_$v
done
