set -x
mkdir -p gpurun_out
true
true
cd /tmp && export TMPDIR=/tmp
for w in "quadrant 8 4" "quadrant 9 1"; do
  tag=$(echo $w | tr ' ' '_')
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -o t -- python3 $GRAFT_REPO_ROOT/tools/vcycle_trace.py $w 5 > $GRAFT_REPO_ROOT/gpurun_out/r2_trace_$tag.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/r2_trace_$tag.log; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -name "*kernel_trace.csv" | head -1)
  head -2 $f; python3 $GRAFT_REPO_ROOT/tools/vcycle_table.py $f 5 $GRAFT_REPO_ROOT/gpurun_out/r2_vcycle_kernels_$tag.csv
  grep "eager" $GRAFT_REPO_ROOT/gpurun_out/r2_trace_$tag.log
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$tag
done
