R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in "quadrant 8 4" "quadrant 9 1"; do
  tag=$(echo $w | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_${c}_$tag
    timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${c}_$tag -o p -- python3 $R/tools/vcycle_trace.py $w 3 > $R/gpurun_out/r2_pmc_${c}_$tag.log 2>&1 || { tail -5 $R/gpurun_out/r2_pmc_${c}_$tag.log; exit 1; }
  done
  f=$(find $R/gpurun_out/pmc_FETCH_SIZE_$tag -name "*counter_collection.csv" | head -1)
  g=$(find $R/gpurun_out/pmc_WRITE_SIZE_$tag -name "*counter_collection.csv" | head -1)
  head -2 $f
  python3 $R/tools/pmc_vcycle.py $f $g 3 $R/gpurun_out/r2_pmc_traffic_$tag.json
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE_$tag $R/gpurun_out/pmc_WRITE_SIZE_$tag
done
