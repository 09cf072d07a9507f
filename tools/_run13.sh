R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
$R/tools/bin/mfma_probe 200 512 | tee $R/gpurun_out/r2_mfma_probe.log || echo "rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_mfma
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mfma -o m -- $R/tools/bin/mfma_probe 200 512 > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_mfma -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200; cp $f $R/gpurun_out/r2_mfma_probe_kernel_stats.csv; rm -rf $R/gpurun_out/prof_mfma
