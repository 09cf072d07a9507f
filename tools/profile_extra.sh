#!/bin/bash
# Kernel tables of two more workloads (run from the repository root on the GPU box):  tools/profile_extra.sh <tag>
#   PMG annulus p=4 NRefGlobal 9 with the AMG coarse solver x2 (BASELINE.json configs[4] at the single-GPU size) and the octant p=4
#   headline workload with MGNumberType float
set -o pipefail
tag=${1:-rXX}
R=$PWD
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_tp -o t -- python3 $R/tools/vcycle_trace.py annulus 9 4 3 PMG amg 2 > $O/${tag}_trace_pmg.log 2>&1 || exit 1
MGAMD_TRACE_FLOAT=1 rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_tf -o t -- python3 $R/tools/vcycle_trace.py quadrant 8 4 3 > $O/${tag}_trace_float.log 2>&1 || exit 1
cd $R
python3 tools/vcycle_table.py $(find $O/${tag}_tp -name "*kernel_trace.csv") 3 $O/${tag}_vcycle_kernels_pmg_annulus9_p4_amg.csv > /dev/null || exit 1
python3 tools/vcycle_table.py $(find $O/${tag}_tf -name "*kernel_trace.csv") 3 $O/${tag}_vcycle_kernels_octant8_p4_float.csv > /dev/null || exit 1
rm -rf $O/${tag}_tp $O/${tag}_tf
tail -n 2 $O/${tag}_trace_pmg.log; tail -n 2 $O/${tag}_trace_float.log
