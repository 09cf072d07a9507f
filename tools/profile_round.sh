#!/bin/bash
# Everything the round's profiles/ entries are made of, in one GPU call (run from the repository root on the GPU box):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{bench_octant8_p4.json, vcycle_kernels_*.csv, pmc_traffic_*.json, stage_tables.txt}
# rocprofv3: kernel trace and the two PMC passes are separate runs (FETCH_SIZE, WRITE_SIZE), each on tools/vcycle_trace.py.
set -o pipefail
tag=${1:-rXX}
R=$PWD
O=$R/gpurun_out
export TMPDIR=/tmp
trace() { # geometry nref degree label
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_t -o t -- python3 $R/tools/vcycle_trace.py $1 $2 $3 3 > $O/${tag}_trace_$4.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_f -o t -- python3 $R/tools/vcycle_trace.py $1 $2 $3 3 > $O/${tag}_fetch_$4.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_w -o t -- python3 $R/tools/vcycle_trace.py $1 $2 $3 3 > $O/${tag}_write_$4.log 2>&1 || return 1
  cd $R
  python3 tools/vcycle_table.py $(find $O/${tag}_t -name "*kernel_trace.csv") 3 $O/${tag}_vcycle_kernels_$4.csv > /dev/null || return 1
  python3 tools/pmc_vcycle.py $(find $O/${tag}_f -name "*counter_collection.csv") $(find $O/${tag}_w -name "*counter_collection.csv") 3 $O/${tag}_pmc_traffic_$4.json > $O/${tag}_pmc_$4.log || return 1
  rm -rf $O/${tag}_t $O/${tag}_f $O/${tag}_w
}
trace quadrant 8 4 octant8_p4 && trace quadrant 9 1 octant9_p1 && trace hypercube 9 1 uniform9_p1 || exit 1
timeout -k 10 300 python3 tools/perf_probe.py quadrant:8:4 quadrant:9:1 hypercube:9:1 hypercube:7:4 annulus:8:4 > $O/${tag}_stage_tables.txt 2>&1 || exit 1
timeout -k 10 900 python3 bench.py > $O/${tag}_bench_octant8_p4.json 2> $O/${tag}_bench.err || exit 1
# the same command under rocprofv3 --kernel-trace --stats (primary workload only): the kernel table whose average for the roofline
# kernel must agree with the bench line's HIP-event figure
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_bs -o b -- python3 $R/bench.py --steps 10 --no-float --no-secondary --no-cpu-baseline \
  > $O/${tag}_bench_octant8_p4_under_rocprof.json 2> $O/${tag}_bench_rocprof.err || exit 1
cd $R
cp $(find $O/${tag}_bs -name "*kernel_stats.csv" | head -1) $O/${tag}_bench_octant8_p4_kernel_stats.csv && rm -rf $O/${tag}_bs
tail -c 600 $O/${tag}_bench_octant8_p4.json
