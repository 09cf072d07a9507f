set -o pipefail
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_vs_cpu_oracle.py tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_harness_gpu.py -x -q > $O/r3f_tests.log 2>&1 || { tail -30 $O/r3f_tests.log; exit 1; }
tail -3 $O/r3f_tests.log
python3 tools/perf_probe.py quadrant:8:4 quadrant:5:4 annulus:8:4 > $O/r3f_waves.txt 2>&1
MGAMD_NO_CELL_WAVES=1 python3 tools/perf_probe.py quadrant:8:4 quadrant:5:4 annulus:8:4 > $O/r3f_nowaves.txt 2>&1
grep -A6 "level  n_dofs" $O/r3f_waves.txt $O/r3f_nowaves.txt | grep -v "^--"
grep "eager" $O/r3f_waves.txt $O/r3f_nowaves.txt
./tools/bin/crosslane_probe > $O/r3f_crosslane.txt 2>&1; cat $O/r3f_crosslane.txt
MGAMD_HARNESS_SHARDED=1 ./dealii_multigrid_amd/bin/multigrid_throughput tests/golden/input_0003.json > $O/r3f_harness_sharded.txt 2>&1; tail -5 $O/r3f_harness_sharded.txt
