set -o pipefail
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for v in lib_A lib lib_exp; do
  MGAMD_LIBRARY=$R/dealii_multigrid_amd/$v/libmgamd.so python3 tools/perf_probe.py quadrant:8:4 hypercube:9:1 > $O/r3d_$v.txt 2>&1
done
grep -A1 "level  n_dofs" $O/r3d_lib_A.txt $O/r3d_lib.txt $O/r3d_lib_exp.txt | grep -v "^--"
grep "eager" $O/r3d_*.txt
