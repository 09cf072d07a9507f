R=$PWD; O=$R/gpurun_out
export MGAMD_LIBRARY=$R/dealii_multigrid_amd/lib_debug/libmgamd.so
for c in "quadrant 4 4 1" "quadrant 5 4 1" "quadrant 6 1 1"; do
  for mode in 0 2; do
    MGAMD_STAMPS=$mode python3 tools/stamps.py $c
  done
done > $O/r3e_stamps.txt 2>&1
cat $O/r3e_stamps.txt
