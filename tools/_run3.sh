mkdir -p gpurun_out
export MGAMD_LIBRARY=$PWD/dealii_multigrid_amd/lib_debug/libmgamd.so MGAMD_NO_PIPELINE=1
for a in 0 2 1 16 3 18; do MGAMD_ABLATE=$a timeout -k 10 200 python tools/kernel_bench.py quadrant 8 4 >> gpurun_out/r2_ablate84.log 2>&1 || exit 1; done
cat gpurun_out/r2_ablate84.log
