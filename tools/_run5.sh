mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_distributed_sim.py tests/test_gpu_parity.py -m gpu -x -q -k "sharded or eight_ranks or amg_coarse" > gpurun_out/r2_t5.log 2>&1 || tail -40 gpurun_out/r2_t5.log
tail -3 gpurun_out/r2_t5.log
