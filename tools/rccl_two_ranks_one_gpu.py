"""Attempt to run the REAL RcclComm path (grouped ncclSend/ncclRecv halo exchange + all-reduces) with 2 ranks on ONE GPU
(development box): the unique id travels over gloo, both ranks create their RCCL communicator on device 0.
RCCL may refuse two ranks on one device ("Duplicate GPU detected"); then this prints the refusal and exits 0 -- the
grouped send/recv can only be exercised on >= 2 GPUs (driver's multi-GPU run).
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_two_ranks_one_gpu.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import dealii_multigrid_amd as m

ctx = m.Context(0)
uid = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    uid.copy_(torch.frombuffer(bytearray(m.Communicator.rccl_unique_id()), dtype=torch.uint8))
dist.broadcast(uid, 0)
try:
    comm = m.Communicator.rccl(ctx, world, rank, bytes(uid.numpy().tobytes()))
except m.MgamdError as e:
    print(f"rank {rank}: RCCL refused {world} ranks on one device: {e}", flush=True)
    dist.barrier()
    sys.exit(0)
print(f"rank {rank}: allreduce ->", comm.allreduce_sum(ctx, float(rank + 1)), flush=True)
h = m.DistributedHierarchy(ctx, comm, "quadrant", 5, 2, min_root_dofs=0)
b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
h.fine_operator.rhs(b)
it, res = m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4)
print(f"rank {rank}: n_dofs {h.n_dofs} local {h.n_local} peers {h.dofs[-1].info.n_peers} solve {it} {res:.3e}", flush=True)
dist.barrier()
