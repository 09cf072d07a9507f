"""RCCL communicator smoke test on ONE GPU (world size 1): unique id, ncclCommInitRank, ncclAllReduce through the
DistributedHierarchy code path (the grouped send/recv halo exchange needs >= 2 GPUs and is covered by the simulator
and gloo tests).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dealii_multigrid_amd as m
ctx = m.Context(0)
uid = m.Communicator.rccl_unique_id()
comm = m.Communicator.rccl(ctx, 1, 0, uid)
print("allreduce(3.5) over 1 rank:", comm.allreduce_sum(ctx, 3.5))
h = m.DistributedHierarchy(ctx, comm, "quadrant", 4, 2, min_root_dofs=0)
b, x = h.fine_operator.initialize_dof_vector(), h.fine_operator.initialize_dof_vector()
h.fine_operator.rhs(b)
print("n_dofs", h.n_dofs, "solve", m.solve_cg(h.fine_operator, h.mg, x, b, 1e-4))
