// Communicators for the sharded path (one rank per GPU).
//   RcclComm : production backend, RCCL over xGMI: grouped ncclSend/ncclRecv for the halo exchange (point-to-point
//              links, every peer driven concurrently), ncclAllReduce for the replicated-level defect and for scalars.
//   SimComm  : in-process backend for tests on ONE GPU: every rank is a host thread with its own stream on the same
//              device; exchanges are device-to-device copies between the ranks' buffers around host barriers.
// Both implement the same blocking-on-stream interface, so the whole distributed driver is exercised by the simulator
// and only this thin layer differs in production.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace mgamd
{
  struct Comm
  {
    int n_ranks = 1, rank = 0;
    virtual ~Comm() = default;
    // for every peer j: send count_j = offsets[j+1]-offsets[j] elements from send+offsets[j], receive as many into
    // recv+offsets[j]; ordered on `stream`
    virtual void
    exchange(const void *send, void *recv, const std::vector<int> &peers, const std::vector<uint32_t> &offsets, size_t elem_size,
             hipStream_t stream) = 0;
    virtual void
    allreduce_sum(void *buf, size_t n, int number_type, hipStream_t stream) = 0; // in place, device buffer
    virtual double
    allreduce_sum_host(double v, hipStream_t stream) = 0;
  };

  // ---------------------------------------------------------------------------------------------- simulator
  struct SimGroup
  {
    int                     n = 1;
    std::mutex              m;
    std::condition_variable cv;
    int                     waiting = 0;
    unsigned long           generation = 0;
    struct Slot
    {
      const void                  *send = nullptr;
      const std::vector<int>      *peers = nullptr;
      const std::vector<uint32_t> *offsets = nullptr;
      double                       scalar = 0;
      std::vector<double>          host;
    };
    std::vector<Slot> slots;
    explicit SimGroup(int n)
      : n(n)
      , slots(n)
    {}
    void
    barrier()
    {
      std::unique_lock<std::mutex> lk(m);
      const unsigned long          g = generation;
      if (++waiting == n)
        {
          waiting = 0;
          ++generation;
          cv.notify_all();
        }
      else
        cv.wait(lk, [&] { return generation != g; });
    }
  };

  struct SimComm : Comm
  {
    std::shared_ptr<SimGroup> g;
    SimComm(std::shared_ptr<SimGroup> grp, int r)
      : g(std::move(grp))
    {
      n_ranks = g->n;
      rank    = r;
    }
    static void
    check(hipError_t e)
    {
      if (e != hipSuccess)
        throw std::runtime_error(std::string("SimComm: ") + hipGetErrorString(e));
    }
    void
    exchange(const void *send, void *recv, const std::vector<int> &peers, const std::vector<uint32_t> &offsets, size_t elem_size,
             hipStream_t stream) override
    {
      check(hipStreamSynchronize(stream)); // my send buffer is packed
      auto &me   = g->slots[rank];
      me.send    = send;
      me.peers   = &peers;
      me.offsets = &offsets;
      g->barrier();
      for (size_t j = 0; j < peers.size(); ++j)
        {
          const auto &other = g->slots[peers[j]];
          size_t      k     = 0;
          while (k < other.peers->size() && (*other.peers)[k] != rank)
            ++k;
          if (k == other.peers->size())
            throw std::runtime_error("SimComm: asymmetric halo plan");
          const size_t cnt = offsets[j + 1] - offsets[j];
          if (cnt != (size_t)((*other.offsets)[k + 1] - (*other.offsets)[k]))
            throw std::runtime_error("SimComm: halo segment sizes differ between the two sides");
          check(hipMemcpyAsync((char *)recv + (size_t)offsets[j] * elem_size, (const char *)other.send + (size_t)(*other.offsets)[k] * elem_size,
                               cnt * elem_size, hipMemcpyDeviceToDevice, stream));
        }
      check(hipStreamSynchronize(stream));
      g->barrier(); // nobody repacks before everybody has read
    }
    void
    allreduce_sum(void *buf, size_t n, int number_type, hipStream_t stream) override
    {
      auto &me = g->slots[rank];
      me.host.resize(n);
      if (number_type == 8)
        check(hipMemcpyAsync(me.host.data(), buf, n * 8, hipMemcpyDeviceToHost, stream));
      else
        {
          std::vector<float> f(n);
          check(hipMemcpyAsync(f.data(), buf, n * 4, hipMemcpyDeviceToHost, stream));
          check(hipStreamSynchronize(stream));
          for (size_t i = 0; i < n; ++i)
            me.host[i] = f[i];
        }
      check(hipStreamSynchronize(stream));
      g->barrier();
      std::vector<double> sum(n, 0.0);
      for (int r = 0; r < n_ranks; ++r)
        for (size_t i = 0; i < n; ++i)
          sum[i] += g->slots[r].host[i];
      g->barrier();
      if (number_type == 8)
        check(hipMemcpyAsync(buf, sum.data(), n * 8, hipMemcpyHostToDevice, stream));
      else
        {
          std::vector<float> f(sum.begin(), sum.end());
          check(hipMemcpyAsync(buf, f.data(), n * 4, hipMemcpyHostToDevice, stream));
          check(hipStreamSynchronize(stream));
        }
      check(hipStreamSynchronize(stream));
    }
    double
    allreduce_sum_host(double v, hipStream_t) override
    {
      g->slots[rank].scalar = v;
      g->barrier();
      double s = 0;
      for (int r = 0; r < n_ranks; ++r)
        s += g->slots[r].scalar;
      g->barrier();
      return s;
    }
  };

  // ---------------------------------------------------------------------------------------------- RCCL
  struct RcclComm : Comm
  {
    ncclComm_t comm = nullptr;
    double    *d_scalar = nullptr, *h_scalar = nullptr;
    static void
    check(ncclResult_t r, const char *what)
    {
      if (r != ncclSuccess)
        throw std::runtime_error(std::string("RCCL ") + what + ": " + ncclGetErrorString(r));
    }
    RcclComm(int n, int r, const ncclUniqueId &id)
    {
      n_ranks = n;
      rank    = r;
      check(ncclCommInitRank(&comm, n, id, r), "ncclCommInitRank");
      if (hipMalloc((void **)&d_scalar, sizeof(double)) != hipSuccess || hipHostMalloc((void **)&h_scalar, sizeof(double)) != hipSuccess)
        throw std::runtime_error("RcclComm: allocation failed");
    }
    ~RcclComm() override
    {
      if (comm)
        ncclCommDestroy(comm);
      (void)hipFree(d_scalar);
      (void)hipHostFree(h_scalar);
    }
    void
    exchange(const void *send, void *recv, const std::vector<int> &peers, const std::vector<uint32_t> &offsets, size_t elem_size,
             hipStream_t stream) override
    {
      if (peers.empty())
        return;
      check(ncclGroupStart(), "ncclGroupStart");
      for (size_t j = 0; j < peers.size(); ++j)
        {
          const size_t bytes = (size_t)(offsets[j + 1] - offsets[j]) * elem_size;
          check(ncclSend((const char *)send + (size_t)offsets[j] * elem_size, bytes, ncclChar, peers[j], comm, stream), "ncclSend");
          check(ncclRecv((char *)recv + (size_t)offsets[j] * elem_size, bytes, ncclChar, peers[j], comm, stream), "ncclRecv");
        }
      check(ncclGroupEnd(), "ncclGroupEnd");
    }
    void
    allreduce_sum(void *buf, size_t n, int number_type, hipStream_t stream) override
    {
      check(ncclAllReduce(buf, buf, n, number_type == 8 ? ncclDouble : ncclFloat, ncclSum, comm, stream), "ncclAllReduce");
    }
    double
    allreduce_sum_host(double v, hipStream_t stream) override
    {
      *h_scalar = v;
      if (hipMemcpyAsync(d_scalar, h_scalar, sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess)
        throw std::runtime_error("RcclComm: copy failed");
      check(ncclAllReduce(d_scalar, d_scalar, 1, ncclDouble, ncclSum, comm, stream), "ncclAllReduce");
      if (hipMemcpyAsync(h_scalar, d_scalar, sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
          hipStreamSynchronize(stream) != hipSuccess)
        throw std::runtime_error("RcclComm: copy failed");
      return *h_scalar;
    }
  };
  // ---------------------------------------------------------------------------------------------- rank subsets
  // A level that is cut into n_parts = base->n_ranks / group parts, each held by `group` consecutive ranks which all do the
  // part's work (Partition, two tiers).  To the level's operator, smoother and transfers this is an ordinary communicator of
  // n_parts ranks: rank = part, the halo exchange with part q goes to the member of q's group with MY position in the group,
  // and a sum over the parts is the base all-reduce divided by the group size (every part arrives `group` times; exact, the
  // group size is a power of two).  replica_sum adds up the DIFFERENT partial vectors the members of one group hold after
  // restricting from a level on which each of them has cells of its own (recursive doubling; a + b = b + a, so all members end
  // with the same bits).
  template <typename T>
  __global__ void
  subset_scale_kernel(T *__restrict__ x, T s, size_t n)
  {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      x[i] *= s;
  }
  template <typename T>
  __global__ void
  subset_add_kernel(T *__restrict__ x, const T *__restrict__ y, size_t n)
  {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      x[i] += y[i];
  }
  struct SubsetComm : Comm
  {
    std::shared_ptr<Comm> base;
    int                   group = 1, index = 0; // ranks per part, my position in my part's group
    void                 *tmp       = nullptr;
    size_t                tmp_bytes = 0;
    SubsetComm(std::shared_ptr<Comm> b, int g)
      : base(std::move(b))
      , group(g)
    {
      if (g < 1 || base->n_ranks % g != 0 || (g & (g - 1)) != 0)
        throw std::invalid_argument("SubsetComm: the group size must be a power of two that divides the number of ranks");
      n_ranks = base->n_ranks / g;
      rank    = base->rank / g;
      index   = base->rank % g;
    }
    ~SubsetComm() override
    {
      (void)hipFree(tmp);
    }
    static int
    grid_for(size_t n)
    {
      return (int)std::min<size_t>((n + 255) / 256, 4096);
    }
    void
    exchange(const void *send, void *recv, const std::vector<int> &peers, const std::vector<uint32_t> &offsets, size_t elem_size,
             hipStream_t stream) override
    {
      std::vector<int> ranks(peers.size());
      for (size_t j = 0; j < peers.size(); ++j)
        ranks[j] = peers[j] * group + index;
      base->exchange(send, recv, ranks, offsets, elem_size, stream);
    }
    void
    allreduce_sum(void *buf, size_t n, int number_type, hipStream_t stream) override
    {
      base->allreduce_sum(buf, n, number_type, stream);
      if (group > 1 && n > 0)
        {
          if (number_type == 8)
            hipLaunchKernelGGL(subset_scale_kernel<double>, grid_for(n), 256, 0, stream, (double *)buf, 1.0 / group, n);
          else
            hipLaunchKernelGGL(subset_scale_kernel<float>, grid_for(n), 256, 0, stream, (float *)buf, 1.0f / group, n);
        }
    }
    double
    allreduce_sum_host(double v, hipStream_t stream) override
    {
      return base->allreduce_sum_host(v, stream) / group;
    }
    // buf <- sum over the members of my group of their buf (n elements of number_type bytes each, device memory)
    void
    replica_sum(void *buf, size_t n, int number_type, hipStream_t stream)
    {
      if (group == 1 || n == 0)
        return;
      const size_t bytes = n * (size_t)number_type;
      if (bytes > tmp_bytes)
        {
          if (hipStreamSynchronize(stream) != hipSuccess)
            throw std::runtime_error("SubsetComm: synchronisation failed");
          (void)hipFree(tmp);
          tmp       = nullptr;
          tmp_bytes = 0;
          if (hipMalloc(&tmp, bytes) != hipSuccess)
            throw std::runtime_error("SubsetComm: allocation failed");
          tmp_bytes = bytes;
        }
      const std::vector<uint32_t> offsets{0u, (uint32_t)n};
      if (n > 0xFFFFFFFFull)
        throw std::runtime_error("SubsetComm: vector too long");
      for (int bit = 1; bit < group; bit <<= 1)
        {
          const std::vector<int> partner{base->rank ^ bit}; // same group: group-aligned blocks of a power-of-two size
          base->exchange(buf, tmp, partner, offsets, (size_t)number_type, stream);
          if (number_type == 8)
            hipLaunchKernelGGL(subset_add_kernel<double>, grid_for(n), 256, 0, stream, (double *)buf, (const double *)tmp, n);
          else
            hipLaunchKernelGGL(subset_add_kernel<float>, grid_for(n), 256, 0, stream, (float *)buf, (const float *)tmp, n);
        }
    }
  };
} // namespace mgamd

struct mgamd_comm
{
  std::shared_ptr<mgamd::Comm> comm;
};
struct mgamd_sim_group
{
  std::shared_ptr<mgamd::SimGroup> group;
};
