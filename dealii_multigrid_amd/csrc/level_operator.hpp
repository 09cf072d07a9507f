// The level operator (Operator<3,1,Number>, ref:include/operator.h:11-557) on the device: slot groups, tail accumulator,
// halo plan, and the launch logic of one operator application (apply_P).  Class template in a header so that the kernel
// instantiations behind apply_P<P, MODE> are compiled in their own translation units (apply_inst.hip, one per number type
// and degree, built in parallel); runtime.hip sees them through the explicit instantiation declarations at the end.
#pragma once
#include "runtime.hpp"
#include "kernels.hpp"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace mgamd
{
  inline int
  grid_for(size_t n)
  {
    size_t g = (n + 255) / 256;
    if (g > 2048)
      g = 2048;
    if (g < 1)
      g = 1;
    return (int)g;
  }

  template <typename T>
  inline double
  dot_raw(Ctx *ctx, const T *x, const T *y, size_t n)
  {
    if (!n)
      return 0.0;
    int g = grid_for(n);
    if (g > 1024)
      g = 1024;
    hipLaunchKernelGGL(vec_dot_kernel<T>, g, 256, 0, ctx->stream, x, y, n, ctx->d_partial);
    hipLaunchKernelGGL(vec_dot_final_kernel<>, 1, 256, 0, ctx->stream, ctx->d_partial, g, ctx->d_result);
    HIP_CHECK(hipMemcpyAsync(ctx->h_result, ctx->d_result, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    return ctx->h_result[0];
  }

  // ------------------------------------------------------------------------------------------
  // Level operator
  // ------------------------------------------------------------------------------------------
  // hipFuncAttributeMaxDynamicSharedMemorySize per (device, kernel), raised whenever a launch asks for more than the value
  // set so far; recorded only after the call has succeeded.  Callers may be concurrent host threads (SimComm).
  inline void
  ensure_dynamic_lds(Ctx *ctx, const void *kern, size_t lds)
  {
    static std::mutex                                      m;
    static std::map<std::pair<int, const void *>, size_t> done;
    std::lock_guard<std::mutex>                            lock(m);
    auto                                                   it = done.find({ctx->device, kern});
    if (it != done.end() && it->second >= lds)
      return;
    HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done[{ctx->device, kern}] = lds;
  }

  // persistent workgroups for the one-slot-per-workgroup lattices (kernels.hpp, lattice_apply_persistent_body); (until the
  // constants of the float kernels became floats, Mats<P, T>, float at p = 4 spilled and was excluded)
  template <typename T, int P>
  inline bool
  use_persistent()
  {
    static const bool on = getenv("MGAMD_NO_PERSISTENT") == nullptr;
    return on;
  }
  inline bool
  use_cell_waves()
  {
    static const bool on = getenv("MGAMD_NO_CELL_WAVES") == nullptr;
    return on;
  }
  // two 4-wave workgroups with a 17^3 lattice pair each fit one CU; a multiple of 8 keeps a workgroup in its XCD's range
  inline int
  resident_workgroups(const Ctx *ctx, int per_cu = 2)
  {
    return std::max(8, per_cu * ctx->n_cu / 8 * 8);
  }

  template <typename T, int P, int B, int MODE, bool CONSTR = false>
  inline void
  launch_lattice(Ctx *ctx, hipStream_t st, const ApplyArgs<T, P> &a, bool diag)
  {
    using G = Geo<P, B>;
    if (a.g.n_slots == 0)
      return;
    const int grid = (int)((a.g.n_slots + G::SPW - 1) / G::SPW);
    if (diag)
      {
        const size_t lds  = 3 * (size_t)G::SPW * G::N3 * sizeof(T);
        auto         kern = lattice_diag_kernel<T, P, B, CONSTR>;
        ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
        hipLaunchKernelGGL(kern, grid, G::BLOCK, lds, st, a);
      }
    else
      {
        const size_t lds  = (2 * (size_t)G::SPW * G::N3 + 2 * P * P * P + G::SPW + face_table_words<P, B>()) * sizeof(T);
        if constexpr (G::SPW == 1 && G::N_INT > 0 && G::ROUNDS > 1)
          {
            // one-slot-per-workgroup lattices (17^3): persistent workgroups with a software pipeline over their slots
            // (kernels.hpp, lattice_apply_persistent_body).  Two workgroups fit a CU (LDS); the grid is a multiple of 8 so
            // that a workgroup stays inside the Morton range of its XCD.  MGAMD_NO_PERSISTENT=1: one workgroup per slot.
            if (use_persistent<T, P>())
              {
                const int resident = resident_workgroups(ctx, persistent_wgs_per_cu<T, P>());
                auto      kern     = lattice_apply_persistent_kernel<T, P, B, MODE, CONSTR>;
                ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
                hipLaunchKernelGGL(kern, std::min(grid, resident), G::ABLOCK, lds, st, a);
                HIP_CHECK(hipGetLastError());
                return;
              }
          }
        auto         kern = lattice_apply_kernel<T, P, B, MODE, CONSTR>;
        ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
        hipLaunchKernelGGL(kern, grid, G::ABLOCK, lds, st, a);
      }
    HIP_CHECK(hipGetLastError());
  }

  // constrained = the group of constrained bricks larger than a family (p = 1 only, LevelTables::build)
  template <typename T, int P, int MODE>
  inline void
  dispatch_B(Ctx *ctx, hipStream_t st, int B, bool constrained, const ApplyArgs<T, P> &a, bool diag)
  {
    if (constrained)
      {
        if constexpr (P * 4 + 1 <= 17)
          if (B == 4)
            return launch_lattice<T, P, 4, MODE, true>(ctx, st, a, diag);
        if constexpr (P * 8 + 1 <= 17)
          if (B == 8)
            return launch_lattice<T, P, 8, MODE, true>(ctx, st, a, diag);
        if constexpr (P * 16 + 1 <= 17)
          if (B == 16)
            return launch_lattice<T, P, 16, MODE, true>(ctx, st, a, diag);
        throw std::runtime_error("constrained bricks of this size/degree are not instantiated");
      }
    switch (B)
      {
        case 1:
          if constexpr (P >= 2)
            if (!diag && use_cell_waves())
              { // single cells wave-scoped (kernels.hpp cell_waves_kernel); MGAMD_NO_CELL_WAVES=1: the workgroup-scoped kernel
                using GW = Geo<P, 1, 64>;
                if (a.g.n_slots == 0)
                  return;
                const uint32_t n_w  = (uint32_t)((a.g.n_slots + GW::SPW - 1) / GW::SPW);
                const uint32_t grid = (n_w + CELL_WAVES - 1) / CELL_WAVES;
                const size_t   lds  = CELL_WAVES * cell_wave_lds<T, P>();
                hipLaunchKernelGGL((cell_waves_kernel<T, P, MODE>), grid, 64 * CELL_WAVES, lds, st, a);
                HIP_CHECK(hipGetLastError());
                return;
              }
          launch_lattice<T, P, 1, MODE>(ctx, st, a, diag);
          return;
        case 2:
          if constexpr (P * 2 + 1 <= 17)
            {
              launch_lattice<T, P, 2, MODE>(ctx, st, a, diag);
              return;
            }
          break;
        case 4:
          if constexpr (P * 4 + 1 <= 17)
            {
              launch_lattice<T, P, 4, MODE>(ctx, st, a, diag);
              return;
            }
          break;
        case 8:
          if constexpr (P * 8 + 1 <= 17)
            {
              launch_lattice<T, P, 8, MODE>(ctx, st, a, diag);
              return;
            }
          break;
        case 16:
          if constexpr (P * 16 + 1 <= 17)
            {
              launch_lattice<T, P, 16, MODE>(ctx, st, a, diag);
              return;
            }
          break;
      }
    throw std::runtime_error("unsupported brick size");
  }

  template <typename T>
  struct GroupDev
  {
    int            B = 1, N = 2;
    bool           constrained = false; // the group of constrained bricks larger than a family
    size_t         n_halo      = 0;     // sharded levels: the first n_halo slots touch DoFs shared with other ranks
    size_t         n_slots = 0;
    DBuf<uint32_t> interior_base, shell_idx;
    DBuf<uint16_t> mask, shell_pos;
    DBuf<double>   h;
    DBuf<uint32_t> fmask; // constrained 2^3 bricks, only if the group has any
    SlotGroupDev
    view() const
    {
      return SlotGroupDev{interior_base.p, shell_idx.p, mask.p, h.p, shell_pos.p, (uint32_t)n_slots, fmask.p};
    }
    // slots [begin, end) only (sharded levels: halo slots first, then the rest)
    SlotGroupDev
    view(size_t begin, size_t end) const
    {
      const size_t n_shell = (size_t)N * N * N - (size_t)(N - 2) * (N - 2) * (N - 2);
      return SlotGroupDev{interior_base.p + begin, shell_idx.p + begin * n_shell, mask.p + begin, h.p + begin, shell_pos.p,
                          (uint32_t)(end - begin), fmask.p ? fmask.p + begin : nullptr};
    }
    // single cells at p = 1: per-cluster distinct node lists for cell_cluster_apply_kernel
    DBuf<uint32_t> uniq_ptr, uniq_idx;
    DBuf<uint16_t> loc;
    uint32_t       max_uniq = 0;
    bool
    has_clusters() const
    {
      return uniq_ptr.p != nullptr;
    }
    CellClusterDev
    cluster_view() const
    {
      return CellClusterDev{uniq_ptr.p, uniq_idx.p, loc.p, mask.p, h.p, (uint32_t)n_slots, max_uniq};
    }
    void
    build_clusters(const SlotGroup &g)
    {
      const size_t          ns = g.n_slots(), ncl = (ns + CLUSTER_CELLS - 1) / CLUSTER_CELLS;
      std::vector<uint32_t> ptr(ncl + 1, 0), idx;
      std::vector<uint16_t> l(ns * 8, 0xFFFFu);
      std::vector<uint32_t> tmp;
      for (size_t c = 0; c < ncl; ++c)
        {
          const size_t s0 = c * CLUSTER_CELLS, s1 = std::min(ns, s0 + CLUSTER_CELLS);
          tmp.clear();
          for (size_t sl = s0; sl < s1; ++sl)
            for (int s = 0; s < 8; ++s)
              if (g.shell_idx[sl * 8 + s] != INVALID_DOF)
                tmp.push_back(g.shell_idx[sl * 8 + s]);
          std::sort(tmp.begin(), tmp.end());
          tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
          for (size_t sl = s0; sl < s1; ++sl)
            for (int s = 0; s < 8; ++s)
              {
                const uint32_t gi = g.shell_idx[sl * 8 + s];
                if (gi != INVALID_DOF)
                  l[sl * 8 + g.shell_pos[s]] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), gi) - tmp.begin());
              }
          idx.insert(idx.end(), tmp.begin(), tmp.end());
          ptr[c + 1] = (uint32_t)idx.size();
          max_uniq   = std::max<uint32_t>(max_uniq, (uint32_t)tmp.size());
        }
      if (idx.empty())
        idx.push_back(0);
      uniq_ptr.upload(ptr);
      uniq_idx.upload(idx);
      loc.upload(l);
    }
  };

  // Host view of the level transfers fused into the operator passes of the FINE level (Transfer2 owns the tables; kernels.hpp,
  // FusedTransferDev): the bricks of slot group `group` flagged in `flags` restrict / prolongate inside the persistent kernel.
  template <typename T>
  struct FusedTransferHost
  {
    int                 group = -1;
    const uint32_t     *flags = nullptr;      // [n_slots of the group][256]
    const uint32_t     *coarse_idx = nullptr; // [n_slots of the group][nc3]
    uint32_t            nc3 = 0;
    std::vector<double> E;                    // 1D h-embedding (2p+1) x (p+1)
    const uint8_t      *tail_flags = nullptr; // [n_tail]: the tail DoF is owned by a fused brick
    // per call
    T *coarse = nullptr, *scratch = nullptr, *x_inout = nullptr;
  };
  // degrees whose 17-point lattice kernel carries the fused modes (p = 3 has 13-point lattices, one workgroup per brick)
  template <typename T>
  inline bool
  fused_transfer_supported(int p)
  {
    return (p == 1 && use_persistent<T, 1>()) || (p == 2 && use_persistent<T, 2>()) || (p == 4 && use_persistent<T, 4>());
  }

  template <typename T>
  struct LevelOperator : LevelOperatorBase
  {
    int                                       p = 1;
    std::vector<std::unique_ptr<GroupDev<T>>> groups;
    DBuf<T>                                   tail_acc;
    // D^-1 codes of the tail / constrained DoFs for tail_kernel (kernels.hpp, Epilogue::dinv_code), valid for the vector dinv_coded
    DBuf<uint8_t>                             dinv_code;
    DBuf<T>                                   dinv_table;
    const T                                  *dinv_coded = nullptr;
    int                                       prof_B = 0; // brick size whose CHEB launches are profiled
    uint32_t                                  ablate = 0; // debug: MGAMD_ABLATE
    DBuf<unsigned long long>                  stamps;     // debug: MGAMD_STAMPS=<mode>, 8 stamps per workgroup of the largest group
    int                                       stamp_mode = -1;
    bool                                      merge_small = true; // MGAMD_NO_MERGE_SMALL=1: separate launches (development A/B)
    bool                                      halo_overlap = true; // MGAMD_NO_HALO_OVERLAP=1: exchange after all slots, on the main queue

    // sharded runs: device image of the halo plan
    struct HaloDev
    {
      DBuf<uint32_t> pack_idx, sh_tail, sh_ptr;
      DBuf<int32_t>  sh_src, sh_owner_src;
      DBuf<T>        send, recv;
    };
    std::unique_ptr<HaloDev> halo;

    LevelOperator(Ctx *c, const mgamd_dofs *dofs, std::shared_ptr<Comm> cm)
    {
      ctx    = c;
      type   = (int)sizeof(T);
      tables = dofs->tables;
      tria   = dofs->tria;
      p      = tables->p;
      if (cm && dofs->halo)
        {
          comm      = cm;
          halo_plan = dofs->halo;
          halo      = std::make_unique<HaloDev>();
          halo->pack_idx.upload(halo_plan->pack_idx);
          halo->sh_tail.upload(halo_plan->sh_tail);
          halo->sh_ptr.upload(halo_plan->sh_ptr);
          halo->sh_src.upload(halo_plan->sh_src);
          halo->sh_owner_src.upload(halo_plan->sh_owner_src);
          halo->send.alloc(std::max<size_t>(halo_plan->pack_idx.size(), 1));
          halo->recv.alloc(std::max<size_t>(halo_plan->pack_idx.size(), 1));
        }
      if (p > 4)
        throw std::runtime_error("degrees above 4 are not instantiated in this build");
      size_t best = 0;
      for (const SlotGroup &g : tables->groups)
        {
          auto d     = std::make_unique<GroupDev<T>>();
          d->B           = g.B;
          d->N           = g.N;
          d->constrained = g.constrained_group;
          d->n_halo      = g.n_halo_slots;
          d->n_slots = g.n_slots();
          if (d->n_slots)
            {
              d->interior_base.upload(g.interior_base);
              d->shell_idx.upload(g.shell_idx);
              d->mask.upload(g.mask);
              d->h.upload(g.h);
              d->shell_pos.upload(g.shell_pos);
              if (std::any_of(g.fmask.begin(), g.fmask.end(), [](uint32_t m) { return m != 0; }))
                d->fmask.upload(g.fmask);
              if (p == 1 && g.B == 1 && !getenv("MGAMD_NO_CELL_CLUSTERS") && !tables->ls_level)
                d->build_clusters(g); // (the cluster tables bake in which nodes are constrained: not on local-smoothing levels)
            }
          const size_t work = d->n_slots * (size_t)g.N * g.N * g.N;
          if (work > best)
            {
              best   = work;
              prof_B = g.B;
            }
          groups.push_back(std::move(d));
        }
      if (const char *e = getenv("MGAMD_ABLATE"))
        ablate = (uint32_t)atoi(e);
      if (const char *e = getenv("MGAMD_STAMP_B")) // debug: stamp the group of this brick size instead of the largest one
        prof_B = atoi(e);
      merge_small = getenv("MGAMD_NO_MERGE_SMALL") == nullptr;
      halo_overlap = getenv("MGAMD_NO_HALO_OVERLAP") == nullptr;
      if (const char *e = getenv("MGAMD_STAMPS"))
        {
          stamp_mode = atoi(e);
          size_t nwg = 0;
          for (auto &g : groups)
            if (g->B == prof_B)
              nwg = g->n_slots; // >= number of workgroups
          stamps.alloc(nwg * 8 + 8);
          stamps.zero(ctx->stream);
        }
      tail_acc.alloc(std::max<uint32_t>(tables->n_tail + tables->n_edge, 1));
      tail_acc.zero(ctx->stream);
    }

    // tail[t] <- sum over the sharing ranks (ascending rank order) of their partial tail[t]
    void
    exchange_add_raw(T *tail, hipStream_t st = nullptr)
    {
      if (!halo)
        return;
      if (!st)
        st = ctx->stream;
      const uint32_t ns = (uint32_t)halo_plan->pack_idx.size();
      if (ns)
        hipLaunchKernelGGL(halo_pack_kernel<T>, grid_for(ns), 256, 0, st, halo->send.p, tail, halo->pack_idx.p, ns);
      comm->exchange(halo->send.p, halo->recv.p, halo_plan->peers, halo_plan->peer_offset, sizeof(T), st);
      const uint32_t nsh = (uint32_t)halo_plan->sh_tail.size();
      if (nsh)
        hipLaunchKernelGGL(halo_combine_kernel<T>, grid_for(nsh), 256, 0, st, tail, halo->recv.p, halo->sh_tail.p, halo->sh_ptr.p,
                           halo->sh_src.p, nsh);
      HIP_CHECK(hipGetLastError());
    }
    // tail[t] <- the owner's value, for the copies this rank holds of DoFs owned elsewhere
    void
    import_from_owner_raw(T *tail)
    {
      if (!halo)
        return;
      const uint32_t ns = (uint32_t)halo_plan->pack_idx.size();
      if (ns)
        hipLaunchKernelGGL(halo_pack_kernel<T>, grid_for(ns), 256, 0, ctx->stream, halo->send.p, tail, halo->pack_idx.p, ns);
      comm->exchange(halo->send.p, halo->recv.p, halo_plan->peers, halo_plan->peer_offset, sizeof(T), ctx->stream);
      const uint32_t nsh = (uint32_t)halo_plan->sh_tail.size();
      if (nsh)
        hipLaunchKernelGGL(halo_import_kernel<T>, grid_for(nsh), 256, 0, ctx->stream, tail, halo->recv.p, halo->sh_tail.p,
                           halo->sh_owner_src.p, nsh);
      HIP_CHECK(hipGetLastError());
    }
    void
    exchange_add_tail(mgamd_vec &v) override
    {
      if (v.n != n_dofs())
        throw std::invalid_argument("exchange_add_tail: vector size mismatch");
      exchange_add_raw(v.as<T>() + tables->n_interior);
    }
    double
    dot_raw_global(const T *x, const T *y)
    {
      if (!comm)
        return dot_raw(ctx, x, y, (size_t)n_dofs());
      // constrained entries are zero in every vector of the sharded solver path (homogeneous data), so the owned
      // prefix [interior | owned tail] counts every DoF exactly once
      const double local = dot_raw(ctx, x, y, (size_t)tables->n_interior + tables->n_tail_owned);
      return comm->allreduce_sum_host(local, ctx->stream);
    }
    double
    dot(const mgamd_vec &x, const mgamd_vec &y) override
    {
      if (x.n != n_dofs() || y.n != n_dofs())
        throw std::invalid_argument("dot: vector size mismatch");
      return dot_raw_global(x.as<T>(), y.as<T>());
    }

    template <int P>
    Mats<P, T>
    mats() const
    {
      Mats<P, T> m;
      const int n = P + 1;
      for (int i = 0; i < n * n; ++i)
        {
          m.M[i]  = tables->fe.M[i];
          m.K[i]  = tables->fe.K[i];
          m.I0[i] = tables->fe.I[0][i];
          m.I1[i] = tables->fe.I[1][i];
        }
      constexpr int NH = Mats<P>::NH, NO = Mats<P>::NO;
      auto          eo = [&](const double *A, T *Ae, T *Ao) {
        for (int i = 0; i < NH; ++i)
          for (int j = 0; j < NH; ++j)
            Ae[i * NH + j] = (j < NO) ? 0.5 * (A[i * n + j] + A[i * n + P - j]) : A[i * n + j];
        for (int i = 0; i < NO; ++i)
          for (int j = 0; j < NO; ++j)
            Ao[i * NO + j] = 0.5 * (A[i * n + j] - A[i * n + P - j]);
      };
      eo(tables->fe.M.data(), m.Me, m.Mo);
      eo(tables->fe.K.data(), m.Ke, m.Ko);
      return m;
    }

    ClusterArgs<T>
    cluster_args(const GroupDev<T> &g, const T *src, const Epilogue<T> &epi, bool first)
    {
      ClusterArgs<T> a;
      a.c          = g.cluster_view();
      a.m          = mats<1>();
      a.src        = src;
      a.tail_acc   = tail_acc.p;
      a.n_interior = tables->n_interior;
      a.b          = epi.b;
      a.dinv       = epi.dinv;
      a.c0         = epi.c0;
      a.from_b     = first ? 1 : 0;
      a.cluster_offset = 0;
      return a;
    }

    void
    launch_clusters(hipStream_t st, const GroupDev<T> &g, const T *src, const Epilogue<T> &epi, bool first, uint32_t cluster_begin = 0,
                    uint32_t cluster_end = 0xFFFFFFFFu)
    {
      ClusterArgs<T> a;
      a.c          = g.cluster_view();
      a.m          = mats<1>();
      a.src        = src;
      a.tail_acc   = tail_acc.p;
      a.n_interior = tables->n_interior;
      a.b          = epi.b;
      a.dinv       = epi.dinv;
      a.c0         = epi.c0;
      a.from_b     = first ? 1 : 0;
      a.cluster_offset = 0;
      const uint32_t n_cl = (uint32_t)((g.n_slots + CLUSTER_CELLS - 1) / CLUSTER_CELLS);
      cluster_end         = std::min(cluster_end, n_cl);
      if (cluster_begin >= cluster_end)
        return;
      a.cluster_offset    = cluster_begin;
      const uint32_t grid = cluster_end - cluster_begin;
      hipLaunchKernelGGL(cell_cluster_apply_kernel<T>, grid, CLUSTER_CELLS, 2 * (size_t)std::max<uint32_t>(g.max_uniq, 1) * sizeof(T), st, a);
      HIP_CHECK(hipGetLastError());
    }

    // launches of one slot group (with its merge partner, if any) on stream st; [begin, end) restricts an unmerged group
    // to a slot range
    template <int P, int B, int MODE>
    void
    launch_pair(hipStream_t st, const ApplyArgs<T, P> &a, GroupDev<T> *g_constrained)
    {
      using G = Geo<P, B>;
      BrickPairArgs<T, P> pa;
      pa.a             = a;
      pa.g_constrained = g_constrained->view();
      pa.n_wg_plain    = (uint32_t)((a.g.n_slots + G::SPW - 1) / G::SPW);
      const uint32_t n_wg_c = (uint32_t)((g_constrained->n_slots + G::SPW - 1) / G::SPW);
      const size_t   lds    = (2 * (size_t)G::SPW * G::N3 + 2 * P * P * P + G::SPW + face_table_words<P, B>()) * sizeof(T);
      if constexpr (G::SPW == 1 && G::N_INT > 0 && G::ROUNDS > 1)
        if (use_persistent<T, P>())
          {
            const int resident = resident_workgroups(ctx);
            uint32_t  np = pa.n_wg_plain, nc = n_wg_c;
            if ((int)(np + nc) > resident)
              { // every workgroup walks both kinds (kernels.hpp)
                np = 0;
                nc = (uint32_t)resident;
              }
            pa.n_wg_plain = np;
            auto kern     = lattice_apply_persistent_pair_kernel<T, P, B, MODE>;
            ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
            hipLaunchKernelGGL(kern, np + nc, G::ABLOCK, lds, st, pa);
            HIP_CHECK(hipGetLastError());
            return;
          }
      if constexpr (MODE != base_mode(MODE))
        throw std::runtime_error("fused transfers need the persistent brick kernel");
      else
        {
          auto kern = lattice_apply_pair_kernel<T, P, B, MODE>;
          ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
          hipLaunchKernelGGL(kern, pa.n_wg_plain + n_wg_c, G::ABLOCK, lds, st, pa);
          HIP_CHECK(hipGetLastError());
        }
    }

    // the 17-point lattice group that carries fused transfers (slots [begin, begin + a.g.n_slots) of it)
    template <int P, int MODE>
    void
    launch_fused(hipStream_t st, ApplyArgs<T, P> &a, size_t begin, GroupDev<T> *partner_constrained, const FusedTransferHost<T> &f)
    {
      if constexpr (P == 1 || P == 2 || P == 4)
        {
          constexpr int B = 16 / P;
          using G         = Geo<P, B>;
          constexpr int NC = P * B / 2 + 1, NC3 = NC * NC * NC;
          if (f.nc3 != (uint32_t)NC3 || f.E.size() != (size_t)(2 * P + 1) * (P + 1) || !use_persistent<T, P>())
            throw std::runtime_error("fused transfer: tables do not match the brick kernel");
          a.fused.flags      = f.flags + begin * G::ABLOCK;
          a.fused.coarse_idx = f.coarse_idx + begin * NC3;
          for (int i = 0; i < (P + 1) * (P + 1); ++i) // rows 0..P; the kernel uses E[2P - a][P - b] = E[a][b] for the rest
            a.fused.Eh[i] = f.E[i];
          for (int ar = 0; ar <= P; ++ar)
            for (int b = 0; b <= P; ++b)
              if (std::fabs(f.E[(2 * P - ar) * (P + 1) + (P - b)] - f.E[ar * (P + 1) + b]) > 1e-14)
                throw std::runtime_error("fused transfer: the 1D embedding is not centro-symmetric");
          a.fused.coarse  = f.coarse;
          a.fused.x_inout = f.x_inout;
          a.fused.scratch = f.scratch;
          if (partner_constrained)
            {
              if constexpr (P == 1)
                return launch_pair<P, B, MODE>(st, a, partner_constrained);
              throw std::runtime_error("brick pair launch: size not instantiated");
            }
          const size_t lds      = (2 * (size_t)G::N3 + 2 * P * P * P + 1 + face_table_words<P, B>()) * sizeof(T);
          const int    resident = resident_workgroups(ctx, persistent_wgs_per_cu<T, P>());
          auto         kern     = lattice_apply_persistent_kernel<T, P, B, MODE>;
          ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), lds);
          hipLaunchKernelGGL(kern, std::min((int)a.g.n_slots, resident), G::ABLOCK, lds, st, a);
          HIP_CHECK(hipGetLastError());
        }
      else
        throw std::runtime_error("fused transfers: degree without 17-point lattice kernel");
    }

    template <int P, int MODE_>
    void
    launch_group(hipStream_t st, ApplyArgs<T, P> &a, GroupDev<T> *g, GroupDev<T> *partner_cells, GroupDev<T> *partner_clusters,
                 const T *src, const Epilogue<T> &epi, bool diag, size_t begin, size_t end, GroupDev<T> *partner_constrained = nullptr,
                 const FusedTransferHost<T> *fused = nullptr)
    {
      constexpr int MODE = base_mode(MODE_); // every group but the fused one runs the base mode of a fused pass
      a.g      = (begin == 0 && end == g->n_slots) ? g->view() : g->view(begin, end);
      if constexpr (MODE_ != MODE)
        if (fused && fused->group >= 0 && g == groups[fused->group].get())
          return launch_fused<P, MODE_>(st, a, begin, partner_constrained, *fused);
      if (partner_constrained)
        {
          if constexpr (P == 1)
            {
              if (g->B == 16)
                return launch_pair<P, 16, MODE>(st, a, partner_constrained);
              if (g->B == 8)
                return launch_pair<P, 8, MODE>(st, a, partner_constrained);
            }
          throw std::runtime_error("brick pair launch: size not instantiated");
        }
      a.stamps = (stamps.p && g->B == prof_B && !g->constrained && MODE == stamp_mode && !diag && begin == 0) ? stamps.p : nullptr;
      if (partner_clusters)
        {
          if constexpr (P == 1)
            {
              using G8 = Geo<1, 8>;
              GroupDev<T>   *gc = partner_clusters;
              P1SmallArgs<T> sa;
              sa.a           = a;
              sa.c           = cluster_args(*gc, src, epi, MODE == MODE_CHEB_FIRST);
              sa.n_wg_bricks = (uint32_t)((g->n_slots + G8::SPW - 1) / G8::SPW);
              const uint32_t n_wg_cl = (uint32_t)((gc->n_slots + CLUSTER_CELLS - 1) / CLUSTER_CELLS);
              const size_t   lds     = std::max((2 * (size_t)G8::SPW * G8::N3 + 2 + G8::SPW) * sizeof(T),
                                          2 * (size_t)std::max<uint32_t>(gc->max_uniq, 1) * sizeof(T));
              hipLaunchKernelGGL((lattice_cluster_kernel<T, MODE>), sa.n_wg_bricks + n_wg_cl, 256, lds, st, sa);
              HIP_CHECK(hipGetLastError());
            }
        }
      else if (P == 1 && !diag && g->has_clusters())
        launch_clusters(st, *g, src, epi, MODE == MODE_CHEB_FIRST);
      else if (partner_cells)
        {
          if constexpr (P >= 2)
            {
              using G2 = Geo<P, 2>;
              using G1 = Geo<P, 1>;
              GroupDev<T>         *g1 = partner_cells;
              SmallSlotsArgs<T, P> sa;
              sa.a           = a;
              sa.g_cells     = g1->view();
              sa.n_wg_bricks = (uint32_t)((g->n_slots + G2::SPW - 1) / G2::SPW);
              // the cells wave-scoped: four wavefronts per workgroup with their own cells (kernels.hpp cell_waves_body)
              using GW = Geo<P, 1, 64>;
              const uint32_t n_w        = (uint32_t)((g1->n_slots + GW::SPW - 1) / GW::SPW);
              const uint32_t n_wg_cells = (n_w + CELL_WAVES - 1) / CELL_WAVES;
              const size_t   lds        = std::max((2 * (size_t)G2::SPW * G2::N3 + 2 * P * P * P + G2::SPW) * sizeof(T), CELL_WAVES * cell_wave_lds<T, P>());
              (void)sizeof(G1);
              hipLaunchKernelGGL((lattice_apply_small_kernel<T, P, MODE>), sa.n_wg_bricks + n_wg_cells, 256, lds, st, sa);
              HIP_CHECK(hipGetLastError());
            }
        }
      else
        dispatch_B<T, P, MODE>(ctx, st, g->B, g->constrained, a, diag);
    }

    template <int MODE>
    void
    launch_tail(hipStream_t st, uint32_t begin, uint32_t end, bool with_rest, const Epilogue<T> &epi_in, bool diag)
    {
      const uint32_t n_rest = with_rest ? tables->n_dofs - tables->n_interior - end : 0;
      const uint32_t n_t    = end - begin + n_rest;
      if (!n_t)
        return;
      Epilogue<T> epi = epi_in;
      if (epi.dinv_code)
        epi.dinv_code += begin;
      if (diag)
        hipLaunchKernelGGL((tail_kernel<T, MODE_INVDIAG>), grid_for(n_t), 256, 0, st, tail_acc.p + begin, tables->n_interior + begin, end - begin,
                           n_rest, epi);
      else
        hipLaunchKernelGGL((tail_kernel<T, MODE>), grid_for(n_t), 256, 0, st, tail_acc.p + begin, tables->n_interior + begin, end - begin, n_rest,
                           epi);
      HIP_CHECK(hipGetLastError());
    }

    // One operator application: the slot groups of the level (merged launches where two groups are a fraction of one round of
    // workgroups each), then the epilogue of the tail and the constrained DoFs.
    // Refinement-edge DoFs of a local-smoothing level (LevelTables::n_edge, numbered right after the tail):
    //   EDGE_OUT   the level operator (Operator::vmult, ref:include/operator.h:152-183): zero input, identity rows
    //   EDGE_ROWS  zero input, but their ROWS are computed: the residual that is restricted (deal.II edge_out /
    //              vmult_interface_down)
    //   EDGE_IN    ordinary unconstrained DoFs: the edge matrix (vmult_interface_up, ref:include/operator.h:203-226)
    enum EdgeMode
    {
      EDGE_OUT  = 0,
      EDGE_ROWS = 1,
      EDGE_IN   = 2
    };

    // defined in level_operator_apply.hpp, instantiated in apply_inst.hip (one translation unit per number type and degree)
    template <int P, int MODE>
    void
    apply_P(const T *src, const Epilogue<T> &epi, bool diag, double words, int edge_mode, const FusedTransferHost<T> *fused);

    template <int MODE>
    void
    apply(const T *src, const Epilogue<T> &epi, bool diag = false, double words = 0, int edge_mode = EDGE_OUT,
          const FusedTransferHost<T> *fused = nullptr)
    {
      constexpr bool FUSED_MODE = MODE != base_mode(MODE);
      if (FUSED_MODE && (!fused || fused->group < 0 || !fused_transfer_supported<T>(p)))
        throw std::runtime_error("fused transfer pass without fused tables");
      switch (p)
        {
          case 1:
            apply_P<1, MODE>(src, epi, diag, words, edge_mode, fused);
            break;
          case 2:
            apply_P<2, MODE>(src, epi, diag, words, edge_mode, fused);
            break;
          case 3:
            if constexpr (!FUSED_MODE)
              apply_P<3, MODE>(src, epi, diag, words, edge_mode, fused);
            break;
          case 4:
            apply_P<4, MODE>(src, epi, diag, words, edge_mode, fused);
            break;
          default:
            throw std::runtime_error("degree not instantiated");
        }
    }

    // raw-pointer entry points used by smoother / multigrid
    void
    vmult_raw(T *dst, const T *src)
    {
      Epilogue<T> e{dst, src, nullptr, nullptr, nullptr, T(0), T(0), T(0)};
      apply<MODE_VMULT>(src, e);
    }
    // t = b - A x, the residual step of Multigrid::level_v_step.  The reference hands Multigrid an mg::Matrix built from
    // MatrixFreeOperators::MGInterfaceOperator<LevelMatrixType> (ref:multigrid_throughput.cc:857-862), whose vmult calls
    // Operator::vmult_interface_down (ref:include/operator.h:191-201): the plain cell loop, identity on the constrained
    // (Dirichlet) rows only -- on a local-smoothing level the refinement-edge DoFs are ordinary DoFs there, rows and columns
    // (EDGE_IN).  After the zero-start pre-smoother x vanishes on them, so the edge rows are t_E = b_E - A_{E,I} x_I.
    void
    residual_raw(T *t, const T *b, const T *x)
    {
      Epilogue<T> e{t, x, nullptr, b, nullptr, T(0), T(0), T(0)};
      apply<MODE_RESIDUAL>(x, e, false, 0, tables->n_edge ? EDGE_IN : EDGE_OUT);
    }
    // Operator::vmult_interface_down (ref:include/operator.h:191-201)
    void
    vmult_interface_down(mgamd_vec &dst, const mgamd_vec &src) override
    {
      if (dst.n != n_dofs() || src.n != n_dofs() || dst.data == src.data)
        throw std::invalid_argument("vmult_interface_down: bad vectors");
      Epilogue<T> e{dst.as<T>(), src.as<T>(), nullptr, nullptr, nullptr, T(0), T(0), T(0)};
      apply<MODE_VMULT>(src.as<T>(), e, false, 0, tables->n_edge ? EDGE_IN : EDGE_OUT);
    }
    // dst = A^{edge DoFs unconstrained} (src restricted to the refinement-edge DoFs); tmp: scratch of n_dofs entries
    // (Operator::vmult_interface_up, ref:include/operator.h:203-226)
    void
    interface_up_raw(T *dst, const T *src, T *tmp)
    {
      const size_t n = n_dofs(), first_edge = (size_t)tables->n_interior + tables->n_tail;
      HIP_CHECK(hipMemsetAsync(tmp, 0, n * sizeof(T), ctx->stream));
      if (tables->n_edge)
        HIP_CHECK(hipMemcpyAsync(tmp + first_edge, src + first_edge, (size_t)tables->n_edge * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
      Epilogue<T> e{dst, tmp, nullptr, nullptr, nullptr, T(0), T(0), T(0)};
      apply<MODE_VMULT>(tmp, e, false, 0, EDGE_IN);
    }
    void
    vmult_interface_up(mgamd_vec &dst, const mgamd_vec &src) override
    {
      if (dst.n != n_dofs() || src.n != n_dofs() || dst.data == src.data)
        throw std::invalid_argument("vmult_interface_up: bad vectors");
      DBuf<T> tmp;
      tmp.alloc(n_dofs());
      interface_up_raw(dst.as<T>(), src.as<T>(), tmp.p);
      ctx->sync();
    }
    // One-byte codes for D^-1 of the DoFs tail_kernel handles (tail, refinement-edge, Dirichlet, hanging): on a level only a
    // few hundred distinct values occur there (node type x cell size x summation order), so the 255 most frequent ones go
    // into a table and the rest keeps reading the vector.  Values are matched by bit pattern: results do not change.
    void
    build_dinv_codes(const T *dinv)
    {
      dinv_coded = nullptr;
      if (getenv("MGAMD_NO_DINV_CODES"))
        return;
      const size_t n0 = tables->n_interior, n = (size_t)n_dofs() - n0;
      if (n < 4096) // small levels are latency-bound: nothing to gain
        return;
      std::vector<T> h(n);
      ctx->sync();
      HIP_CHECK(hipMemcpy(h.data(), dinv + n0, n * sizeof(T), hipMemcpyDeviceToHost));
      using Bits = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
      std::unordered_map<Bits, uint32_t> count;
      count.reserve(1024);
      auto bits = [](T v) {
        Bits b;
        std::memcpy(&b, &v, sizeof(T));
        return b;
      };
      for (size_t i = 0; i < n && count.size() < (1u << 20); ++i)
        ++count[bits(h[i])];
      std::vector<std::pair<uint32_t, Bits>> top;
      top.reserve(count.size());
      for (auto &kv : count)
        top.push_back({kv.second, kv.first});
      std::sort(top.begin(), top.end(), [](const auto &a, const auto &b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
      if (top.size() > 255)
        top.resize(255);
      std::vector<T>                    table(256, T(1));
      std::unordered_map<Bits, uint8_t> code_of;
      for (size_t k = 0; k < top.size(); ++k)
        {
          std::memcpy(&table[k], &top[k].second, sizeof(T));
          code_of[top[k].second] = (uint8_t)k;
        }
      std::vector<uint8_t> codes(n);
      for (size_t i = 0; i < n; ++i)
        {
          auto it  = code_of.find(bits(h[i]));
          codes[i] = it == code_of.end() ? (uint8_t)255 : it->second;
        }
      dinv_code.upload(codes);
      dinv_table.upload(table);
      dinv_coded = dinv;
    }

    void
    cheb_raw(T *out, const T *x, const T *xold, const T *b, const T *dinv, double f1, double f2, int from_b = 0, double c0 = 0.0,
             double *out_wide = nullptr)
    {
      // from_b = 1: x is c0 dinv b and xold = 0 (x is not read); from_b = 2: xold is c0 dinv b (xold is not read)
      // out_wide (float levels, plain Chebyshev pass only): the result goes there as doubles, `out` is not written
      Epilogue<T> e{out, x, from_b ? nullptr : xold, b, dinv, T(f1), T(f2), T(c0)};
      if (out_wide && (from_b != 0 || sizeof(T) != 4))
        throw std::invalid_argument("cheb_raw: a wide result needs a plain Chebyshev pass on float vectors");
      e.out_wide = out_wide;
      if (dinv == dinv_coded && dinv_code.p)
        {
          e.dinv_code  = dinv_code.p;
          e.dinv_table = dinv_table.p;
        }
      if (from_b == 1)
        apply<MODE_CHEB_FIRST>(x, e, false, 3.0);
      else if (from_b == 2)
        apply<MODE_CHEB_SECOND>(x, e, false, 4.0);
      else
        apply<MODE_CHEB>(x, e, false, xold ? 5.0 : 4.0);
    }

    // first pass of a smoothing step (x_old = 0, f1 = 0) on x + P x_c with the prolongation fused into the brick kernel
    // (kernels.hpp MODE_CHEB_PROLONGATE): x is updated in place to x + P x_c, f.scratch is clobbered on the tail
    void
    cheb_prolongate_raw(T *out, T *x, const T *b, const T *dinv, double f2, const FusedTransferHost<T> &f_in)
    {
      FusedTransferHost<T> f = f_in;
      f.x_inout              = x;
      Epilogue<T> e{out, x, nullptr, b, dinv, T(0), T(f2), T(0)};
      if (dinv == dinv_coded && dinv_code.p)
        {
          e.dinv_code  = dinv_code.p;
          e.dinv_table = dinv_table.p;
        }
      e.xs_flag = f.tail_flags;
      e.xs      = f.scratch;
      e.x_inout = x;
      apply<MODE_CHEB_PROLONGATE>(x, e, false, 4.0, EDGE_OUT, &f);
    }

    void
    vmult(mgamd_vec &dst, const mgamd_vec &src) override
    {
      if (dst.n != n_dofs() || src.n != n_dofs())
        throw std::invalid_argument("vmult: vector size mismatch");
      if (dst.data == src.data)
        throw std::invalid_argument("vmult: dst and src must differ");
      vmult_raw(dst.as<T>(), src.as<T>());
    }

    size_t
    read_debug_stamps(unsigned long long *out, size_t max_count) override
    {
      ctx->sync();
      const size_t n = std::min(max_count, stamps.n);
      if (n)
        HIP_CHECK(hipMemcpy(out, stamps.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      return n;
    }

    void
    compute_inverse_diagonal(mgamd_vec &d) override
    {
      if (d.n != n_dofs())
        throw std::invalid_argument("compute_inverse_diagonal: vector size mismatch");
      Epilogue<T> e{d.as<T>(), nullptr, nullptr, nullptr, nullptr, T(0), T(0)};
      apply<MODE_VMULT>(nullptr, e, true);
    }
  };

} // namespace mgamd
