// LevelOperator<T>::apply_P: the launch logic of one operator application.  Included by apply_inst.hip only (see
// level_operator.hpp): every (number type, degree, mode) instantiation pulls in the kernels of kernels.hpp for its slot groups.
#pragma once
#include "level_operator.hpp"

namespace mgamd
{
  template <typename T>
  template <int P, int MODE>
  void
  LevelOperator<T>::apply_P(const T *src, const Epilogue<T> &epi, bool diag, double words, int edge_mode, const FusedTransferHost<T> *fused)
  {
    constexpr int BASE = base_mode(MODE);
    ApplyArgs<T, P> a;
    const uint32_t  first_edge = tables->n_interior + tables->n_tail;
    a.gather_limit  = first_edge + (edge_mode == EDGE_IN ? tables->n_edge : 0);
    a.scatter_limit = first_edge + (edge_mode != EDGE_OUT ? tables->n_edge : 0);
    const uint32_t tail_end = tables->n_tail + (edge_mode != EDGE_OUT ? tables->n_edge : 0);
    a.m          = mats<P>();
    a.src        = src;
    a.tail_acc   = tail_acc.p;
    a.n_interior = tables->n_interior;
    a.ablate     = ablate;
    a.stamps     = nullptr;
    a.epi        = epi;
    // 2^3 bricks and single cells in one launch (lattice_apply_small_kernel) when both exist
    GroupDev<T> *g2 = nullptr, *g1 = nullptr;
    if (P >= 2 && !diag && merge_small)
      for (auto &g : groups)
        {
          if (g->n_slots && g->B == 2)
            g2 = g.get();
          else if (g->n_slots && g->B == 1)
            g1 = g.get();
        }
    const bool merged = g2 && g1;
    // p = 1: the 8^3 bricks together with the cell clusters (lattice_cluster_kernel)
    GroupDev<T> *g8 = nullptr, *gc = nullptr;
    if (P == 1 && !diag && merge_small)
      for (auto &g : groups)
        {
          if (g->n_slots && g->B == 8 && g->B != prof_B && !g->constrained)
            g8 = g.get();
          else if (g->n_slots && g->B == 1 && g->has_clusters())
            gc = g.get();
        }
    const bool merged_p1 = g8 && gc;

    // Sharded level: the slots that touch shared DoFs first (LevelTables puts them at the front of every group), then
    // the halo exchange on the side queue UNDERNEATH the remaining slots (ref: MatrixFree::cell_loop overlaps its ghost
    // exchange with the interior cell ranges, ref:include/operator.h:166-167).  Plain launches per group and range.
    if (halo && !diag && halo_overlap)
      {
        auto halo_end = [&](const GroupDev<T> &g) -> size_t {
          size_t nh = g.n_halo;
          if (P == 1 && g.has_clusters()) // the cluster kernel works on whole 256-cell clusters
            nh = std::min(g.n_slots, (nh + CLUSTER_CELLS - 1) / CLUSTER_CELLS * CLUSTER_CELLS);
          return nh;
        };
        auto launch_range = [&](GroupDev<T> *g, size_t begin, size_t end) {
          if (begin >= end)
            return;
          if (P == 1 && g->has_clusters())
            launch_clusters(ctx->stream, *g, src, epi, BASE == MODE_CHEB_FIRST, (uint32_t)(begin / CLUSTER_CELLS),
                            (uint32_t)((end + CLUSTER_CELLS - 1) / CLUSTER_CELLS));
          else
            {
              a.g = g->view(begin, end);
              if (MODE != BASE && fused && fused->group >= 0 && g == groups[fused->group].get())
                {
                  if constexpr (MODE != BASE)
                    launch_fused<P, MODE>(ctx->stream, a, begin, nullptr, *fused);
                }
              else
                dispatch_B<T, P, BASE>(ctx, ctx->stream, g->B, g->constrained, a, diag);
            }
        };
        size_t n_interior_slots = 0;
        for (auto &g : groups)
          n_interior_slots += g->n_slots - halo_end(*g);
        if (n_interior_slots > 0)
          {
            for (auto &g : groups)
              launch_range(g.get(), 0, halo_end(*g));
            ctx->order_after(ctx->side, ctx->stream); // the side queue waits for the halo slots only
            // the exchange is enqueued BEFORE the interior slots: their persistent workgroups fill every CU's LDS until the
            // launch ends, so RCCL's send/recv kernels must be resident first to run underneath them (the simulator's
            // exchange blocks the host instead: no overlap there, same results)
            exchange_add_raw(tail_acc.p, ctx->side);
            for (auto &g : groups)
              launch_range(g.get(), halo_end(*g), g->n_slots);
            ctx->order_after(ctx->stream, ctx->side);
            launch_tail<MODE>(ctx->stream, 0, tail_end, true, epi, diag);
            return;
          }
      }
    const hipStream_t main = ctx->stream;
    auto prof_begin = [&](const GroupDev<T> &g) {
      const bool prof = ctx->profile && !diag && MODE == MODE_CHEB && !g.constrained && g.B == (ctx->prof_brick ? ctx->prof_brick : prof_B);
      if (prof)
        {
          if (ctx->prof_used == ctx->prof_events.size())
            {
              hipEvent_t e0, e1;
              HIP_CHECK(hipEventCreate(&e0));
              HIP_CHECK(hipEventCreate(&e1));
              ctx->prof_events.push_back({e0, e1});
            }
          HIP_CHECK(hipEventRecord(ctx->prof_events[ctx->prof_used].first, main));
        }
      return prof;
    };
    auto prof_end = [&](const GroupDev<T> &g, size_t n_slots) {
      HIP_CHECK(hipEventRecord(ctx->prof_events[ctx->prof_used].second, main));
      ++ctx->prof_used;
      // algorithmic bytes of THIS kernel: `words` per slot-interior DoF (the fused epilogue is complete for them); for
      // the (N-1)^3 - (N-2)^3 shell DoFs a brick is responsible for, one gathered word and one partial sum (the other
      // words of their epilogue are tail_kernel's)
      // (D^-1 of slot-interior DoFs is evaluated in closed form, not read, by the p = 1 kernels and by the persistent
      // 17-point lattice kernels: one word less per interior DoF)
      // prof_bytes keeps SURVEY 8(d)'s per-unit figure (the algorithm's words); prof_bytes_moved is the kernel's own count
      const bool   closed_dinv = P == 1 || (g.N * g.N > 256 && use_persistent<T, P>());
      const double w_interior  = words - (closed_dinv ? 1.0 : 0.0);
      const double n1 = (double)(g.N - 1), n2 = (double)(g.N - 2);
      ctx->prof_bytes += sizeof(T) * (double)n_slots * (words * n2 * n2 * n2 + 2.0 * (n1 * n1 * n1 - n2 * n2 * n2));
      ctx->prof_bytes_moved += sizeof(T) * (double)n_slots * (w_interior * n2 * n2 * n2 + 2.0 * (n1 * n1 * n1 - n2 * n2 * n2));
    };
    // plain + constrained bricks of one size share a launch (p = 1: the only degree with constrained bricks above B = 2)
    auto constrained_partner = [&](GroupDev<T> *g) -> GroupDev<T> * {
      if (P != 1 || diag || !merge_small || g->constrained || g->B <= 2 || (merged_p1 && g == g8))
        return nullptr;
      for (auto &q : groups)
        if (q->constrained && q->B == g->B && q->n_slots)
          return q.get();
      return nullptr;
    };
    for (size_t gi = 0; gi < groups.size(); ++gi)
      {
        GroupDev<T> *g = groups[gi].get();
        if (!g->n_slots)
          continue;
        if ((merged && g == g1) || (merged_p1 && g == gc))
          continue; // done together with the 2^3 (8^3) bricks
        if (g->constrained && !diag && merge_small && P == 1)
          {
            bool has_plain = false;
            for (auto &q : groups)
              has_plain |= !q->constrained && q->B == g->B && q->n_slots && !(merged_p1 && q.get() == g8);
            if (has_plain)
              continue; // launched together with the plain bricks of its size
          }
        const bool prof = prof_begin(*g);
        launch_group<P, MODE>(main, a, g, (merged && g == g2) ? g1 : nullptr, (merged_p1 && g == g8) ? gc : nullptr, src, epi, diag, 0,
                              g->n_slots, constrained_partner(g), fused);
        if (prof)
          prof_end(*g, g->n_slots);
      }
    if (halo)
      exchange_add_raw(tail_acc.p); // complete the shared tail sums across ranks before the epilogue
    launch_tail<MODE>(main, 0, tail_end, true, epi, diag);
  }

} // namespace mgamd
