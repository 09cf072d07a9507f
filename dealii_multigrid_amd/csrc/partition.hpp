// Spatial domain decomposition of the level hierarchy for one rank per GPU (the reference
// partitions with p4est + RepartitioningPolicyTools, ref:multigrid_throughput.cc:2066-2175,2219-2224).
//
//  * One ROOT level l_r is chosen (first level with enough cells per rank).  Its leaves are cut, in Morton order,
//    into n_ranks contiguous chunks of equal weight (weight = number of finest-level leaves below a root cell,
//    hanging-node cells weighted like ref:multigrid_throughput.cc:283-291 `CellWeightPolicy`).
//  * Every cell of a finer level belongs to the rank of its root-level ancestor, so a cell and its parent are always
//    on the same rank (the idea of the reference's FirstChildPolicy, ref:multigrid_throughput.cc:2156-2170): all
//    level transfers above the root level are rank-local.
//  * Levels below the root level are REPLICATED on every rank (they are tiny); the restriction onto the first
//    replicated level is completed by one all-reduce.
//  * DoFs on inter-rank interfaces exist as consistent copies on every sharing rank; partial sums are exchanged
//    symmetrically between the sharers and summed in ascending rank order, so all copies stay bitwise identical.
//    Which DoFs rank q shares is computed from the replicated mesh without communication: the keys referenced by q's
//    cells that have a foreign leaf in their 26-neighbourhood (candidate set); shared(r,q) = cand(r) n cand(q).
#pragma once
#include "level_tables.hpp"

#include <map>
#include <set>

namespace mgamd
{
  struct Partition
  {
    int n_ranks    = 1;
    int root_level = 0; // index into the level list (coarse -> fine): levels >= root_level are cut into n_ranks chunks
    // Two tiers (round 3): the levels [sub_root_level, root_level) are cut into n_ranks / group PARTS, and every part is held by
    // the `group` consecutive ranks part * group ... part * group + group - 1, which all do its work (the same idea as the
    // replicated levels below, applied to a rank subset: a level of 2-4 M DoFs costs a rank 1 / n_parts of its single-GPU time
    // instead of all of it).  The cuts are nested: a rank's cells of the root level descend from its part's cells, so the level
    // transfers stay rank-local; the partial sums of a restriction onto a part are summed over its group (SubsetComm::
    // replica_sum).  group == 1: sub_root_level == root_level, no such levels.
    int                                group = 1, sub_root_level = 0;
    std::vector<std::vector<uint16_t>> owner; // per level >= sub_root_level: rank (levels >= root_level) or part of each cell
    const std::vector<uint16_t> &
    level_owner(int level) const
    {
      return owner[level - sub_root_level];
    }
    bool
    replicated(int level) const
    {
      return level < sub_root_level;
    }
    bool
    subset(int level) const
    {
      return level >= sub_root_level && level < root_level;
    }
    int
    n_parts(int level) const // number of distinct pieces of the level
    {
      return replicated(level) ? 1 : (subset(level) ? n_ranks / group : n_ranks);
    }
    int
    part_of(int level, int rank) const // the piece of the level that `rank` works on
    {
      return replicated(level) ? 0 : (subset(level) ? rank / group : rank);
    }
  };

  inline Partition
  make_partition(const std::vector<const Tria *> &trias, int n_ranks, double hanging_weight = 2.0, size_t min_cells_per_rank = 32,
                 size_t min_root_cells = 0, int group = 1, size_t min_sub_root_cells = 0)
  {
    // min_root_cells: levels with fewer cells are not cut into n_ranks chunks.  A distributed level pays a halo exchange per
    // operator application (latency of a pack kernel + grouped send/recv + combine kernel) whatever its size, a replicated one
    // only its own, latency-bound, single-GPU time.  group > 1: levels below the root level with at least min_sub_root_cells
    // cells are cut into n_ranks / group parts (see Partition).
    Partition P;
    P.n_ranks       = n_ranks;
    const int nl    = (int)trias.size();
    int       root  = nl - 1;
    for (int l = 1; l < nl; ++l)
      if (trias[l]->n_cells() >= min_cells_per_rank * (size_t)n_ranks && trias[l]->n_cells() >= min_root_cells)
        {
          root = l;
          break;
        }
    if (root < 1)
      root = std::min(1, nl - 1);
    P.root_level = root;
    if (group < 1 || n_ranks % group != 0 || (group & (group - 1)) != 0)
      throw std::invalid_argument("partition: the group size must be a power of two that divides the number of ranks");
    if (group == n_ranks)
      group = 1; // one part = a replicated level
    const int n_parts = n_ranks / group;
    int       sub     = root;
    if (group > 1)
      for (int l = 1; l < root; ++l)
        if (trias[l]->n_cells() >= min_cells_per_rank * (size_t)n_parts && trias[l]->n_cells() >= min_sub_root_cells)
          {
            sub = l;
            break;
          }
    if (sub == root)
      group = 1;
    P.group          = group;
    P.sub_root_level = sub;
    const Tria &tf   = *trias[nl - 1];
    // weight of every cell of `t`: the finest-level leaves below it, hanging-node cells weighted
    auto weights = [&](const Tria &t) {
      std::vector<double> w(t.n_cells(), 0.0);
      for (size_t c = 0; c < tf.n_cells(); ++c)
        {
          const Cell &fc  = tf.cells[c];
          const int   anc = t.find_leaf(fc.level, fc.i, fc.j, fc.k);
          if (anc < 0)
            throw std::runtime_error("partition: level meshes are not nested");
          w[anc] += (tf.masks[c] >> MASK_FACE_SHIFT) ? hanging_weight : 1.0;
        }
      return w;
    };
    // cells [begin, end) of a level (Morton order) into `n` contiguous chunks of equal weight, numbered from `first`
    auto cut = [&](const std::vector<double> &w, size_t begin, size_t end, int n, int first, std::vector<uint16_t> &out) {
      double total = 0;
      for (size_t c = begin; c < end; ++c)
        total += w[c];
      double acc = 0;
      for (size_t c = begin; c < end; ++c)
        {
          const double mid = acc + 0.5 * w[c];
          const int    r   = total > 0 ? (int)(mid / total * n) : 0;
          out[c]           = (uint16_t)(first + std::min(std::max(r, 0), n - 1));
          acc += w[c];
        }
    };
    // owner of every cell of level l = owner of its ancestor on level `from`
    auto inherit = [&](int l, const Tria &from, const std::vector<uint16_t> &from_owner) {
      const Tria           &t = *trias[l];
      std::vector<uint16_t> o(t.n_cells());
      for (size_t c = 0; c < t.n_cells(); ++c)
        {
          const Cell &fc  = t.cells[c];
          const int   anc = from.find_leaf(fc.level, fc.i, fc.j, fc.k);
          if (anc < 0)
            throw std::runtime_error("partition: level meshes are not nested");
          o[c] = from_owner[anc];
        }
      return o;
    };
    P.owner.resize(nl - sub);
    const Tria           &tr = *trias[root];
    std::vector<uint16_t> root_owner(tr.n_cells());
    const auto            w_root = weights(tr);
    if (group == 1)
      cut(w_root, 0, tr.n_cells(), n_ranks, 0, root_owner);
    else
      {
        // parts on the sub-root level, inherited by the levels up to the root level; there every part's cells (a contiguous
        // Morton range) are cut into `group` chunks
        const Tria           &ts = *trias[sub];
        std::vector<uint16_t> part(ts.n_cells());
        cut(weights(ts), 0, ts.n_cells(), n_parts, 0, part);
        P.owner[0] = part;
        for (int l = sub + 1; l < root; ++l)
          P.owner[l - sub] = inherit(l, ts, part);
        const std::vector<uint16_t> root_part = inherit(root, ts, part);
        size_t                      b         = 0;
        while (b < root_part.size())
          {
            size_t e = b;
            while (e < root_part.size() && root_part[e] == root_part[b])
              ++e;
            cut(w_root, b, e, group, (int)root_part[b] * group, root_owner);
            b = e;
          }
        for (size_t c = 1; c < root_part.size(); ++c)
          if (root_part[c] < root_part[c - 1])
            throw std::runtime_error("partition: the cells of a part are not a contiguous Morton range");
      }
    P.owner[root - sub] = root_owner;
    for (int l = root + 1; l < nl; ++l)
      P.owner[l - sub] = inherit(l, tr, root_owner);
    return P;
  }

  // Partition-quality statistics of the level hierarchy, as the reference prints them in verbose mode
  // (MGTools::print_multigrid_statistics for a vector of triangulations, ref:include/mg_tools.h:267-512,
  // ref:multigrid_throughput.cc:1657-1665), evaluated for THIS partition (every rank knows all owners, so no communication):
  //   workload_eff       = [sum_l sum_r n_l,r / n_ranks] / [sum_l max_r n_l,r]       (mg_tools.h:9-38,285-286)
  //   workload_path_max  = sum_l max_r n_l,r                                          (mg_tools.h:306-309)
  //   vertical_eff       = children on the parent's rank / all children of refined cells  (mg_tools.h:140-204,372-373)
  //   horizontal_eff     = (local + ghost / 2) / (local + ghost), ghost = foreign cells in the 26-neighbourhood of owned
  //                        cells (mg_tools.h:228-247,437-439)
  //   mem_total          = bytes of the level meshes held by all ranks                (mg_tools.h:249-258,487-488)
  // n_l,r = cells of level l owned by rank r; a replicated level counts fully on every rank (every rank does that work).
  struct PartitionStatistics
  {
    double workload_eff = 1, workload_path_max = 0, vertical_eff = 1, horizontal_eff = 1, mem_total = 0;
  };

  inline PartitionStatistics
  partition_statistics(const std::vector<const Tria *> &trias, const Partition &P)
  {
    PartitionStatistics st;
    const int           nl = (int)trias.size(), nr = P.n_ranks;
    double              sum_all = 0, path = 0, v_local = 0, v_remote = 0, h_local = 0, h_remote = 0;
    for (int l = 0; l < nl; ++l)
      {
        const Tria         &t = *trias[l];
        std::vector<double> n(nr, 0.0);
        const int grp = P.subset(l) ? P.group : 1; // a part's cells are worked on by every rank of its group
        if (P.replicated(l) || nr == 1)
          for (int r = 0; r < nr; ++r)
            n[r] = (double)t.n_cells();
        else
          for (uint16_t o : P.level_owner(l))
            for (int m = 0; m < grp; ++m)
              n[o * grp + m] += 1.0;
        double mx = 0;
        for (double v : n)
          {
            sum_all += v;
            mx = std::max(mx, v);
          }
        path += mx;
        st.mem_total += (double)nr * ((double)t.cells.size() * (sizeof(Cell) + sizeof(uint16_t) + 16.0));
        const bool dist = !(P.replicated(l) || nr == 1);
        // horizontal: ghost cells of every rank
        for (int r = 0; r < nr; ++r)
          h_local += n[r];
        if (dist)
          {
            const auto                 &own = P.level_owner(l);
            for (size_t c = 0; c < t.n_cells(); ++c)
              {
                // ranks (other than the owner) that see cell c as a ghost: owners of its 26 neighbours
                const Cell &cc = t.cells[c];
                uint64_t    ranks = 0;
                for (int dz = -1; dz <= 1; ++dz)
                  for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                      {
                        if (!dx && !dy && !dz)
                          continue;
                        const int nb = t.find_leaf(cc.level, (int64_t)cc.i + dx, (int64_t)cc.j + dy, (int64_t)cc.k + dz);
                        if (nb >= 0)
                          ranks |= 1ull << own[nb];
                        else
                          for (int q = 0; q < 8; ++q)
                            {
                              // the neighbour position is refined (or outside): its children on the side facing c touch c
                              const int qb[3] = {q & 1, (q >> 1) & 1, q >> 2}, dd[3] = {dx, dy, dz};
                              bool      faces = true;
                              for (int d = 0; d < 3; ++d)
                                faces &= dd[d] == 0 || qb[d] == (dd[d] < 0 ? 1 : 0);
                              if (!faces)
                                continue;
                              const int f = t.find_leaf(cc.level + 1, 2 * ((int64_t)cc.i + dx) + qb[0], 2 * ((int64_t)cc.j + dy) + qb[1],
                                                        2 * ((int64_t)cc.k + dz) + qb[2]);
                              if (f >= 0)
                                ranks |= 1ull << own[f];
                            }
                      }
                ranks &= ~(1ull << own[c]);
                h_remote += (double)(grp * __builtin_popcountll(ranks)); // (subset levels: every member of a neighbouring part)
              }
          }
        // vertical: children of the refined cells of level l on level l + 1
        if (l + 1 < nl)
          {
            const Tria &tf    = *trias[l + 1];
            const bool  distf = !(P.replicated(l + 1) || nr == 1);
            for (size_t c = 0; c < t.n_cells(); ++c)
              {
                const Cell &cc = t.cells[c];
                if (tf.index.find(cell_key(cc)))
                  continue; // not refined between the two levels
                for (int q = 0; q < 8; ++q)
                  {
                    const int32_t *f = tf.index.find(cell_key(cc.level + 1, 2 * cc.i + (q & 1), 2 * cc.j + ((q >> 1) & 1), 2 * cc.k + (q >> 2)));
                    if (!f)
                      continue;
                    if (!dist && !distf)
                      v_local += nr; // replicated pair: local on every rank
                    else if (!dist)
                      { // replicated parent, distributed child: local for the child's owner, remote for nobody
                        v_local += 1;
                      }
                    else if (P.part_of(l, P.level_owner(l + 1)[*f] * (P.subset(l + 1) ? P.group : 1)) == P.level_owner(l)[c])
                      v_local += 1; // (owner of the child, as a rank, works on the parent's piece: nested cuts)
                    else
                      v_remote += 1;
                  }
              }
          }
      }
    st.workload_path_max = path;
    st.workload_eff      = path > 0 ? (sum_all / nr) / path : 1.0;
    st.vertical_eff      = (v_local + v_remote) > 0 ? v_local / (v_local + v_remote) : 1.0;
    st.horizontal_eff    = (h_local + h_remote) > 0 ? (h_local + 0.5 * h_remote) / (h_local + h_remote) : 1.0;
    return st;
  }

  // Keys referenced by the cells of `rank` that have a leaf of another rank in their 26-neighbourhood, sorted by key.
  // The flag byte holds the class (bits 0-1: 0 unconstrained, 1 Dirichlet, 2 hanging-own) and bit 7 = REGULAR: the key
  // is a node of one of those cells itself (not only reached through hanging-node resolution to a parent entity).
  // Only regular referencers can own a shared DoF: they are the ranks that have a transfer patch writing/reading it.
  inline std::vector<std::pair<uint64_t, uint8_t>>
  interface_candidate_keys(const Tria &tria, const std::vector<uint16_t> &owner, int rank, int p)
  {
    std::vector<std::pair<uint64_t, uint8_t>> out;
    LevelTables                               probe(tria, p, 1, nullptr, /*tables_only_helpers=*/true);
    const int                                 n = p + 1;
    for (size_t ci = 0; ci < tria.n_cells(); ++ci)
      {
        if (owner[ci] != rank)
          continue;
        const Cell &c       = tria.cells[ci];
        bool        foreign = false;
        // neighbourhood at the cell's own level, at the parent's level (coarser neighbours) and the children's level
        for (int dz = -1; dz <= 1 && !foreign; ++dz)
          for (int dy = -1; dy <= 1 && !foreign; ++dy)
            for (int dx = -1; dx <= 1 && !foreign; ++dx)
              {
                if (!dx && !dy && !dz)
                  continue;
                const int nb = tria.find_leaf(c.level, (int64_t)c.i + dx, (int64_t)c.j + dy, (int64_t)c.k + dz);
                if (nb >= 0)
                  {
                    if (owner[nb] != rank)
                      foreign = true;
                  }
                else
                  {
                    // region is outside the domain or covered by finer leaves: probe the 8 children positions
                    const int64_t bi = 2 * ((int64_t)c.i + dx), bj = 2 * ((int64_t)c.j + dy), bk = 2 * ((int64_t)c.k + dz);
                    for (int t = 0; t < 8 && !foreign; ++t)
                      {
                        const int f = tria.find_leaf(c.level + 1, bi + (t & 1), bj + ((t >> 1) & 1), bk + (t >> 2));
                        if (f >= 0 && owner[f] != rank)
                          foreign = true;
                      }
                  }
              }
        if (!foreign)
          continue;
        const uint16_t mask = tria.masks[ci];
        for (int z = 0; z < n; ++z)
          for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x)
              {
                const int      a[3] = {x, y, z};
                const uint64_t key  = probe.resolved_key(ci, a);
                bool           corner = false;
                const bool     via_parent =
                  (mask >> MASK_FACE_SHIFT) && LevelTables::node_on_constrained_entity(mask, p, a, &corner) && !corner;
                out.push_back({key, (uint8_t)((probe.key_on_boundary(key) ? 1 : 0) | (via_parent ? 0 : 0x80))});
                if (via_parent)
                  out.push_back({probe.own_key(c, a), (uint8_t)(2 | 0x80)});
              }
      }
    std::sort(out.begin(), out.end());
    // merge duplicates: a key is regular if any reference is
    std::vector<std::pair<uint64_t, uint8_t>> merged;
    for (const auto &e : out)
      if (!merged.empty() && merged.back().first == e.first)
        merged.back().second |= e.second;
      else
        merged.push_back(e);
    return merged;
  }



  // For `my_rank`: key -> the OTHER ranks sharing it and the set of regular referencers (n_ranks <= 64)
  inline std::map<uint64_t, SharedInfo>
  shared_keys(const Tria &tria, const std::vector<uint16_t> &owner, int n_ranks, int my_rank, int p)
  {
    if (n_ranks > 64)
      throw std::runtime_error("at most 64 ranks are supported");
    std::map<uint64_t, SharedInfo> out;
    const auto                   mine = interface_candidate_keys(tria, owner, my_rank, p);
    for (int q = 0; q < n_ranks; ++q)
      {
        if (q == my_rank)
          continue;
        const auto theirs = interface_candidate_keys(tria, owner, q, p);
        size_t     a = 0, b = 0;
        while (a < mine.size() && b < theirs.size())
          {
            if (mine[a].first < theirs[b].first)
              ++a;
            else if (theirs[b].first < mine[a].first)
              ++b;
            else
              {
                SharedInfo &si = out[mine[a].first];
                si.others |= 1ull << q;
                if (theirs[b].second & 0x80)
                  si.regular |= 1ull << q;
                if (mine[a].second & 0x80)
                  si.regular |= 1ull << my_rank;
                ++a;
                ++b;
              }
          }
      }
    for (const auto &kv : out)
      if (!kv.second.regular)
        throw std::runtime_error("partition: shared DoF without a regular referencer");
    return out;
  }



  // Halo plan of one rank for one level: which tail entries are exchanged with which peer, and how the received
  // partial sums are combined (ascending rank order, own contribution included at its position).
  struct HaloPlan
  {
    int                   n_ranks = 1, my_rank = 0;
    std::vector<int>      peers;        // ascending
    std::vector<uint32_t> peer_offset;  // [peers+1] into pack_idx / the concatenated send and recv buffers
    std::vector<uint32_t> pack_idx;     // tail index (global index - n_interior) of every send entry
    std::vector<uint32_t> sh_tail;      // per shared tail DoF: tail index
    std::vector<uint32_t> sh_ptr;       // CSR over contributions
    std::vector<int32_t>  sh_src;       // -1 = own value, else position in the concatenated recv buffer; ascending rank
    std::vector<int32_t>  sh_owner_src; // per shared tail DoF: -1 if this rank owns it, else recv position of the owner's value
  };

  inline HaloPlan
  make_halo_plan(const LevelTables &L, const std::map<uint64_t, SharedInfo> &shared, int n_ranks, int my_rank)
  {
    HaloPlan H;
    H.n_ranks = n_ranks;
    H.my_rank = my_rank;
    // tail DoFs of this rank that are shared, by key (std::map iterates in ascending key order = canonical order)
    std::vector<std::vector<std::pair<uint64_t, uint32_t>>> per_peer(n_ranks);
    std::map<uint32_t, int>                                 owner_of; // tail index -> owning rank
    for (const auto &kv : shared)
      {
        const int32_t *idx = L.keymap.find(kv.first);
        if (!idx)
          continue; // candidate key of an interior brick node cannot happen: candidates lie on cell closures of boundary cells
        const uint32_t gi = (uint32_t)*idx;
        if (gi < L.n_interior || gi >= L.n_interior + L.n_tail)
          continue; // Dirichlet / hanging copies carry no partial sums
        const uint32_t t = gi - L.n_interior;
        for (int q = 0; q < n_ranks; ++q)
          if ((kv.second.others >> q) & 1)
            per_peer[q].push_back({kv.first, t});
        owner_of[t] = shared_owner(kv.second);
      }
    H.peer_offset.push_back(0);
    std::map<uint32_t, std::vector<std::pair<int, int32_t>>> contrib; // tail index -> (rank, recv position)
    for (int q = 0; q < n_ranks; ++q)
      {
        if (per_peer[q].empty())
          continue;
        H.peers.push_back(q);
        for (const auto &e : per_peer[q])
          {
            contrib[e.second].push_back({q, (int32_t)H.pack_idx.size()});
            H.pack_idx.push_back(e.second);
          }
        H.peer_offset.push_back((uint32_t)H.pack_idx.size());
      }
    H.sh_ptr.push_back(0);
    for (auto &kv : contrib)
      {
        auto &v = kv.second;
        v.push_back({my_rank, -1});
        std::sort(v.begin(), v.end());
        H.sh_tail.push_back(kv.first);
        int32_t osrc = -1;
        for (const auto &e : v)
          {
            H.sh_src.push_back(e.second);
            if (e.first == owner_of[kv.first])
              osrc = e.second;
          }
        H.sh_owner_src.push_back(owner_of[kv.first] == my_rank ? -1 : osrc);
        if (owner_of[kv.first] != my_rank && osrc < 0)
          throw std::runtime_error("halo plan: owner of a shared DoF is not a peer");
        H.sh_ptr.push_back((uint32_t)H.sh_src.size());
      }
    return H;
  }
} // namespace mgamd
