// Octree triangulations of the cube [-1,1]^3 with full (face+edge+corner) 2:1 balance:
// the host-side stand-in for deal.II's parallel::distributed::Triangulation<3> (p4est) as
// the reference uses it (ref:multigrid_throughput.cc:2041-2062), the mesh generators of
// ref:include/grid_generator.h:3-140, and one step of
// MGTransferGlobalCoarseningTools::create_geometric_coarsening_sequence
// (ref:multigrid_throughput.cc:2219-2224).  Setup-time code; nothing here runs per V-cycle.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mgamd
{
  constexpr int LMAX = 15; // integer coordinates: a level-l cell has edge 2^(LMAX-l)

  struct Cell
  {
    uint32_t i, j, k;
    uint8_t  level;
  };

  inline uint64_t
  cell_key(int l, uint32_t i, uint32_t j, uint32_t k)
  {
    return ((uint64_t)l << 57) | ((uint64_t)i << 38) | ((uint64_t)j << 19) | (uint64_t)k;
  }
  inline uint64_t
  cell_key(const Cell &c)
  {
    return cell_key(c.level, c.i, c.j, c.k);
  }

  inline uint64_t
  spread3(uint64_t x)
  {
    x &= 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffULL;
    x = (x | x << 16) & 0x1f0000ff0000ffULL;
    x = (x | x << 8) & 0x100f00f00f00f00fULL;
    x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
    x = (x | x << 2) & 0x1249249249249249ULL;
    return x;
  }
  inline uint64_t
  morton(const Cell &c)
  {
    const int s = LMAX - c.level;
    return spread3((uint64_t)c.i << s) | (spread3((uint64_t)c.j << s) << 1) | (spread3((uint64_t)c.k << s) << 2);
  }

  // open-addressing hash map uint64 -> int32 (keys never removed except via rebuild)
  class FlatMap
  {
  public:
    static constexpr uint64_t EMPTY = ~0ULL;
    FlatMap() { rehash(1024); }
    void
    reserve(size_t n)
    {
      if (n * 2 > cap)
        rehash(n * 2);
    }
    void
    clear()
    {
      std::fill(keys.begin(), keys.end(), EMPTY);
      count = 0;
    }
    static uint64_t
    mix(uint64_t x)
    {
      x ^= x >> 33;
      x *= 0xff51afd7ed558ccdULL;
      x ^= x >> 33;
      x *= 0xc4ceb9fe1a85ec53ULL;
      x ^= x >> 33;
      return x;
    }
    int32_t *
    find(uint64_t key)
    {
      size_t h = mix(key) & (cap - 1);
      while (keys[h] != EMPTY)
        {
          if (keys[h] == key)
            return &vals[h];
          h = (h + 1) & (cap - 1);
        }
      return nullptr;
    }
    const int32_t *
    find(uint64_t key) const
    {
      return const_cast<FlatMap *>(this)->find(key);
    }
    // returns pointer to value; `inserted` tells whether the key was new (value then = init)
    int32_t *
    insert(uint64_t key, int32_t init, bool *inserted = nullptr)
    {
      if ((count + 1) * 2 > cap)
        rehash(cap * 2);
      size_t h = mix(key) & (cap - 1);
      while (keys[h] != EMPTY)
        {
          if (keys[h] == key)
            {
              if (inserted)
                *inserted = false;
              return &vals[h];
            }
          h = (h + 1) & (cap - 1);
        }
      keys[h] = key;
      vals[h] = init;
      ++count;
      if (inserted)
        *inserted = true;
      return &vals[h];
    }
    void
    erase_all_and_reserve(size_t n)
    {
      size_t c = 1024;
      while (c < 2 * n)
        c <<= 1;
      cap = c;
      keys.assign(cap, EMPTY);
      vals.assign(cap, 0);
      count = 0;
    }
    size_t
    size() const
    {
      return count;
    }

  private:
    void
    rehash(size_t ncap)
    {
      size_t c = 1024;
      while (c < ncap)
        c <<= 1;
      std::vector<uint64_t> ok;
      std::vector<int32_t>  ov;
      ok.swap(keys);
      ov.swap(vals);
      cap = c;
      keys.assign(cap, EMPTY);
      vals.assign(cap, 0);
      count = 0;
      for (size_t t = 0; t < ok.size(); ++t)
        if (ok[t] != EMPTY)
          insert(ok[t], ov[t]);
    }
    std::vector<uint64_t> keys;
    std::vector<int32_t>  vals;
    size_t                cap = 0, count = 0;
  };

  // Constraint configuration of a cell, in the spirit of deal.II's
  // internal::MatrixFreeFunctions::HangingNodes: bits 0-2 child position (cx,cy,cz) inside the
  // parent, bits 3-5 hanging faces normal to x/y/z (on the parent's boundary), bits 6-8 hanging
  // edges running along x/y/z (on the parent's edges).
  enum : uint16_t
  {
    MASK_FACE_SHIFT = 3,
    MASK_EDGE_SHIFT = 6
  };

  class Tria
  {
  public:
    std::vector<Cell>     cells; // leaves in Morton (p4est) order
    std::vector<uint16_t> masks; // constraint configuration per leaf
    FlatMap               index; // cell_key -> position in `cells`

    static Tria
    create(const std::string &geometry, unsigned n_ref_global, unsigned n_ref_local);
    static Tria
    from_cells(std::vector<Cell> leaves); // must be a balanced partition of the cube
    // a caller-built octree (e.g. the active cells of a parallel::distributed::Triangulation over one root cell): checks that
    // the leaves tile the cube exactly and are 2:1 balanced across faces, edges and corners (p4est's full balance, which
    // deal.II requests); throws std::invalid_argument otherwise
    static Tria
    from_leaves_checked(std::vector<Cell> leaves);
    Tria
    coarsen_global() const; // one level of the geometric coarsening sequence
    // local smoothing: ALL cells of refinement level `level`, active or not (DoFHandler::distribute_mg_dofs levels)
    Tria
    level_mesh(int level) const
    {
      std::vector<Cell> out;
      FlatMap           seen;
      seen.erase_all_and_reserve(cells.size());
      for (const Cell &c : cells)
        if (c.level >= level)
          {
            const int  s = c.level - level;
            const Cell a{c.i >> s, c.j >> s, c.k >> s, (uint8_t)level};
            bool       ins;
            seen.insert(cell_key(a), 1, &ins);
            if (ins)
              out.push_back(a);
          }
      return from_cells(std::move(out));
    }

    size_t
    n_cells() const
    {
      return cells.size();
    }
    int
    n_levels() const; // n_global_levels()
    // index of the leaf covering region (l,i,j,k) with level <= l; -1 if outside or finer
    int
    find_leaf(int l, int64_t i, int64_t j, int64_t k) const;
    // dealii::parallel::Helper::is_constrained (ref:multigrid_throughput.cc:231-267)
    size_t
    n_cells_with_hanging_nodes() const;

  private:
    void
    finalize(); // sort, index, masks
  };

  // ------------------------------------------------------------------ implementation
  namespace detail
  {
    struct LeafSet
    {
      FlatMap set; // key -> 1 (present) / 0 (removed)
      size_t  n = 0;
      bool
      has(int l, uint32_t i, uint32_t j, uint32_t k) const
      {
        const int32_t *v = set.find(cell_key(l, i, j, k));
        return v && *v;
      }
      void
      add(const Cell &c)
      {
        bool     ins;
        int32_t *v = set.insert(cell_key(c), 1, &ins);
        if (!ins && !*v)
          *v = 1;
        ++n;
      }
      void
      remove(const Cell &c)
      {
        *set.find(cell_key(c)) = 0;
        --n;
      }
      // covering leaf of region at level <= l
      bool
      find(int l, int64_t i, int64_t j, int64_t k, Cell &out) const
      {
        const int64_t n1 = (int64_t)1 << l;
        if (i < 0 || j < 0 || k < 0 || i >= n1 || j >= n1 || k >= n1)
          return false;
        for (int ll = l; ll >= 0; --ll)
          {
            const int s = l - ll;
            if (has(ll, (uint32_t)(i >> s), (uint32_t)(j >> s), (uint32_t)(k >> s)))
              {
                out = Cell{(uint32_t)(i >> s), (uint32_t)(j >> s), (uint32_t)(k >> s), (uint8_t)ll};
                return true;
              }
          }
        return false;
      }
    };

    inline void
    children(const Cell &c, Cell out[8])
    {
      for (int t = 0; t < 8; ++t)
        out[t] = Cell{2 * c.i + (t & 1), 2 * c.j + ((t >> 1) & 1), 2 * c.k + (t >> 2), (uint8_t)(c.level + 1)};
    }

    // ripple 2:1 balance seeded with `work` (the cells that may be too fine for a neighbour);
    // every cell created on the way is appended to `created`
    inline void
    balance(LeafSet &ls, std::vector<Cell> &work, std::vector<Cell> &created)
    {
      while (!work.empty())
        {
          const Cell c = work.back();
          work.pop_back();
          if (!ls.has(c.level, c.i, c.j, c.k))
            continue;
          bool again = false;
          for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
              for (int dx = -1; dx <= 1; ++dx)
                {
                  if (!dx && !dy && !dz)
                    continue;
                  Cell nb;
                  if (ls.find(c.level, (int64_t)c.i + dx, (int64_t)c.j + dy, (int64_t)c.k + dz, nb) &&
                      nb.level + 1 < c.level)
                    {
                      ls.remove(nb);
                      Cell ch[8];
                      children(nb, ch);
                      for (auto &x : ch)
                        {
                          ls.add(x);
                          work.push_back(x);
                          created.push_back(x);
                        }
                      again = true;
                    }
                }
          if (again)
            work.push_back(c);
        }
    }

    // The same ripple seeded with cells that were just REFINED (`parents`: their eight children are leaves, or refined further).  A
    // child is too fine for a neighbour exactly if the leaf that covers one of the PARENT's 26 same-level neighbour regions is coarser
    // than the parent (such a leaf touches the parent, hence the child in that corner; and a leaf that is too coarse for a child
    // covers the parent's neighbour region the child's neighbour region lies in): one check per family instead of eight, which is
    // what the refinement rounds spend their time on (octant NRefGlobal 9, 16.9 M cells: mesh generation 32 -> 18 s together with
    // the threaded mask computation of finalize(); the sort and the hash index of finalize() are what is left).
    inline void
    balance_families(LeafSet &ls, std::vector<Cell> &parents, std::vector<Cell> &created)
    {
      while (!parents.empty())
        {
          const Cell p = parents.back();
          parents.pop_back();
          for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
              for (int dx = -1; dx <= 1; ++dx)
                {
                  if (!dx && !dy && !dz)
                    continue;
                  Cell nb;
                  // (the leaf found after a refinement is a child of the one refined: one level finer, maybe still too coarse)
                  while (ls.find(p.level, (int64_t)p.i + dx, (int64_t)p.j + dy, (int64_t)p.k + dz, nb) && nb.level < p.level)
                    {
                      ls.remove(nb);
                      Cell ch[8];
                      children(nb, ch);
                      for (auto &x : ch)
                        {
                          ls.add(x);
                          created.push_back(x);
                        }
                      parents.push_back(nb);
                    }
                }
        }
    }

    inline std::vector<Cell>
    collect(const LeafSet &ls, const std::vector<Cell> &candidates)
    {
      std::vector<Cell> out;
      out.reserve(ls.n);
      for (const Cell &c : candidates)
        if (ls.has(c.level, c.i, c.j, c.k))
          out.push_back(c);
      return out;
    }
  } // namespace detail

  inline int
  Tria::find_leaf(int l, int64_t i, int64_t j, int64_t k) const
  {
    const int64_t n1 = (int64_t)1 << l;
    if (i < 0 || j < 0 || k < 0 || i >= n1 || j >= n1 || k >= n1)
      return -1;
    for (int ll = l; ll >= 0; --ll)
      {
        const int      s = l - ll;
        const int32_t *v = index.find(cell_key(ll, (uint32_t)(i >> s), (uint32_t)(j >> s), (uint32_t)(k >> s)));
        if (v)
          return *v;
      }
    return -1;
  }

  inline int
  Tria::n_levels() const
  {
    int m = 0;
    for (const Cell &c : cells)
      m = std::max<int>(m, c.level);
    return m + 1;
  }

  inline void
  Tria::finalize()
  {
    std::vector<std::pair<uint64_t, uint32_t>> order(cells.size());
    for (size_t t = 0; t < cells.size(); ++t)
      order[t] = {morton(cells[t]), (uint32_t)t};
    std::sort(order.begin(), order.end());
    std::vector<Cell> sorted(cells.size());
    for (size_t t = 0; t < cells.size(); ++t)
      sorted[t] = cells[order[t].second];
    cells.swap(sorted);
    index.erase_all_and_reserve(cells.size());
    for (size_t t = 0; t < cells.size(); ++t)
      index.insert(cell_key(cells[t]), (int32_t)t);
    masks.assign(cells.size(), 0);
    // (read-only neighbour searches, six per cell: split over the host threads -- a third of the mesh generation time)
    auto mask_range = [&](size_t begin, size_t end) {
    for (size_t t = begin; t < end; ++t)
      {
        const Cell &c = cells[t];
        if (c.level == 0)
          continue;
        const int      l     = c.level;
        const int      cp[3] = {(int)(c.i & 1), (int)(c.j & 1), (int)(c.k & 1)};
        const int64_t  id[3] = {c.i, c.j, c.k};
        uint16_t       m     = (uint16_t)(cp[0] | (cp[1] << 1) | (cp[2] << 2));
        bool           face[3];
        for (int d = 0; d < 3; ++d)
          {
            int64_t n[3] = {id[0], id[1], id[2]};
            n[d] += cp[d] ? 1 : -1;
            const int nb = find_leaf(l, n[0], n[1], n[2]);
            face[d]      = nb >= 0 && cells[nb].level < l;
            if (face[d])
              m |= (uint16_t)(1u << (MASK_FACE_SHIFT + d));
          }
        for (int d = 0; d < 3; ++d)
          {
            const int e = (d + 1) % 3, f = (d + 2) % 3;
            int64_t   n[3] = {id[0], id[1], id[2]};
            n[e] += cp[e] ? 1 : -1;
            n[f] += cp[f] ? 1 : -1;
            const int nb = find_leaf(l, n[0], n[1], n[2]);
            if (face[e] || face[f] || (nb >= 0 && cells[nb].level < l))
              m |= (uint16_t)(1u << (MASK_EDGE_SHIFT + d));
          }
        masks[t] = m;
      }
    };
    const size_t n_threads = cells.size() < 200000 ? 1 : std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16);
    if (n_threads == 1)
      mask_range(0, cells.size());
    else
      {
        std::vector<std::thread> pool;
        for (size_t w = 0; w < n_threads; ++w)
          pool.emplace_back(mask_range, cells.size() * w / n_threads, cells.size() * (w + 1) / n_threads);
        for (auto &th : pool)
          th.join();
      }
  }

  inline size_t
  Tria::n_cells_with_hanging_nodes() const
  {
    size_t n = 0;
    for (uint16_t m : masks)
      if (m >> MASK_FACE_SHIFT)
        ++n;
    return n;
  }

  inline Tria
  Tria::from_cells(std::vector<Cell> leaves)
  {
    Tria t;
    t.cells = std::move(leaves);
    t.finalize();
    return t;
  }

  inline Tria
  Tria::from_leaves_checked(std::vector<Cell> leaves)
  {
    if (leaves.empty())
      throw std::invalid_argument("octree: no leaves");
    for (const Cell &c : leaves)
      if (c.level > LMAX - 1 || ((uint64_t)c.i >> c.level) || ((uint64_t)c.j >> c.level) || ((uint64_t)c.k >> c.level))
        throw std::invalid_argument("octree: leaf (level " + std::to_string(c.level) + ", " + std::to_string(c.i) + ", " + std::to_string(c.j) +
                                    ", " + std::to_string(c.k) + ") outside the level's index range or deeper than " + std::to_string(LMAX - 1));
    Tria t = from_cells(std::move(leaves));
    // exact tiling: in Morton order leaf t covers [morton, morton + 8^(LMAX - level))
    uint64_t next = 0;
    for (const Cell &c : t.cells)
      {
        if (morton(c) != next)
          throw std::invalid_argument(morton(c) < next ? "octree: leaves overlap (a cell and one of its descendants, or a duplicate)" :
                                                         "octree: the leaves do not cover the cube (gap in the Morton order)");
        next += (uint64_t)1 << (3 * (LMAX - c.level));
      }
    if (next != ((uint64_t)1 << (3 * LMAX)))
      throw std::invalid_argument("octree: the leaves do not cover the cube");
    // 2:1 balance over all 26 neighbour directions: the leaf that covers a same-level neighbour region may be at most one
    // level coarser (a finer neighbourhood is checked from the finer side)
    for (const Cell &c : t.cells)
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx)
            {
              if (!dx && !dy && !dz)
                continue;
              const int nb = t.find_leaf(c.level, (int64_t)c.i + dx, (int64_t)c.j + dy, (int64_t)c.k + dz);
              if (nb >= 0 && (int)t.cells[nb].level < (int)c.level - 1)
                throw std::invalid_argument("octree: not 2:1 balanced across faces, edges and corners at leaf (level " + std::to_string(c.level) +
                                            ", " + std::to_string(c.i) + ", " + std::to_string(c.j) + ", " + std::to_string(c.k) + ")");
            }
    return t;
  }

  inline Tria
  Tria::create(const std::string &geometry, unsigned n_ref_global, unsigned n_ref_local)
  {
    using namespace detail;
    if (n_ref_global > (unsigned)LMAX - 1)
      throw std::runtime_error("NRefGlobal too large");
    auto uniform = [](unsigned L) {
      std::vector<Cell> v;
      const uint32_t    n = 1u << L;
      v.reserve((size_t)n * n * n);
      for (uint32_t k = 0; k < n; ++k)
        for (uint32_t j = 0; j < n; ++j)
          for (uint32_t i = 0; i < n; ++i)
            v.push_back(Cell{i, j, k, (uint8_t)L});
      return v;
    };
    auto center = [](const Cell &c, double x[3]) {
      const double h = 2.0 / (double)(1u << c.level);
      x[0]           = -1.0 + (c.i + 0.5) * h;
      x[1]           = -1.0 + (c.j + 0.5) * h;
      x[2]           = -1.0 + (c.k + 0.5) * h;
    };
    auto in_octant = [&](const Cell &c) {
      double x[3];
      center(c, x);
      return x[0] <= 0.0 && x[1] <= 0.0 && x[2] <= 0.0;
    };
    auto norm = [&](const Cell &c) {
      double x[3];
      center(c, x);
      return std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    };
    // generic driver: start from a uniform mesh, then apply flagged-refinement rounds
    auto run = [&](unsigned L0, const std::vector<std::function<bool(const Cell &)>> &rounds) {
      std::vector<Cell> leaves = uniform(L0);
      if (rounds.empty())
        return from_cells(std::move(leaves));
      LeafSet ls;
      ls.set.erase_all_and_reserve(leaves.size() * 2);
      for (auto &c : leaves)
        ls.add(c);
      std::vector<Cell> all = leaves; // the leaves at the start of a round, then every cell created in it
      for (auto &flag : rounds)
        {
          const size_t n_cur = all.size();
          // refine, recording new cells
          std::vector<Cell> parents;
          for (size_t t = 0; t < n_cur; ++t)
            {
              const Cell c = all[t];
              if (flag(c))
                {
                  ls.remove(c);
                  Cell ch[8];
                  children(c, ch);
                  for (auto &x : ch)
                    {
                      ls.add(x);
                      all.push_back(x);
                    }
                  parents.push_back(c);
                }
            }
          balance_families(ls, parents, all); // newly created cells are remembered in `all`
          // compact the candidate list: the leaves
          all = collect(ls, all);
        }
      return from_cells(std::move(all));
    };
    using Round = std::function<bool(const Cell &)>;
    if (geometry == "hypercube") // ref:multigrid_throughput.cc:2056-2060
      return run(n_ref_global, {});
    if (geometry == "quadrant") // ref:include/grid_generator.h:34-65
      {
        if (n_ref_global == 0)
          return run(0, {});
        std::vector<Round> r(n_ref_global - 1, in_octant);
        return run(1, r);
      }
    if (geometry == "quadrant_flexible") // ref:include/grid_generator.h:69-92
      {
        std::vector<Round> r(n_ref_local, in_octant);
        return run(n_ref_global, r);
      }
    if (geometry == "annulus") // ref:include/grid_generator.h:96-140
      {
        if (n_ref_global == 0)
          return run(0, {});
        std::vector<Round> r;
        if (n_ref_global >= 1)
          r.push_back([&](const Cell &c) { return norm(c) < 0.55; });
        if (n_ref_global >= 2)
          r.push_back([&](const Cell &c) { const double n = norm(c); return 0.3 <= n && n <= 0.43; });
        if (n_ref_global >= 3)
          r.push_back([&](const Cell &c) { const double n = norm(c); return 0.335 <= n && n <= 0.39; });
        return run(n_ref_global > 3 ? n_ref_global - 3 : 0, r);
      }
    if (geometry == "circle") // ref:include/grid_generator.h:3-30
      {
        auto near_origin = [](const Cell &c) {
          const double h = 2.0 / (double)(1u << c.level);
          for (int v = 0; v < 8; ++v)
            {
              const double x = -1.0 + (c.i + (v & 1)) * h, y = -1.0 + (c.j + ((v >> 1) & 1)) * h,
                           z = -1.0 + (c.k + (v >> 2)) * h;
              if (std::sqrt(x * x + y * y + z * z) < 1.0 / (4.0 * 3.14159265358979323846))
                return true;
            }
          return false;
        };
        std::vector<Round> r(n_ref_global > 3 ? n_ref_global - 3 : 0, near_origin);
        return run(std::min(n_ref_global, 3u), r);
      }
    throw std::runtime_error("GeometryType '" + geometry + "' not implemented");
  }

  inline Tria
  Tria::coarsen_global() const
  {
    using namespace detail;
    // complete families: 8 consecutive leaves (Morton order) of equal level with a common parent
    std::vector<Cell> out, kept;
    out.reserve(cells.size());
    size_t t = 0;
    while (t < cells.size())
      {
        const Cell &c      = cells[t];
        bool        family = c.level > 0 && t + 8 <= cells.size() && !(c.i & 1) && !(c.j & 1) && !(c.k & 1);
        if (family)
          for (int s = 1; s < 8; ++s)
            {
              const Cell &d = cells[t + s];
              if (d.level != c.level || (d.i >> 1) != (c.i >> 1) || (d.j >> 1) != (c.j >> 1) || (d.k >> 1) != (c.k >> 1))
                {
                  family = false;
                  break;
                }
            }
        if (family)
          {
            out.push_back(Cell{c.i >> 1, c.j >> 1, c.k >> 1, (uint8_t)(c.level - 1)});
            t += 8;
          }
        else
          {
            out.push_back(c);
            kept.push_back(c);
            ++t;
          }
      }
    if (kept.empty())
      return from_cells(std::move(out));
    LeafSet ls;
    ls.set.erase_all_and_reserve(out.size() * 2);
    for (auto &c : out)
      ls.add(c);
    // re-balance: only cells that were not coarsened can be too fine for a neighbour
    std::vector<Cell> all = out, work = kept;
    balance(ls, work, all);
    return from_cells(collect(ls, all));
  }
} // namespace mgamd
