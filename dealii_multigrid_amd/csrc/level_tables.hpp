// Host-side level description: what deal.II's DoFHandler + AffineConstraints + MatrixFree::reinit
// produce for one multigrid level in the reference (ref:include/operator.h:24-47,
// ref:multigrid_throughput.cc:1578-1595), re-designed for a GPU with 160 KB of LDS per CU.
//
// Data layout:
//  * The leaves are grouped into SLOTS: a slot is either a BRICK = complete uniform subtree of
//    B^3 equal-size cells without hanging nodes (B a power of two, node lattice N = p*B+1 <= 17 per
//    direction), or a single cell (B = 1, may carry hanging faces/edges).
//  * DoFs are numbered  [ I | T | D | H ]:
//      I  slot-interior DoFs (strictly inside a slot's node lattice; touched by that slot only),
//         contiguous per slot in lattice-lexicographic order  -> implicit, coalesced addressing
//      T  'tail': unconstrained DoFs on slot shells (shared between slots), in first-touch order
//      D  Dirichlet DoFs (boundary id 0 = whole boundary, ref:multigrid_throughput.cc:1585-1591)
//      H  hanging-node DoFs (own DoFs of fine faces/edges next to a coarser cell); like deal.II
//         they are part of the vector (n_dofs parity) but only ever see the identity row
//         (ref:include/operator.h:170-172).
//  * A cell with hanging faces/edges gathers, like deal.II's matrix-free hanging-node path, the
//    DoFs of the *parent* face/edge in the same local slot and then interpolates in-cell with
//    the 1D matrices FE1D::I[c]; its configuration is Tria::masks.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include "fe1d.hpp"
#include "octree.hpp"

#include <map>

namespace mgamd
{
  // sharded runs: who else references a DoF on an inter-rank interface (see partition.hpp)
  struct SharedInfo
  {
    uint64_t others  = 0; // other ranks referencing the DoF
    uint64_t regular = 0; // ranks (including this one) referencing it as a node of one of their own cells
  };
  // the owner is the lowest rank among the regular referencers: it is guaranteed to have a transfer patch for the DoF
  inline int
  shared_owner(const SharedInfo &si)
  {
    return __builtin_ctzll(si.regular);
  }
} // namespace mgamd

namespace mgamd
{
  constexpr uint32_t INVALID_DOF = 0xFFFFFFFFu;

  // max_brick = -1 ("auto"): levels below this many DoFs are latency-bound (a few microseconds per kernel whatever it
  // does).  If such a level needs more than one slot group, it uses single-cell slots only: one lattice launch per
  // operator application instead of one per brick size.  Measured on MI355X: -8 % V-cycle time for the octant at p=1
  // (17 M DoFs) and p=4 (17.5 M DoFs); uniform meshes (one brick group per level anyway) keep their bricks.
  constexpr size_t SMALL_LEVEL_DOFS = 500000;
  inline bool
  is_small_level(size_t n_cells, int p)
  {
    return n_cells * (size_t)(p * p * p) < SMALL_LEVEL_DOFS;
  }

  inline int
  max_brick_for_degree(int p)
  {
    int B = 1;
    while (2 * B * p + 1 <= 17)
      B *= 2;
    return B;
  }

  // position of lattice node (x,y,z) in the shell enumeration.  The shell is listed FACE BY FACE -- z=0, z=N-1,
  // then y=0, y=N-1 (without the z-face rows), then x=0, x=N-1 (without the y- and z-face rows) -- each face as a
  // contiguous 2D block in (outer, inner) lexicographic order.  Tail DoFs are numbered in first-touch order of
  // this enumeration, so a face shared by two bricks is ONE contiguous run of the tail segment for both of
  // them: shell gathers and shell atomics of a wave hit consecutive addresses (measured: 2.7x over-fetch with
  // the plain lattice-order enumeration).
  inline int
  shell_slot_of(int N, int x, int y, int z)
  {
    const int M = N - 2;
    if (z == 0)
      return y * N + x;
    if (z == N - 1)
      return N * N + y * N + x;
    int s = 2 * N * N;
    if (y == 0)
      return s + (z - 1) * N + x;
    s += M * N;
    if (y == N - 1)
      return s + (z - 1) * N + x;
    s += M * N;
    if (x == 0)
      return s + (z - 1) * M + (y - 1);
    s += M * M;
    return s + (z - 1) * M + (y - 1);
  }

  struct SlotGroup
  {
    int                   B = 1, N = 2, n_interior = 0, n_shell = 8;
    std::vector<uint32_t> interior_base; // per slot: global index of its first interior DoF
    std::vector<uint32_t> shell_idx;     // per slot x n_shell: global DoF or INVALID_DOF (Dirichlet)
    std::vector<uint16_t> mask;          // per slot: constraint configuration (0 for bricks)
    std::vector<uint32_t> fmask;         // bricks (B >= 2), per slot: hanging faces/edges of the whole brick (family_* helpers), 0 = none
    std::vector<double>   h;             // per slot: cell edge length
    std::vector<uint32_t> first_cell;    // per slot: index of its first cell in Tria::cells
    std::vector<uint16_t> shell_pos;     // n_shell: lattice index (z*N+y)*N+x of each shell entry
    // sharded levels: the first n_halo_slots slots of the group touch DoFs shared with other ranks; the halo exchange of an
    // operator application only needs them, so the others run underneath it (runtime.hip, apply_P)
    uint32_t n_halo_slots = 0;
    // p = 1, B > 2: the constrained bricks form a group of their own (their kernel carries the embedding passes; the
    // unconstrained bricks keep the lean kernel).  Families (B = 2) share the group of the 2^3 bricks.
    bool constrained_group = false;
    size_t
    n_slots() const
    {
      return interior_base.size();
    }
  };

  struct DofKey
  {
    int32_t px, py, pz, dirmask, level;
  };

  inline uint64_t
  pack_key(uint32_t px, uint32_t py, uint32_t pz, int dm, int level)
  {
    return ((uint64_t)px << 43) | ((uint64_t)py << 25) | ((uint64_t)pz << 7) | ((uint64_t)dm << 4) | (uint64_t)(dm ? level : 0);
  }
  inline DofKey
  unpack_key(uint64_t k)
  {
    return DofKey{(int32_t)(k >> 43), (int32_t)((k >> 25) & 0x3ffff), (int32_t)((k >> 7) & 0x3ffff), (int32_t)((k >> 4) & 7),
                  (int32_t)(k & 15)};
  }

  // in-cell hanging-node interpolation (and its transpose) on (p+1)^3 gathered values
  inline void
  interpolate_hanging(const FE1D &fe, uint16_t mask, double *v, bool transpose)
  {
    if (!(mask >> MASK_FACE_SHIFT))
      return;
    const int p = fe.p, n = p + 1;
    const int cp[3]     = {mask & 1, (mask >> 1) & 1, (mask >> 2) & 1};
    const int stride[3] = {1, n, n * n};
    double    tmp[MAX_DEGREE + 1];
    for (int dd = 0; dd < 3; ++dd)
      {
        const int     d = transpose ? 2 - dd : dd;
        const int     e = (d + 1) % 3, f = (d + 2) % 3;
        const bool    fe_c = (mask >> (MASK_FACE_SHIFT + e)) & 1, ff_c = (mask >> (MASK_FACE_SHIFT + f)) & 1,
                   ed_c    = (mask >> (MASK_EDGE_SHIFT + d)) & 1;
        const double *Ic   = fe.I[cp[d]].data();
        for (int ae = 0; ae < n; ++ae)
          for (int af = 0; af < n; ++af)
            {
              const bool on_e = ae == cp[e] * p, on_f = af == cp[f] * p;
              if (!((fe_c && on_e) || (ff_c && on_f) || (ed_c && on_e && on_f)))
                continue;
              double *line = v + ae * stride[e] + af * stride[f];
              for (int a = 0; a < n; ++a)
                {
                  double s = 0;
                  for (int b = 0; b < n; ++b)
                    s += (transpose ? Ic[b * n + a] : Ic[a * n + b]) * line[b * stride[d]];
                  tmp[a] = s;
                }
              for (int a = 0; a < n; ++a)
                line[a * stride[d]] = tmp[a];
            }
      }
  }

  // ---- families with hanging nodes as constrained 2^3 bricks -------------------------------------------------
  // The 8 children of one parent next to a coarser neighbour: the neighbour has the parent's size, so a hanging face of
  // the family IS one face of the parent cell and its (2p+1)^2 lattice nodes are E (x) E times the parent face's
  // (p+1)^2 DoFs, E = [I0; I1] the 1D two-cell embedding; a hanging edge likewise with one E.  The brick lattice stores
  // the parent DoFs ON the entity: parent coordinate c < p along a free direction at lattice coordinate c, c = p at the
  // far end 2p (so vertices and the nodes of edges shared by two hanging faces have ONE position, their own); the other
  // positions of the entity carry no DoF.  The embedding runs in place, direction by direction, before the sweeps and
  // its transpose after them.
  // Family mask: bit 2*d + side = face with normal d hangs; bit 6 + 4*d + s1 + 2*s2 = edge along d hangs, s1/s2 the
  // sides in directions (d+1)%3, (d+2)%3.
  inline bool
  family_face(uint32_t fm, int d, int side)
  {
    return (fm >> (2 * d + side)) & 1;
  }
  inline bool
  family_edge(uint32_t fm, int d, int s1, int s2)
  {
    return (fm >> (6 + 4 * d + s1 + 2 * s2)) & 1;
  }
  // lattice node n (coordinates 0..B p) of a constrained brick: does it lie on a hanging face/edge?  pinned[d]: the
  // coordinate is fixed by the hanging entity (its parent coordinate is 0 or p of the outermost parent cell), otherwise it
  // runs along the entity
  inline bool
  family_node_constrained(uint32_t fm, int p, int B, const int n[3], bool pinned[3])
  {
    const int N1 = B * p;
    bool      any = false;
    for (int d = 0; d < 3; ++d)
      {
        pinned[d] = (n[d] == 0 && family_face(fm, d, 0)) || (n[d] == N1 && family_face(fm, d, 1));
        any |= pinned[d];
      }
    if (any)
      return true;
    for (int d = 0; d < 3; ++d)
      {
        const int e = (d + 1) % 3, f = (d + 2) % 3;
        if ((n[e] == 0 || n[e] == N1) && (n[f] == 0 || n[f] == N1) && family_edge(fm, d, n[e] == N1, n[f] == N1))
          {
            pinned[e] = pinned[f] = true;
            return true;
          }
      }
    return false;
  }
  // Constrained BRICKS of any size (B = 2: a family).  A brick of B^3 equal cells next to coarser cells qualifies if the
  // hanging entities are WHOLE faces / WHOLE edges of the brick: a hanging brick face then is (B/2)^2 faces of parent-size
  // cells and carries the parents' (p B/2 + 1)^2 DoFs.  Along a lattice line the parent DoF k p + c (parent cell k, local
  // coordinate c) is stored at lattice coordinate 2 k p + c for c < p (c = p: the next parent cell's c = 0, or B p at the end);
  // the other positions of the entity carry no DoF and are filled by the in-place embedding E = [I0; I1] per parent cell.
  // mask of the brick from its cells: bpos[c] = position of cell c inside the brick, cm[c] = its cell mask; false if the
  // cells do not describe whole-face / whole-edge constraints
  inline bool
  brick_mask_from_cells(int B, int p, const std::vector<std::array<int, 3>> &bpos, const std::vector<uint16_t> &cm, uint32_t &fm)
  {
    fm = 0;
    const size_t nc = bpos.size();
    for (size_t c = 0; c < nc; ++c)
      {
        const int cp[3] = {bpos[c][0] & 1, bpos[c][1] & 1, bpos[c][2] & 1};
        if ((cm[c] & 7) != (cp[0] | (cp[1] << 1) | (cp[2] << 2)))
          return false;
        bool out[3]; // is the parent-side face of this cell in direction d on the brick boundary?
        for (int d = 0; d < 3; ++d)
          out[d] = bpos[c][d] == (cp[d] ? B - 1 : 0);
        for (int d = 0; d < 3; ++d)
          {
            if ((cm[c] >> (MASK_FACE_SHIFT + d)) & 1)
              {
                if (!out[d])
                  return false;
                fm |= 1u << (2 * d + cp[d]);
              }
            if ((cm[c] >> (MASK_EDGE_SHIFT + d)) & 1)
              {
                const int e = (d + 1) % 3, f = (d + 2) % 3;
                // an edge bit that only mirrors a hanging face of this cell needs no edge of its own
                const bool fe = (cm[c] >> (MASK_FACE_SHIFT + e)) & 1, ff = (cm[c] >> (MASK_FACE_SHIFT + f)) & 1;
                if (fe || ff)
                  continue;
                if (!out[e] || !out[f])
                  return false;
                fm |= 1u << (6 + 4 * d + cp[e] + 2 * cp[f]);
              }
          }
      }
    // every cell must see exactly the constraints the brick mask implies
    for (size_t c = 0; c < nc; ++c)
      {
        const int cp[3] = {bpos[c][0] & 1, bpos[c][1] & 1, bpos[c][2] & 1};
        for (int z = 0; z <= p; ++z)
          for (int y = 0; y <= p; ++y)
            for (int x = 0; x <= p; ++x)
              {
                const int  a[3] = {x, y, z};
                const int  n[3] = {bpos[c][0] * p + x, bpos[c][1] * p + y, bpos[c][2] * p + z};
                bool       pinned[3];
                const bool fam  = family_node_constrained(fm, p, B, n, pinned);
                bool       cell = false;
                if (cm[c] >> MASK_FACE_SHIFT)
                  {
                    bool on[3];
                    for (int d = 0; d < 3; ++d)
                      on[d] = a[d] == cp[d] * p;
                    for (int d = 0; d < 3; ++d)
                      cell |= (((cm[c] >> (MASK_FACE_SHIFT + d)) & 1) && on[d]) ||
                              (((cm[c] >> (MASK_EDGE_SHIFT + d)) & 1) && on[(d + 1) % 3] && on[(d + 2) % 3]);
                  }
                if (fam != cell)
                  return false;
              }
      }
    return true;
  }
  // lattice coordinate -> does a parent DoF live there along a free direction, and which (parent cell k, local c)?
  inline bool
  family_parent_position(int p, int B, int n, int &k, int &c)
  {
    if (n == B * p)
      {
        k = B / 2 - 1;
        c = p;
        return true;
      }
    k = n / (2 * p);
    c = n % (2 * p);
    return c < p;
  }
  // the embedding (transpose = false) or its transpose on the (B p + 1)^3 lattice of a constrained brick, host version
  inline void
  interpolate_family(const FE1D &fe, int B, uint32_t fm, double *v, bool transpose)
  {
    if (!fm)
      return;
    const int p = fe.p, n = p + 1, N = B * p + 1, BC = B / 2;
    const int stride[3] = {1, N, N * N};
    std::vector<double> in(N), out(N);
    auto      E = [&](int a, int b) { return a <= p ? fe.I[0][a * n + b] : fe.I[1][(a - p) * n + b]; };
    for (int dd = 0; dd < 3; ++dd)
      {
        const int d = transpose ? 2 - dd : dd, e = (d + 1) % 3, f = (d + 2) % 3;
        for (int ae = 0; ae < N; ++ae)
          for (int af = 0; af < N; ++af)
            {
              const bool xe = ae == 0 || ae == N - 1, xf = af == 0 || af == N - 1;
              const bool on = (ae == 0 && family_face(fm, e, 0)) || (ae == N - 1 && family_face(fm, e, 1)) ||
                              (af == 0 && family_face(fm, f, 0)) || (af == N - 1 && family_face(fm, f, 1)) ||
                              (xe && xf && family_edge(fm, d, ae == N - 1, af == N - 1));
              if (!on)
                continue;
              double *line = v + ae * stride[e] + af * stride[f];
              for (int i = 0; i < N; ++i)
                in[i] = line[i * stride[d]];
              for (int i = 0; i < N; ++i)
                out[i] = 0;
              for (int k = 0; k < BC; ++k)
                {
                  auto pos = [&](int b) { return 2 * k * p + (b < p ? b : 2 * p); }; // parent DoF b of parent cell k
                  if (!transpose)
                    for (int a = 0; a <= 2 * p; ++a)
                      {
                        double s = 0;
                        for (int b = 0; b < n; ++b)
                          s += E(a, b) * in[pos(b)];
                        out[2 * k * p + a] = s; // the node shared with the next parent cell gets the same value twice
                      }
                  else
                    for (int a = (k == 0 ? 0 : 1); a <= 2 * p; ++a) // a fine node shared by two parent cells counts once
                      for (int b = 0; b < n; ++b)
                        out[pos(b)] += E(a, b) * in[2 * k * p + a];
                }
              for (int i = 0; i < N; ++i)
                line[i * stride[d]] = out[i];
            }
      }
  }

  class LevelTables
  {
  public:
    int                    p = 1;
    const Tria            *tria = nullptr;
    FE1D                   fe;
    std::vector<SlotGroup> groups; // one per brick size, largest first; last = single cells
    uint32_t               n_dofs = 0, n_interior = 0, n_tail = 0, n_dirichlet = 0, n_hanging = 0;
    // Local smoothing (`HMG-local`): a LEVEL of the refinement hierarchy = all cells of one refinement level, which cover
    // only part of the domain.  The DoFs on the boundary of that region which are not on the domain boundary are the
    // REFINEMENT-EDGE DoFs (MGTools::extract_inner_interface_dofs, ref:include/operator.h:49-70,539-556): numbered
    // [ I | T | E | D | H ].  The level operator treats E like D (zero input, identity row); the edge matrices treat them
    // like T.  Their shell entries keep the real index: the kernels compare it with a gather / scatter LIMIT (D entries
    // are INVALID = 0xFFFFFFFF, above every limit).
    bool     ls_level = false;
    uint32_t n_edge   = 0;
    static constexpr int      CLS_SHIFT = 29;
    static constexpr uint32_t CLS_MASK  = (1u << CLS_SHIFT) - 1;
    // distributed runs: the tail is [owned | copies of DoFs owned by a lower rank]; *_owned count each DoF once globally
    uint32_t n_tail_owned = 0, n_dirichlet_owned = 0, n_hanging_owned = 0;
    // per cell: group, slot-in-group
    std::vector<uint8_t>  cell_group;
    std::vector<uint32_t> cell_slot;
    FlatMap               keymap; // packed key -> global index (T, D, H DoFs only)

    // owned: optional per-cell flags (distributed runs: this rank's cells); shared: key -> mask of other sharing ranks
    LevelTables(const Tria &t, int degree, int max_brick = 0, const std::vector<uint8_t> *owned = nullptr, bool helpers_only = false,
                const std::map<uint64_t, SharedInfo> *shared = nullptr, int my_rank = 0, bool local_smoothing_level = false)
      : p(degree)
      , tria(&t)
      , fe(degree)
      , ls_level(local_smoothing_level)
      , owned(owned)
      , shared(shared)
      , my_rank(my_rank)
    {
      if (ls_level && (owned || shared))
        throw std::runtime_error("local-smoothing levels are not sharded in this build");
      if (!helpers_only)
        build(max_brick > 0 ? std::min(max_brick, max_brick_for_degree(p)) : max_brick_for_degree(p));
    }
    bool
    cell_is_local(size_t ci) const
    {
      return cell_group[ci] < 0xFE;
    }

    uint32_t
    first_constrained() const
    {
      return n_interior + n_tail; // refinement-edge DoFs count as constrained for the level operator
    }
    uint32_t
    first_dirichlet() const
    {
      return n_interior + n_tail + n_edge;
    }

    // is local node (a,b,c) of a masked cell on one of its hanging faces/edges?
    static bool
    node_on_constrained_entity(uint16_t mask, int p, const int a[3], bool *parent_corner = nullptr)
    {
      const int cp[3] = {mask & 1, (mask >> 1) & 1, (mask >> 2) & 1};
      bool      on[3];
      for (int d = 0; d < 3; ++d)
        on[d] = a[d] == cp[d] * p;
      if (parent_corner)
        *parent_corner = on[0] && on[1] && on[2];
      for (int d = 0; d < 3; ++d)
        {
          if (((mask >> (MASK_FACE_SHIFT + d)) & 1) && on[d])
            return true;
          if (((mask >> (MASK_EDGE_SHIFT + d)) & 1) && on[(d + 1) % 3] && on[(d + 2) % 3])
            return true;
        }
      return false;
    }

    // key of the DoF that cell `ci` gathers at local node a (parent-resolved for hanging entities)
    uint64_t
    resolved_key(size_t ci, const int a[3]) const
    {
      const Cell    &c    = tria->cells[ci];
      const uint16_t mask = tria->masks[ci];
      if ((mask >> MASK_FACE_SHIFT) && node_on_constrained_entity(mask, p, a))
        return own_key(Cell{c.i >> 1, c.j >> 1, c.k >> 1, (uint8_t)(c.level - 1)}, a);
      return own_key(c, a);
    }
    uint64_t
    own_key(const Cell &c, const int a[3]) const
    {
      const uint32_t S  = 1u << (LMAX - c.level);
      const int      dm = (a[0] % p ? 1 : 0) | (a[1] % p ? 2 : 0) | (a[2] % p ? 4 : 0);
      return pack_key((c.i * p + a[0]) * S, (c.j * p + a[1]) * S, (c.k * p + a[2]) * S, dm, c.level);
    }
    bool
    key_on_boundary(uint64_t key) const
    {
      const DofKey   k   = unpack_key(key);
      const int32_t  top = p << LMAX;
      return k.px == 0 || k.py == 0 || k.pz == 0 || k.px == top || k.py == top || k.pz == top;
    }

    // global DoF index gathered by cell ci at local node (a0,a1,a2); INVALID_DOF for Dirichlet.
    // *constrained: node lies on a hanging face/edge (index then refers to the parent entity's DoF)
    uint32_t
    cell_node_index(size_t ci, const int a[3], bool *constrained = nullptr, bool *parent_corner = nullptr) const
    {
      if (!cell_is_local(ci))
        {
          if (constrained)
            *constrained = false;
          if (parent_corner)
            *parent_corner = false;
          return INVALID_DOF;
        }
      const SlotGroup &g    = groups[cell_group[ci]];
      const uint32_t   s    = cell_slot[ci];
      const Cell      &c    = tria->cells[ci];
      const uint16_t   mask = tria->masks[ci];
      if (constrained)
        *constrained = (mask >> MASK_FACE_SHIFT) ? node_on_constrained_entity(mask, p, a, parent_corner) : false;
      if (g.B >= 2 && g.fmask[s] && (mask >> MASK_FACE_SHIFT) && node_on_constrained_entity(mask, p, a))
        {
          // cell of a constrained brick: the parent entity's DoF, as for a single cell
          const int32_t *idx = keymap.find(resolved_key(ci, a));
          if (!idx)
            throw std::runtime_error("constrained family: parent DoF not numbered");
          const uint32_t gi = (uint32_t)*idx;
          return (gi >= first_dirichlet() && gi < first_dirichlet() + n_dirichlet) ? INVALID_DOF : gi;
        }
      const int x = (c.i & (g.B - 1)) * p + a[0], y = (c.j & (g.B - 1)) * p + a[1], z = (c.k & (g.B - 1)) * p + a[2];
      const int N = g.N;
      if (x > 0 && y > 0 && z > 0 && x < N - 1 && y < N - 1 && z < N - 1)
        return g.interior_base[s] + ((z - 1) * (N - 2) + (y - 1)) * (N - 2) + (x - 1);
      return g.shell_idx[(size_t)s * g.n_shell + shell_slot_of(N, x, y, z)];
    }

    // full per-cell table (resolved indices, x fastest), for the CPU oracle and tests
    void
    export_cell_dofs(std::vector<uint32_t> &out) const
    {
      const int n = p + 1;
      out.resize(tria->cells.size() * n * n * n);
      size_t t = 0;
      for (size_t ci = 0; ci < tria->cells.size(); ++ci)
        for (int c = 0; c < n; ++c)
          for (int b = 0; b < n; ++b)
            for (int a = 0; a < n; ++a)
              {
                const int l[3] = {a, b, c};
                out[t++]       = cell_node_index(ci, l);
              }
    }

    // geometric identity of every DoF (for matching against an independently numbered oracle)
    void
    export_dof_keys(std::vector<DofKey> &out) const
    {
      out.assign(n_dofs, DofKey{-1, -1, -1, -1, -1});
      for (const SlotGroup &g : groups)
        for (size_t s = 0; s < g.n_slots(); ++s)
          {
            const Cell &c = tria->cells[g.first_cell[s]];
            const Cell  anchor{c.i & ~(uint32_t)(g.B - 1), c.j & ~(uint32_t)(g.B - 1), c.k & ~(uint32_t)(g.B - 1), c.level};
            const int   N = g.N;
            for (int z = 1; z < N - 1; ++z)
              for (int y = 1; y < N - 1; ++y)
                for (int x = 1; x < N - 1; ++x)
                  {
                    const int a[3] = {x, y, z};
                    out[g.interior_base[s] + ((z - 1) * (N - 2) + (y - 1)) * (N - 2) + (x - 1)] = unpack_key(own_key(anchor, a));
                  }
          }
      keymap_for_each([&](uint64_t key, int32_t idx) {
        DofKey k = unpack_key(key);
        out[idx] = k;
      });
    }

    // right-hand side for f == 1, homogeneous Dirichlet data (ref:include/operator.h:362-413;
    // the lifting of ref:include/operator.h:415-446 vanishes for g == 0)
    void
    compute_rhs_constant(std::vector<double> &b) const
    {
      b.assign(n_dofs, 0.0);
      const int n = p + 1;
      for (const SlotGroup &g : groups)
        {
          const int           N = g.N;
          std::vector<double> m1(N, 0.0); // assembled 1D load vector over the brick line
          for (int cb = 0; cb < g.B; ++cb)
            for (int a = 0; a < n; ++a)
              m1[cb * p + a] += fe.m[a];
          std::vector<double> loc((size_t)N * N * N);
          for (size_t s = 0; s < g.n_slots(); ++s)
            {
              const double h3 = g.h[s] * g.h[s] * g.h[s];
              for (int z = 0; z < N; ++z)
                for (int y = 0; y < N; ++y)
                  for (int x = 0; x < N; ++x)
                    loc[(z * N + y) * N + x] = h3 * m1[x] * m1[y] * m1[z];
              if (g.B == 1)
                interpolate_hanging(fe, g.mask[s], loc.data(), true);
              else if (g.B >= 2 && g.fmask[s])
                interpolate_family(fe, g.B, g.fmask[s], loc.data(), true);
              for (int z = 0; z < N; ++z)
                for (int y = 0; y < N; ++y)
                  for (int x = 0; x < N; ++x)
                    {
                      uint32_t idx;
                      if (x > 0 && y > 0 && z > 0 && x < N - 1 && y < N - 1 && z < N - 1)
                        idx = g.interior_base[s] + ((z - 1) * (N - 2) + (y - 1)) * (N - 2) + (x - 1);
                      else
                        idx = g.shell_idx[s * g.n_shell + shell_slot_of(N, x, y, z)];
                      if (idx != INVALID_DOF)
                        b[idx] += loc[(z * N + y) * N + x];
                    }
            }
        }
    }

    // ---- general data: SimulationType "Gaussian" (ref:multigrid_throughput.cc:60-125,2294-2298) -------------------
    // u_exact = sum over centres of exp(-|x - c|^2 / w^2) / (sqrt(2 pi) w)^3, f = -Laplace u_exact; the reference uses
    // one centre (-0.5,-0.5,-0.5) and w = 0.1.  kind 0: f = 1, g = 0 (the default "Constant").
    static double
    gaussian_solution(const double x[3])
    {
      const double w = 0.1, c[3] = {-0.5, -0.5, -0.5};
      double       r2 = 0;
      for (int d = 0; d < 3; ++d)
        r2 += (x[d] - c[d]) * (x[d] - c[d]);
      const double s = std::sqrt(2.0 * 3.14159265358979323846) * w;
      return std::exp(-r2 / (w * w)) / (s * s * s);
    }
    static double
    gaussian_rhs(const double x[3])
    {
      const double w = 0.1, c[3] = {-0.5, -0.5, -0.5};
      double       r2 = 0;
      for (int d = 0; d < 3; ++d)
        r2 += (x[d] - c[d]) * (x[d] - c[d]);
      const double s = std::sqrt(2.0 * 3.14159265358979323846) * w;
      return (2.0 * 3 - 4.0 * r2 / (w * w)) / (w * w) * std::exp(-r2 / (w * w)) / (s * s * s);
    }
    static double
    data_f(int kind, const double x[3])
    {
      return kind == 0 ? 1.0 : gaussian_rhs(x);
    }
    static double
    data_g(int kind, const double x[3])
    {
      return kind == 0 ? 0.0 : gaussian_solution(x);
    }
    // position of the DoF that cell ci gathers at local node a (the parent's node on hanging entities)
    void
    gathered_node_position(size_t ci, const int a[3], double x[3]) const
    {
      const Cell    &c    = tria->cells[ci];
      const uint16_t mask = tria->masks[ci];
      const bool     par  = (mask >> MASK_FACE_SHIFT) && node_on_constrained_entity(mask, p, a);
      const uint32_t ijk[3] = {c.i, c.j, c.k};
      const int      lev  = par ? c.level - 1 : c.level;
      const double   h    = 2.0 / (double)(1u << lev);
      for (int d = 0; d < 3; ++d)
        x[d] = -1.0 + h * ((double)(par ? ijk[d] >> 1 : ijk[d]) + fe.nodes[a[d]]);
    }

    // y = K_cell x on one cell's (p+1)^3 values (x fastest), h = cell size
    void
    cell_stiffness_apply(double h, const double *x, double *y) const
    {
      const int           n = p + 1, n3 = n * n * n;
      std::vector<double> t1(n3), t2(n3), t3(n3);
      auto                sweep = [&](const std::vector<double> &A, int d, const double *in, double *out) {
        const int st = d == 0 ? 1 : (d == 1 ? n : n * n);
        for (int i = 0; i < n3; ++i)
          {
            const int id = (i / st) % n;
            double    s  = 0;
            for (int b = 0; b < n; ++b)
              s += A[id * n + b] * in[i + (b - id) * st];
            out[i] = s;
          }
      };
      for (int i = 0; i < n3; ++i)
        y[i] = 0;
      for (int dk = 0; dk < 3; ++dk)
        { // K in direction dk, M in the others
          sweep(dk == 0 ? fe.K : fe.M, 0, x, t1.data());
          sweep(dk == 1 ? fe.K : fe.M, 1, t1.data(), t2.data());
          sweep(dk == 2 ? fe.K : fe.M, 2, t2.data(), t3.data());
          for (int i = 0; i < n3; ++i)
            y[i] += h * t3[i];
        }
    }

    // right-hand side for data `kind` (ref:include/operator.h:362-447): load vector by QGauss(p+1) quadrature of f,
    // minus the operator without Dirichlet constraints applied to the boundary interpolant of g; constrained rows 0.
    void
    compute_rhs_function(int kind, std::vector<double> &b) const
    {
      if (kind == 0)
        {
          compute_rhs_constant(b);
          return;
        }
      b.assign(n_dofs, 0.0);
      const int           n = p + 1, n3 = n * n * n;
      std::vector<double> fq(n3), t1(n3), t2(n3), load(n3), xg(n3), lift(n3);
      for (size_t ci = 0; ci < tria->cells.size(); ++ci)
        {
          if (!cell_is_local(ci))
            continue;
          const Cell    &c    = tria->cells[ci];
          const uint16_t mask = tria->masks[ci];
          const double   h    = 2.0 / (double)(1u << c.level);
          const double   o[3] = {-1.0 + h * c.i, -1.0 + h * c.j, -1.0 + h * c.k};
          // f at the quadrature points, times JxW
          for (int qz = 0; qz < n; ++qz)
            for (int qy = 0; qy < n; ++qy)
              for (int qx = 0; qx < n; ++qx)
                {
                  const double x[3] = {o[0] + h * fe.xq[qx], o[1] + h * fe.xq[qy], o[2] + h * fe.xq[qz]};
                  fq[(qz * n + qy) * n + qx] = data_f(kind, x) * h * h * h * fe.wq[qx] * fe.wq[qy] * fe.wq[qz];
                }
          // integrate against the shape functions: load[a] = sum_q S[q][a] ... per direction
          auto integrate = [&](int d, const double *in, double *out) {
            const int st = d == 0 ? 1 : (d == 1 ? n : n * n);
            for (int i = 0; i < n3; ++i)
              {
                const int id = (i / st) % n;
                double    s  = 0;
                for (int q = 0; q < n; ++q)
                  s += fe.S[q * n + id] * in[i + (q - id) * st];
                out[i] = s;
              }
          };
          integrate(0, fq.data(), t1.data());
          integrate(1, t1.data(), t2.data());
          integrate(2, t2.data(), load.data());
          // boundary interpolant of g on this cell's gathered DoFs, operator without Dirichlet constraints
          bool any_g = false;
          uint32_t idx[512];
          for (int z = 0; z < n; ++z)
            for (int y = 0; y < n; ++y)
              for (int x = 0; x < n; ++x)
                {
                  const int a[3] = {x, y, z};
                  const int t    = (z * n + y) * n + x;
                  idx[t]         = cell_node_index(ci, a);
                  xg[t]          = 0.0;
                  if (idx[t] == INVALID_DOF)
                    {
                      double pos[3];
                      gathered_node_position(ci, a, pos);
                      xg[t] = data_g(kind, pos);
                      any_g |= xg[t] != 0.0;
                    }
                }
          if (any_g)
            {
              interpolate_hanging(fe, mask, xg.data(), false);
              cell_stiffness_apply(h, xg.data(), lift.data());
              for (int t = 0; t < n3; ++t)
                load[t] -= lift[t];
            }
          interpolate_hanging(fe, mask, load.data(), true);
          for (int t = 0; t < n3; ++t)
            if (idx[t] != INVALID_DOF)
              b[idx[t]] += load[t];
        }
      for (uint32_t i = first_constrained(); i < n_dofs; ++i)
        b[i] = 0.0;
    }

    // constraints.distribute(x) for data `kind`: Dirichlet DoFs = g at their support point, hanging-node DoFs = the
    // interpolant of their parent face/edge (x holds the solved free DoFs)
    void
    distribute(int kind, std::vector<double> &x) const
    {
      const int           n = p + 1, n3 = n * n * n;
      std::vector<double> v(n3);
      // Dirichlet values first (hanging nodes may depend on them)
      for (size_t ci = 0; ci < tria->cells.size(); ++ci)
        if (cell_is_local(ci))
          for (int z = 0; z < n; ++z)
            for (int y = 0; y < n; ++y)
              for (int xx = 0; xx < n; ++xx)
                {
                  const int      a[3] = {xx, y, z};
                  const uint64_t key  = resolved_key(ci, a);
                  if (!key_on_boundary(key))
                    continue;
                  const int32_t *gi = keymap.find(key);
                  if (gi && (uint32_t)*gi >= first_dirichlet() && (uint32_t)*gi < first_dirichlet() + n_dirichlet)
                    {
                      double pos[3];
                      gathered_node_position(ci, a, pos);
                      x[*gi] = data_g(kind, pos);
                    }
                }
      for (size_t ci = 0; ci < tria->cells.size(); ++ci)
        {
          const uint16_t mask = tria->masks[ci];
          if (!cell_is_local(ci) || !(mask >> MASK_FACE_SHIFT))
            continue;
          for (int z = 0; z < n; ++z)
            for (int y = 0; y < n; ++y)
              for (int xx = 0; xx < n; ++xx)
                {
                  const int      a[3] = {xx, y, z};
                  const int32_t *gi   = keymap.find(resolved_key(ci, a));
                  uint32_t       id   = cell_node_index(ci, a);
                  if (id == INVALID_DOF && gi)
                    id = (uint32_t)*gi; // Dirichlet DoF: its value was set above
                  v[(z * n + y) * n + xx] = id != INVALID_DOF ? x[id] : 0.0;
                }
          interpolate_hanging(fe, mask, v.data(), false);
          for (int z = 0; z < n; ++z)
            for (int y = 0; y < n; ++y)
              for (int xx = 0; xx < n; ++xx)
                {
                  const int a[3] = {xx, y, z};
                  bool      corner;
                  if (node_on_constrained_entity(mask, p, a, &corner) && !corner)
                    if (const int32_t *own = keymap.find(own_key(tria->cells[ci], a)))
                      x[*own] = v[(z * n + y) * n + xx];
                }
        }
    }

    template <class F>
    void
    keymap_for_each(F f) const
    {
      for (size_t t = 0; t < key_list.size(); ++t)
        f(key_list[t], key_index[t]);
    }

  private:
    std::vector<uint64_t>               key_list;  // keys of T/D/H DoFs in creation order
    std::vector<int32_t>                key_index; // their final global indices
    const std::vector<uint8_t>         *owned   = nullptr;
    const std::map<uint64_t, SharedInfo> *shared  = nullptr;
    int                                 my_rank = 0;

    void
    build(int Bmax)
    {
      const auto  &cells = tria->cells;
      const auto  &masks = tria->masks;
      const size_t nc    = cells.size();
      // ---- 1. slot decomposition
      std::vector<int> sizes;
      // at p = 1 the 2^3 and 4^3 bricks are left to the single-cell cluster kernel, which is faster per cell than the
      // lattice kernel on small lattices and saves a launch per size and application (octant, 17 M DoFs, measured:
      // sizes {16,8,4,2,1} 2.93 ms per V-cycle, {16,8,4,1} 2.66, later {16,8,4,1} 2.31, {16,8,1} 2.23, {16,1} 2.29)
      const bool hanging_bricks = getenv("MGAMD_NO_HANGING_BRICKS") == nullptr;
      // MGAMD_MAX_CONSTRAINED_BRICK=2: only families (the 8 children of one cell) may be constrained bricks (development A/B)
      // (larger constrained bricks only at p = 1: kernels.hpp brick_may_be_constrained, with the measurements)
      const int  max_constrained_brick =
        getenv("MGAMD_MAX_CONSTRAINED_BRICK") ? std::max(2, atoi(getenv("MGAMD_MAX_CONSTRAINED_BRICK"))) : (p != 1 ? 2 : 1 << 30);
      int        skip           = p == 1 ? 6 : 0;
      if (const char *e = getenv("MGAMD_SKIP_BRICKS")) // development: bit mask of brick sizes to leave out
        skip = atoi(e);
      std::vector<bool> constrained_only;
      for (int B = Bmax; B >= 1; B /= 2)
        if (B == 1 || B == Bmax || !(skip & B))
          {
            sizes.push_back(B);
            constrained_only.push_back(false);
            if (B > 2 && hanging_bricks && B <= max_constrained_brick)
              { // second group of this size for the constrained bricks
                sizes.push_back(B);
                constrained_only.push_back(true);
              }
          }
      groups.resize(sizes.size());
      cell_group.assign(nc, 0xFF);
      cell_slot.assign(nc, 0);
      if (owned)
        for (size_t t = 0; t < nc; ++t)
          if (!(*owned)[t])
            cell_group[t] = 0xFE; // not ours
      for (size_t gi = 0; gi < sizes.size(); ++gi)
        {
          SlotGroup &g = groups[gi];
          g.B                 = sizes[gi];
          g.constrained_group = constrained_only[gi];
          g.N          = p * g.B + 1;
          g.n_interior = (g.N - 2) * (g.N - 2) * (g.N - 2);
          g.n_shell    = g.N * g.N * g.N - g.n_interior;
          g.shell_pos.resize(g.n_shell);
          for (int z = 0; z < g.N; ++z)
            for (int y = 0; y < g.N; ++y)
              for (int x = 0; x < g.N; ++x)
                if (!(x > 0 && y > 0 && z > 0 && x < g.N - 1 && y < g.N - 1 && z < g.N - 1))
                  g.shell_pos[shell_slot_of(g.N, x, y, z)] = (uint16_t)((z * g.N + y) * g.N + x);
          const int    B  = g.B;
          const size_t B3 = (size_t)B * B * B;
          int          b  = 0;
          while ((1 << b) < B)
            ++b;
          size_t t = 0;
          while (t < nc)
            {
              if (cell_group[t] != 0xFF)
                {
                  ++t;
                  continue;
                }
              const Cell &c = cells[t];
              bool        ok;
              uint32_t    fm = 0;
              if (B == 1)
                ok = true;
              else
                {
                  ok = c.level >= b && !(c.i & (B - 1)) && !(c.j & (B - 1)) && !(c.k & (B - 1)) && t + B3 <= nc;
                  bool hanging = false;
                  for (size_t s = 0; ok && s < B3; ++s)
                    {
                      const Cell &d = cells[t + s];
                      ok = d.level == c.level && (d.i >> b) == (c.i >> b) && (d.j >> b) == (c.j >> b) && (d.k >> b) == (c.k >> b) &&
                           cell_group[t + s] == 0xFF;
                      hanging |= (masks[t + s] >> MASK_FACE_SHIFT) != 0;
                    }
                  if (ok && hanging)
                    {
                      // a brick whose hanging entities are whole faces / whole edges stays a (constrained) brick
                      ok = false;
                      if (hanging_bricks && B <= max_constrained_brick && c.level >= 1)
                        {
                          std::vector<std::array<int, 3>> bpos(B3);
                          std::vector<uint16_t>           cm(B3);
                          for (size_t s2 = 0; s2 < B3; ++s2)
                            {
                              const Cell &d = cells[t + s2];
                              bpos[s2]      = {(int)(d.i - c.i), (int)(d.j - c.j), (int)(d.k - c.k)};
                              cm[s2]        = masks[t + s2];
                            }
                          ok = brick_mask_from_cells(B, p, bpos, cm, fm);
                        }
                    }
                  // B > 2: constrained bricks only into the constrained group of this size, the others into the plain one
                  // (the plain group is scanned first; an unclaimed constrained brick is picked up by the next group)
                  if (ok && B > 2 && (fm != 0) != g.constrained_group)
                    ok = false;
                }
              if (!ok)
                {
                  ++t;
                  continue;
                }
              const uint32_t slot = (uint32_t)g.first_cell.size();
              g.first_cell.push_back((uint32_t)t);
              g.mask.push_back(B == 1 ? masks[t] : 0);
              if (B >= 2)
                g.fmask.push_back(fm);
              g.h.push_back(2.0 / (double)(1u << c.level));
              for (size_t s = 0; s < B3; ++s)
                {
                  cell_group[t + s] = (uint8_t)gi;
                  cell_slot[t + s]  = slot;
                }
              t += B3;
            }
          g.interior_base.assign(g.first_cell.size(), 0);
          g.shell_idx.assign(g.first_cell.size() * (size_t)g.n_shell, INVALID_DOF);
        }
      // ---- 2. interior numbering, slots in Morton order of their first cell
      struct Ref
      {
        uint32_t first_cell, group, slot;
      };
      std::vector<Ref> order;
      for (size_t gi = 0; gi < groups.size(); ++gi)
        for (size_t s = 0; s < groups[gi].n_slots(); ++s)
          order.push_back(Ref{groups[gi].first_cell[s], (uint32_t)gi, (uint32_t)s});
      std::sort(order.begin(), order.end(), [](const Ref &a, const Ref &b) { return a.first_cell < b.first_cell; });
      uint64_t next = 0;
      for (const Ref &r : order)
        {
          groups[r.group].interior_base[r.slot] = (uint32_t)next;
          next += groups[r.group].n_interior;
        }
      if (next > 0xFFFFFFF0ull)
        throw std::runtime_error("level exceeds 32-bit DoF indices");
      n_interior = (uint32_t)next;
      // ---- 3. shell DoFs: class 0 tail, 1 Dirichlet, 2 hanging; provisional id = (class<<30 | counter)
      keymap.erase_all_and_reserve(1024);
      // local smoothing: keys of the DoFs on faces of level cells whose neighbour position is inside the domain but not
      // covered by a cell of this level
      FlatMap edge_keys;
      edge_keys.erase_all_and_reserve(1024);
      if (ls_level)
        for (size_t ci = 0; ci < nc; ++ci)
          {
            const Cell   &c     = cells[ci];
            const int64_t n1    = (int64_t)1 << c.level;
            const int64_t id[3] = {c.i, c.j, c.k};
            for (int d = 0; d < 3; ++d)
              for (int side = 0; side < 2; ++side)
                {
                  int64_t nb[3] = {id[0], id[1], id[2]};
                  nb[d] += side ? 1 : -1;
                  if (nb[d] < 0 || nb[d] >= n1 || tria->find_leaf(c.level, nb[0], nb[1], nb[2]) >= 0)
                    continue;
                  const int e = (d + 1) % 3, f = (d + 2) % 3;
                  for (int ae = 0; ae <= p; ++ae)
                    for (int af = 0; af <= p; ++af)
                      {
                        int a[3];
                        a[d] = side * p;
                        a[e] = ae;
                        a[f] = af;
                        edge_keys.insert(own_key(c, a), 1);
                      }
                }
          }
      uint32_t counter[5] = {0, 0, 0, 0, 0}; // 0 tail (owned), 1 Dirichlet, 2 hanging, 3 tail copy of a lower rank's DoF, 4 refinement edge
      uint32_t n_copy_d = 0, n_copy_h = 0;
      auto     classify = [&](uint64_t key, int cls) -> int32_t {
        bool     ins;
        int32_t *v = keymap.insert(key, 0, &ins);
        if (ins)
          {
            if (shared)
              {
                auto it = shared->find(key);
                if (it != shared->end() && shared_owner(it->second) != my_rank)
                  { // another rank owns this DoF (the lowest rank that references it as a node of one of its cells)
                    if (cls == 0)
                      cls = 3;
                    else if (cls == 1)
                      ++n_copy_d;
                    else
                      ++n_copy_h;
                  }
              }
            *v = (int32_t)(((uint32_t)cls << CLS_SHIFT) | counter[cls]++);
            key_list.push_back(key);
          }
        return *v;
      };
      for (const Ref &r : order)
        {
          SlotGroup   &g = groups[r.group];
          const size_t ci = g.first_cell[r.slot];
          const Cell  &c  = cells[ci];
          const Cell   anchor{c.i & ~(uint32_t)(g.B - 1), c.j & ~(uint32_t)(g.B - 1), c.k & ~(uint32_t)(g.B - 1), c.level};
          for (int s = 0; s < g.n_shell; ++s)
            {
              const int lin  = g.shell_pos[s];
              const int a[3] = {lin % g.N, (lin / g.N) % g.N, lin / (g.N * g.N)};
              uint64_t key;
              if (g.B == 1)
                key = resolved_key(ci, a);
              else if (g.B >= 2 && g.fmask[r.slot])
                {
                  // hanging face/edge of a constrained brick: the parent DoF k p + c of parent cell k sits at lattice
                  // coordinate 2 k p + c (c < p; c = p: B p at the far end); the other lattice positions carry no DoF (filled
                  // by the embedding)
                  bool pinned[3];
                  if (family_node_constrained(g.fmask[r.slot], p, g.B, a, pinned))
                    {
                      int  ap[3], kc[3];
                      bool dof = true;
                      for (int d = 0; d < 3; ++d)
                        {
                          if (pinned[d])
                            {
                              kc[d] = a[d] ? g.B / 2 - 1 : 0;
                              ap[d] = a[d] ? p : 0;
                            }
                          else
                            dof &= family_parent_position(p, g.B, a[d], kc[d], ap[d]);
                        }
                      if (!dof)
                        {
                          g.shell_idx[(size_t)r.slot * g.n_shell + s] = INVALID_DOF;
                          continue;
                        }
                      key = own_key(Cell{(anchor.i >> 1) + (uint32_t)kc[0], (anchor.j >> 1) + (uint32_t)kc[1], (anchor.k >> 1) + (uint32_t)kc[2],
                                         (uint8_t)(c.level - 1)},
                                    ap);
                    }
                  else
                    key = own_key(anchor, a);
                }
              else
                key = own_key(anchor, a);
              const int32_t id = classify(key, key_on_boundary(key) ? 1 : ((ls_level && edge_keys.find(key)) ? 4 : 0));
              g.shell_idx[(size_t)r.slot * g.n_shell + s] = (uint32_t)id; // provisional
            }
        }
      // own DoFs of hanging faces/edges (single cells and cells of constrained families alike)
      for (size_t ci = 0; ci < nc; ++ci)
        if (cell_is_local(ci) && (masks[ci] >> MASK_FACE_SHIFT))
          for (int z = 0; z <= p; ++z)
            for (int y = 0; y <= p; ++y)
              for (int x = 0; x <= p; ++x)
                {
                  const int a[3] = {x, y, z};
                  bool      corner;
                  if (node_on_constrained_entity(masks[ci], p, a, &corner) && !corner)
                    {
                      const uint64_t key = own_key(cells[ci], a);
                      if (!keymap.find(key))
                        classify(key, 2);
                    }
                }
      n_tail_owned      = counter[0];
      n_tail            = counter[0] + counter[3];
      n_dirichlet       = counter[1];
      n_hanging         = counter[2];
      n_edge            = counter[4];
      n_dirichlet_owned = n_dirichlet - n_copy_d;
      n_hanging_owned   = n_hanging - n_copy_h;
      const uint64_t total = (uint64_t)n_interior + n_tail + n_edge + n_dirichlet + n_hanging;
      if (total > 0xFFFFFFF0ull)
        throw std::runtime_error("level exceeds 32-bit DoF indices");
      n_dofs                 = (uint32_t)total;
      const uint32_t base[5] = {n_interior, n_interior + n_tail + n_edge, n_interior + n_tail + n_edge + n_dirichlet, n_interior + n_tail_owned,
                                n_interior + n_tail};
      auto           final_index = [&](uint32_t prov) {
        const uint32_t c = prov & CLS_MASK;
        return base[prov >> CLS_SHIFT] + c;
      };
      key_index.resize(key_list.size());
      for (size_t t = 0; t < key_list.size(); ++t)
        {
          int32_t *v   = keymap.find(key_list[t]);
          *v           = (int32_t)final_index((uint32_t)*v);
          key_index[t] = *v;
        }
      for (SlotGroup &g : groups)
        for (uint32_t &v : g.shell_idx)
          v = (v == INVALID_DOF || (v >> CLS_SHIFT) == 1) ? INVALID_DOF : final_index(v);
      // ---- 4. sharded level: slots that touch shared DoFs first (stable), per group
      if (shared && !shared->empty())
        {
          std::vector<uint8_t> is_shared(n_dofs, 0);
          for (size_t t = 0; t < key_list.size(); ++t)
            if (shared->find(key_list[t]) != shared->end())
              is_shared[key_index[t]] = 1;
          for (size_t gi = 0; gi < groups.size(); ++gi)
            {
              SlotGroup   &g  = groups[gi];
              const size_t ns = g.n_slots();
              if (!ns)
                continue;
              std::vector<uint32_t> perm; // new position -> old slot
              std::vector<uint8_t>  halo(ns, 0);
              for (size_t sl = 0; sl < ns; ++sl)
                for (int t = 0; t < g.n_shell && !halo[sl]; ++t)
                  {
                    const uint32_t v = g.shell_idx[sl * g.n_shell + t];
                    if (v != INVALID_DOF && is_shared[v])
                      halo[sl] = 1;
                  }
              for (size_t sl = 0; sl < ns; ++sl)
                if (halo[sl])
                  perm.push_back((uint32_t)sl);
              g.n_halo_slots = (uint32_t)perm.size();
              for (size_t sl = 0; sl < ns; ++sl)
                if (!halo[sl])
                  perm.push_back((uint32_t)sl);
              auto permute = [&](auto &v, size_t width) {
                if (v.empty())
                  return;
                auto old = v;
                for (size_t n2 = 0; n2 < ns; ++n2)
                  std::copy(old.begin() + perm[n2] * width, old.begin() + (perm[n2] + 1) * width, v.begin() + n2 * width);
              };
              permute(g.interior_base, 1);
              permute(g.shell_idx, (size_t)g.n_shell);
              permute(g.mask, 1);
              permute(g.fmask, 1);
              permute(g.h, 1);
              permute(g.first_cell, 1);
              std::vector<uint32_t> new_of_old(ns);
              for (size_t n2 = 0; n2 < ns; ++n2)
                new_of_old[perm[n2]] = (uint32_t)n2;
              for (size_t ci = 0; ci < nc; ++ci)
                if (cell_group[ci] == gi)
                  cell_slot[ci] = new_of_old[cell_slot[ci]];
            }
        }
    }
  };
} // namespace mgamd
