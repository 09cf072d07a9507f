// Host-side tables of one two-level transfer: what deal.II's
// MGTwoLevelTransfer<dim,VectorType>::reinit(dof_fine, dof_coarse, constraint_fine, constraint_coarse)
// builds in the reference (ref:multigrid_throughput.cc:1600-1604).
//
// Per coarse cell the embedding is a tensor product of a 1D matrix (FE1D::embedding):
//   kind 0  identity      coarse cell is also a leaf of the fine mesh, same degree
//   kind 1  h-embedding   coarse cell has 8 children on the fine mesh, (2p+1) x (p+1)
//   kind 2  p-embedding   same cell, fine degree pf > pc,              (pf+1) x (pc+1)
// Coarse values are gathered through the coarse level's constraints (parent-resolved indices +
// in-cell interpolation, Dirichlet = 0).  deal.II weights every fine DoF with 1/(number of
// touching patches) and 0 on constrained fine DoFs; because the gathered coarse function is
// conforming, all patches touching a fine DoF produce the same value, so we give every fine DoF to
// exactly ONE patch (weight 1 there, 0 elsewhere): mathematically the same P and P^T, but
// prolongation needs no atomics and is deterministic.
#pragma once
#include "level_tables.hpp"

namespace mgamd
{
  struct TransferGroup
  {
    int                   kind = 0, nf = 2; // nf = fine nodes per direction of the patch
    std::vector<uint32_t> coarse_idx;       // per patch x (pc+1)^3: coarse DoFs (resolved, INVALID = Dirichlet)
    std::vector<uint16_t> coarse_mask;      // per patch: hanging configuration of the coarse cell
    std::vector<uint32_t> fine_idx;         // per patch x nf^3: owned fine DoF or INVALID_DOF
    size_t
    n_patches() const
    {
      return coarse_mask.size();
    }
  };

  // h-transfer of a whole fine brick (B^3 cells, lattice Nf = p*B+1) from the (B/2)^3 coarse cells under it
  // (coarse lattice Nc = p*B/2+1).  The parents of a hanging-node-free brick never carry hanging nodes (all 8
  // children of a parent lie in the same brick), so no coarse interpolation is needed here.
  struct BrickTransferGroup
  {
    int                   fine_group = 0, B = 2, Nf = 3, Nc = 2;
    std::vector<uint32_t> slot;        // fine slot index within its group
    std::vector<uint32_t> coarse_idx;  // per brick x Nc^3 (resolved, INVALID = Dirichlet)
    std::vector<uint32_t> own_shell;   // per brick x n_shell: fine shell DoF if this brick owns it, else INVALID
    // Transfers FUSED into the operator passes (kernels.hpp, MODE_RESIDUAL_RESTRICT / MODE_CHEB_PROLONGATE): the bricks
    // [n_unfused, n_bricks) of this group do their restriction / prolongation inside the level operator's brick kernel.
    // Ownership is then assigned so that every DoF touched by ANY slot outside the fused set belongs to a patch outside the
    // fused set (those patches claim first); per shell entry of a fused brick:
    //   SHELL_OWN    this brick owns the DoF: it adds b_i to the restricted residual and stores x_i + (P x_c)_i
    //   SHELL_OTHER  an un-fused patch owns it: its residual travels through the tail accumulator, and the un-fused
    //                prolongation has already corrected x_i when the fused pass gathers it
    //   0            another fused brick owns it (a brick still restricts ITS partial sum and adds the correction to its copy)
    static constexpr uint8_t SHELL_OWN = 1, SHELL_OTHER = 2;
    size_t                   n_unfused = 0;
    bool                     fused     = false;
    std::vector<uint8_t>     shell_flags; // per brick x n_shell (fused groups only; zero rows for the un-fused bricks)
    size_t
    n_bricks() const
    {
      return slot.size();
    }
  };

  class TransferTables
  {
  public:
    int                             pc = 1, pf = 1;
    TransferGroup                   groups[3];
    std::vector<BrickTransferGroup> bricks; // only with use_bricks
    // fused transfers (fuse_group >= 0): tail DoFs (index - n_interior) owned by a fused brick
    std::vector<uint8_t> tail_owned_by_fused;

    // fuse_group: fine slot group whose bricks take part in the fused transfers (-1: none); on a sharded fine level its halo
    // slots (the first n_halo_slots, which touch DoFs shared with other ranks) stay un-fused
    TransferTables(const LevelTables &fine, const LevelTables &coarse, bool use_bricks = false, int fuse_group = -1)
      : pc(coarse.p)
      , pf(fine.p)
    {
      const Tria &tf = *fine.tria, &tc = *coarse.tria;
      const int   nc1 = pc + 1;
      for (int k = 0; k < 3; ++k)
        {
          groups[k].kind = k;
          groups[k].nf   = k == 0 ? pc + 1 : (k == 1 ? 2 * pc + 1 : pf + 1);
        }
      if (&tf != &tc && pf != pc)
        throw std::runtime_error("transfer: simultaneous h- and p-coarsening is not supported");
      std::vector<bool> claimed(fine.n_dofs, false);
      std::vector<bool> covered(tc.cells.size(), false); // coarse cells handled by a brick patch
      // claims of one brick: interior DoFs are owned by construction, free shell DoFs are claimed if nobody owns them yet
      auto claim_brick = [&](BrickTransferGroup &bg, const SlotGroup &fg, size_t s, const std::vector<bool> *before, uint8_t *flags) {
        for (int t = 0; t < fg.n_interior; ++t)
          claimed[fg.interior_base[s] + t] = true;
        for (int t = 0; t < fg.n_shell; ++t)
          {
            uint32_t idx = fg.shell_idx[s * fg.n_shell + t];
            uint8_t fl = 0;
            if (idx != INVALID_DOF)
              {
                if (before && (*before)[idx])
                  fl = BrickTransferGroup::SHELL_OTHER;
                if (claimed[idx])
                  idx = INVALID_DOF;
                else
                  {
                    claimed[idx] = true;
                    fl           = BrickTransferGroup::SHELL_OWN;
                  }
              }
            bg.own_shell.push_back(idx);
            if (flags)
              flags[t] = fl;
          }
      };
      // fused transfers: DoFs that some cell reaches through a hanging-node constraint.  The constrained cell's patch does not
      // claim them (weight 0 on constrained nodes), so the "un-fused patches claim first" rule below would leave them to a
      // fused brick although a slot outside the fused set adds to their residual: a brick that touches one stays un-fused
      // (tools/fuse_plan_check.cpp checks the invariant on the host; annulus L = 8, p = 4 has 24 such bricks)
      std::vector<bool> reached_through_constraint;
      if (use_bricks && fuse_group >= 0 && &tf != &tc && pf == pc)
        {
          reached_through_constraint.assign(fine.n_dofs, false);
          const int n = pf + 1;
          for (size_t ci = 0; ci < tf.cells.size(); ++ci)
            if ((tf.masks[ci] >> MASK_FACE_SHIFT) && fine.cell_is_local(ci))
              for (int c = 0; c < n; ++c)
                for (int b = 0; b < n; ++b)
                  for (int a = 0; a < n; ++a)
                    {
                      const int      l[3] = {a, b, c};
                      bool           constrained = false;
                      const uint32_t idx = fine.cell_node_index(ci, l, &constrained);
                      if (constrained && idx != INVALID_DOF)
                        reached_through_constraint[idx] = true;
                    }
        }
      if (use_bricks && &tf != &tc && pf == pc)
        for (size_t gi = 0; gi < fine.groups.size(); ++gi)
          {
            const SlotGroup &fg = fine.groups[gi];
            if (fg.B < 2 || fg.n_slots() == 0)
              continue;
            BrickTransferGroup bg;
            bg.fine_group = (int)gi;
            bg.B          = fg.B;
            bg.Nf         = fg.N;
            bg.Nc         = pc * fg.B / 2 + 1;
            bg.fused      = (int)gi == fuse_group;
            const int Bc  = fg.B / 2;
            // pass 1: which slots have a brick patch.  All parents must be hanging-node-free leaves of the coarse mesh;
            // otherwise (cells not coarsened on this level, or re-refined by the 2:1 balance) the per-cell patches below
            // take over.  The un-fused bricks of a fused group (halo slots of a sharded level) come first.
            std::vector<uint32_t>             slots[2]; // [0] un-fused, [1] fused
            std::vector<std::vector<int32_t>> pars[2];
            for (size_t s = 0; s < fg.n_slots(); ++s)
              {
                if (fg.fmask[s])
                  continue; // constrained brick: per-cell patches handle its hanging nodes
                const Cell &fc = tf.cells[fg.first_cell[s]];
                const Cell  anchor{fc.i & ~(uint32_t)(fg.B - 1), fc.j & ~(uint32_t)(fg.B - 1), fc.k & ~(uint32_t)(fg.B - 1), fc.level};
                std::vector<int32_t> parents((size_t)Bc * Bc * Bc, -1);
                bool                 ok = true;
                for (int cz = 0; ok && cz < Bc; ++cz)
                  for (int cy = 0; ok && cy < Bc; ++cy)
                    for (int cx = 0; ok && cx < Bc; ++cx)
                      {
                        const int32_t *par = tc.index.find(cell_key(anchor.level - 1, (anchor.i >> 1) + cx, (anchor.j >> 1) + cy, (anchor.k >> 1) + cz));
                        if (!par || (tc.masks[*par] >> MASK_FACE_SHIFT) || !coarse.cell_is_local((size_t)*par))
                          ok = false;
                        else
                          parents[(cz * Bc + cy) * Bc + cx] = *par;
                      }
                if (!ok)
                  continue;
                int f = (bg.fused && s >= fg.n_halo_slots) ? 1 : 0;
                for (int t = 0; f && t < fg.n_shell; ++t)
                  {
                    const uint32_t idx = fg.shell_idx[s * fg.n_shell + t];
                    if (idx != INVALID_DOF && reached_through_constraint[idx])
                      f = 0;
                  }
                slots[f].push_back((uint32_t)s);
                for (int32_t par : parents)
                  covered[par] = true;
                pars[f].push_back(std::move(parents));
              }
            bg.n_unfused = slots[0].size();
            for (int f = 0; f < 2; ++f)
              for (size_t q = 0; q < slots[f].size(); ++q)
                {
                  bg.slot.push_back(slots[f][q]);
                  for (int Z = 0; Z < bg.Nc; ++Z)
                    for (int Y = 0; Y < bg.Nc; ++Y)
                      for (int X = 0; X < bg.Nc; ++X)
                        {
                          const int c[3] = {std::min(X / pc, Bc - 1), std::min(Y / pc, Bc - 1), std::min(Z / pc, Bc - 1)};
                          const int l[3] = {X - c[0] * pc, Y - c[1] * pc, Z - c[2] * pc};
                          bg.coarse_idx.push_back(coarse.cell_node_index((size_t)pars[f][q][(c[2] * Bc + c[1]) * Bc + c[0]], l));
                        }
                }
            // pass 2: the un-fused bricks claim now; the fused ones after every other patch (below)
            for (size_t q = 0; q < bg.n_unfused; ++q)
              claim_brick(bg, fg, bg.slot[q], nullptr, nullptr);
            if (bg.n_bricks())
              bricks.push_back(std::move(bg));
          }
      for (size_t ci = 0; ci < tc.cells.size(); ++ci)
        {
          if (covered[ci] || !coarse.cell_is_local(ci))
            continue;
          const Cell    &cc = tc.cells[ci];
          const int32_t *same = tf.index.find(cell_key(cc));
          int            kind;
          if (same)
            {
              kind = pf == pc ? 0 : 2;
              if (!fine.cell_is_local((size_t)*same))
                continue; // distributed fine level over a replicated coarse level: another rank's patch
            }
          else
            {
              if (pf != pc)
                throw std::runtime_error("transfer: refined cell in a p-transfer");
              kind = 1;
              // local smoothing (MGTransferMatrixFree between refinement levels): the ACTIVE cells of the coarser level
              // have no counterpart on the finer level and take no part in the transfer
              if (coarse.ls_level && !tf.index.find(cell_key(cc.level + 1, 2 * cc.i, 2 * cc.j, 2 * cc.k)))
                continue;
              // at the partition's root level the 8 children may belong to different ranks: keep the patch if any is ours
              bool any_local = false;
              for (int t = 0; t < 8; ++t)
                {
                  const int32_t *f = tf.index.find(cell_key(cc.level + 1, 2 * cc.i + (t & 1), 2 * cc.j + ((t >> 1) & 1), 2 * cc.k + (t >> 2)));
                  if (f && fine.cell_is_local((size_t)*f))
                    any_local = true;
                }
              if (!any_local)
                continue;
            }
          TransferGroup &g = groups[kind];
          g.coarse_mask.push_back(tc.masks[ci]);
          for (int c = 0; c < nc1; ++c)
            for (int b = 0; b < nc1; ++b)
              for (int a = 0; a < nc1; ++a)
                {
                  const int l[3] = {a, b, c};
                  g.coarse_idx.push_back(coarse.cell_node_index(ci, l));
                }
          const int nf = g.nf;
          for (int Z = 0; Z < nf; ++Z)
            for (int Y = 0; Y < nf; ++Y)
              for (int X = 0; X < nf; ++X)
                {
                  uint32_t idx = INVALID_DOF;
                  if (kind == 1)
                    {
                      // a node on the plane between two children belongs to both: take the first LOCAL child for which
                      // it is a regular (non-hanging) node
                      const int lo[3] = {X > pc ? 1 : 0, Y > pc ? 1 : 0, Z > pc ? 1 : 0};
                      const int hi[3] = {X >= pc ? 1 : 0, Y >= pc ? 1 : 0, Z >= pc ? 1 : 0};
                      for (int cz = lo[2]; cz <= hi[2] && idx == INVALID_DOF; ++cz)
                        for (int cy = lo[1]; cy <= hi[1] && idx == INVALID_DOF; ++cy)
                          for (int cx = lo[0]; cx <= hi[0] && idx == INVALID_DOF; ++cx)
                            {
                              const int32_t *f = tf.index.find(cell_key(cc.level + 1, 2 * cc.i + cx, 2 * cc.j + cy, 2 * cc.k + cz));
                              if (!f)
                                throw std::runtime_error("transfer: fine mesh is not a one-level refinement of the coarse mesh");
                              if (!fine.cell_is_local((size_t)*f))
                                continue;
                              const int l[3] = {X - cx * pc, Y - cy * pc, Z - cz * pc};
                              bool      constrained, corner;
                              uint32_t  v = fine.cell_node_index((size_t)*f, l, &constrained, &corner);
                              if (constrained && !corner)
                                v = INVALID_DOF; // own DoF is a hanging node: weight 0
                              idx = v;
                            }
                    }
                  else
                    {
                      const int l[3] = {X, Y, Z};
                      bool      constrained, corner;
                      idx = fine.cell_node_index((size_t)*same, l, &constrained, &corner);
                      if (constrained && !corner)
                        idx = INVALID_DOF;
                    }
                  if (idx != INVALID_DOF)
                    {
                      if (claimed[idx])
                        idx = INVALID_DOF;
                      else
                        claimed[idx] = true;
                    }
                  g.fine_idx.push_back(idx);
                }
        }
      // fused bricks claim last: what is left for them is touched by fused bricks only
      for (BrickTransferGroup &bg : bricks)
        if (bg.fused)
          {
            const SlotGroup        &fg     = fine.groups[bg.fine_group];
            const std::vector<bool> before = claimed;
            bg.shell_flags.assign(bg.n_bricks() * (size_t)fg.n_shell, 0);
            for (size_t q = bg.n_unfused; q < bg.n_bricks(); ++q)
              claim_brick(bg, fg, bg.slot[q], &before, &bg.shell_flags[q * (size_t)fg.n_shell]);
            tail_owned_by_fused.assign(fine.n_tail, 0);
            for (size_t q = bg.n_unfused; q < bg.n_bricks(); ++q)
              for (int t = 0; t < fg.n_shell; ++t)
                if (bg.shell_flags[q * (size_t)fg.n_shell + t] == BrickTransferGroup::SHELL_OWN)
                  tail_owned_by_fused[bg.own_shell[q * (size_t)fg.n_shell + t] - fine.n_interior] = 1;
          }
    }
  };
} // namespace mgamd

namespace mgamd
{
  // Local smoothing: copy_to_mg / copy_from_mg index pairs of one level (MGLevelGlobalTransfer with skip_interface_dofs,
  // ref:multigrid_throughput.cc:1789-1791 MGTransferMatrixFree::build): the DoFs of the ACTIVE cells of refinement level
  // `level`, except those on the level's refinement edge (they live on the coarser level) and the Dirichlet DoFs.
  // global: tables of the active mesh (outer problem); lev: tables of the level mesh.
  inline void
  ls_copy_indices(const LevelTables &global, const LevelTables &lev, int level, std::vector<uint32_t> &gidx, std::vector<uint32_t> &lidx)
  {
    gidx.clear();
    lidx.clear();
    if (global.p != lev.p || !lev.ls_level)
      throw std::invalid_argument("ls_copy_indices: need the active-mesh tables and a local-smoothing level of the same degree");
    const Tria       &tg = *global.tria, &tl = *lev.tria;
    const int         n  = global.p + 1;
    std::vector<bool> seen(lev.n_dofs, false);
    for (size_t gc = 0; gc < tg.cells.size(); ++gc)
      {
        const Cell &c = tg.cells[gc];
        if ((int)c.level != level || !global.cell_is_local(gc))
          continue;
        const int32_t *lc = tl.index.find(cell_key(c));
        if (!lc)
          throw std::runtime_error("ls_copy_indices: active cell missing on its level mesh");
        for (int z = 0; z < n; ++z)
          for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x)
              {
                const int      a[3] = {x, y, z};
                const uint32_t li   = lev.cell_node_index((size_t)*lc, a);
                if (li == INVALID_DOF || li >= lev.first_constrained() || seen[li])
                  continue; // Dirichlet, refinement edge, or already listed
                bool           constrained = false;
                const uint32_t gi          = global.cell_node_index(gc, a, &constrained);
                if (constrained || gi == INVALID_DOF || gi >= global.first_constrained())
                  throw std::runtime_error("ls_copy_indices: a level-interior DoF of an active cell is constrained on the active mesh");
                seen[li] = true;
                gidx.push_back(gi);
                lidx.push_back(li);
              }
      }
  }
} // namespace mgamd
