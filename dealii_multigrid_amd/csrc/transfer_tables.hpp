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

  class TransferTables
  {
  public:
    int           pc = 1, pf = 1;
    TransferGroup groups[3];

    TransferTables(const LevelTables &fine, const LevelTables &coarse)
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
      for (size_t ci = 0; ci < tc.cells.size(); ++ci)
        {
          const Cell    &cc = tc.cells[ci];
          const int32_t *same = tf.index.find(cell_key(cc));
          int            kind;
          if (same)
            kind = pf == pc ? 0 : 2;
          else
            {
              if (pf != pc)
                throw std::runtime_error("transfer: refined cell in a p-transfer");
              kind = 1;
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
                  size_t fci;
                  int    l[3] = {X, Y, Z};
                  if (kind == 1)
                    {
                      const int      ch[3] = {X > pc ? 1 : 0, Y > pc ? 1 : 0, Z > pc ? 1 : 0};
                      const int32_t *f     = tf.index.find(cell_key(cc.level + 1, 2 * cc.i + ch[0], 2 * cc.j + ch[1], 2 * cc.k + ch[2]));
                      if (!f)
                        throw std::runtime_error("transfer: fine mesh is not a one-level refinement of the coarse mesh");
                      fci = (size_t)*f;
                      for (int d = 0; d < 3; ++d)
                        l[d] -= ch[d] * pc;
                    }
                  else
                    fci = (size_t)*same;
                  bool     constrained, corner;
                  uint32_t idx = fine.cell_node_index(fci, l, &constrained, &corner);
                  if (constrained && !corner)
                    idx = INVALID_DOF; // own DoF is a hanging node: weight 0
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
    }
  };
} // namespace mgamd
