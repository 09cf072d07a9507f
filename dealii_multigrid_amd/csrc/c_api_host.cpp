// C ABI, host-side setup entry points (include/mgamd.h, section "Host-side setup").
#include "api_common.hpp"

using namespace mgamd;

namespace mgamd
{
  thread_local std::string g_last_error;
}

extern "C" {

const char *
mgamd_last_error(void)
{
  return g_last_error.c_str();
}

const char *
mgamd_version(void)
{
  return "mgamd 0.1 (gfx950)";
}

int
mgamd_tria_create(const char *geometry, unsigned n_ref_global, unsigned n_ref_local, mgamd_tria **out)
{
  MGAMD_TRY
  if (!geometry || !out)
    throw std::invalid_argument("null argument");
  auto *t = new mgamd_tria;
  t->tria = std::make_shared<Tria>(Tria::create(geometry, n_ref_global, n_ref_local));
  *out    = t;
  MGAMD_CATCH
}

int
mgamd_tria_create_from_leaves(uint64_t n_leaves, const uint8_t *level, const uint32_t *i, const uint32_t *j, const uint32_t *k, mgamd_tria **out)
{
  MGAMD_TRY
  if (!level || !i || !j || !k || !out)
    throw std::invalid_argument("null argument");
  std::vector<Cell> leaves(n_leaves);
  for (uint64_t t = 0; t < n_leaves; ++t)
    leaves[t] = Cell{i[t], j[t], k[t], level[t]};
  auto *t = new mgamd_tria;
  try
    {
      t->tria = std::make_shared<Tria>(Tria::from_leaves_checked(std::move(leaves)));
    }
  catch (...)
    {
      delete t;
      throw;
    }
  *out = t;
  MGAMD_CATCH
}

int
mgamd_dofs_matrix(const mgamd_dofs *d, uint64_t *nnz, uint32_t *row_ptr, uint32_t *col, double *val)
{
  MGAMD_TRY
  if (!d || !nnz)
    throw std::invalid_argument("null argument");
  if (d->halo)
    throw std::invalid_argument("mgamd_dofs_matrix: the level is distributed");
  const CSR A = assemble_level_matrix(*d->tables);
  *nnz        = A.nnz();
  if (row_ptr)
    std::copy(A.ptr.begin(), A.ptr.end(), row_ptr);
  if (col)
    std::copy(A.col.begin(), A.col.end(), col);
  if (val)
    std::copy(A.val.begin(), A.val.end(), val);
  MGAMD_CATCH
}

int
mgamd_dofs_amg_setup_info(const mgamd_dofs *d, uint32_t *n_levels, uint32_t *rows, uint64_t *nnz, uint32_t max_levels)
{
  MGAMD_TRY
  if (!d || !n_levels)
    throw std::invalid_argument("null argument");
  const AmgHierarchyHost H = build_smoothed_aggregation(assemble_level_matrix(*d->tables));
  *n_levels                = (uint32_t)H.levels.size();
  for (uint32_t l = 0; l < H.levels.size() && l < max_levels; ++l)
    {
      if (rows)
        rows[l] = H.levels[l].A.n_rows;
      if (nnz)
        nnz[l] = H.levels[l].A.nnz();
    }
  MGAMD_CATCH
}

int
mgamd_tria_coarsen(const mgamd_tria *fine, mgamd_tria **out)
{
  MGAMD_TRY
  if (!fine || !out)
    throw std::invalid_argument("null argument");
  auto *t = new mgamd_tria;
  t->tria = std::make_shared<Tria>(fine->tria->coarsen_global());
  *out    = t;
  MGAMD_CATCH
}

int
mgamd_tria_destroy(mgamd_tria *t)
{
  delete t;
  return MGAMD_OK;
}

int
mgamd_tria_info(const mgamd_tria *t, uint64_t *n_cells, uint32_t *n_levels, uint64_t *n_cells_hn)
{
  MGAMD_TRY
  if (!t)
    throw std::invalid_argument("null argument");
  if (n_cells)
    *n_cells = t->tria->n_cells();
  if (n_levels)
    *n_levels = (uint32_t)t->tria->n_levels();
  if (n_cells_hn)
    *n_cells_hn = t->tria->n_cells_with_hanging_nodes();
  MGAMD_CATCH
}

int
mgamd_tria_get_cells(const mgamd_tria *t, uint8_t *level, uint32_t *i, uint32_t *j, uint32_t *k, uint16_t *mask)
{
  MGAMD_TRY
  if (!t)
    throw std::invalid_argument("null argument");
  const auto &c = t->tria->cells;
  for (size_t n = 0; n < c.size(); ++n)
    {
      if (level)
        level[n] = c[n].level;
      if (i)
        i[n] = c[n].i;
      if (j)
        j[n] = c[n].j;
      if (k)
        k[n] = c[n].k;
      if (mask)
        mask[n] = t->tria->masks[n];
    }
  MGAMD_CATCH
}

static size_t
n_nonempty_groups(const LevelTables &L)
{
  size_t n = 0;
  for (const auto &g : L.groups)
    n += g.n_slots() > 0;
  return n;
}

int
mgamd_dofs_create(const mgamd_tria *t, int degree, int max_brick, mgamd_dofs **out)
{
  MGAMD_TRY
  if (!t || !out)
    throw std::invalid_argument("null argument");
  if (degree < 1 || degree > MAX_DEGREE)
    throw std::invalid_argument("degree must be in [1," + std::to_string(MAX_DEGREE) + "]");
  auto *d   = new mgamd_dofs;
  d->tria   = t->tria;
  d->tables = std::make_shared<LevelTables>(*d->tria, degree, std::max(max_brick, 0));
  if (max_brick < 0 && is_small_level(d->tria->n_cells(), degree) && n_nonempty_groups(*d->tables) > 1)
    d->tables = std::make_shared<LevelTables>(*d->tria, degree, 1);
  *out      = d;
  MGAMD_CATCH
}

int
mgamd_dofs_destroy(mgamd_dofs *d)
{
  delete d;
  return MGAMD_OK;
}

int
mgamd_dofs_info(const mgamd_dofs *d, mgamd_dofs_info_t *info)
{
  MGAMD_TRY
  if (!d || !info)
    throw std::invalid_argument("null argument");
  const LevelTables &L = *d->tables;
  std::memset(info, 0, sizeof(*info));
  info->degree      = L.p;
  info->n_cells     = L.tria->n_cells();
  info->n_dofs      = L.n_dofs;
  info->n_interior  = L.n_interior;
  info->n_tail      = L.n_tail;
  info->n_dirichlet = L.n_dirichlet;
  info->n_hanging   = L.n_hanging;
  info->n_groups    = (uint32_t)L.groups.size();
  info->n_tail_owned      = L.n_tail_owned;
  info->n_dirichlet_owned = L.n_dirichlet_owned;
  info->n_hanging_owned   = L.n_hanging_owned;
  info->n_peers           = d->halo ? (uint32_t)d->halo->peers.size() : 0;
  info->n_halo_send       = d->halo ? (uint32_t)d->halo->pack_idx.size() : 0;
  info->n_edge            = L.n_edge;
  for (size_t g = 0; g < L.groups.size() && g < 8; ++g)
    {
      info->group_B[g]     = L.groups[g].B;
      info->group_slots[g] = L.groups[g].n_slots();
      info->group_halo_slots[g] = L.groups[g].n_halo_slots;
    }
  MGAMD_CATCH
}

int
mgamd_tria_level_mesh(const mgamd_tria *fine, unsigned level, mgamd_tria **out)
{
  MGAMD_TRY
  if (!fine || !out)
    throw std::invalid_argument("null argument");
  if ((int)level >= fine->tria->n_levels())
    throw std::invalid_argument("level_mesh: the mesh has no cells on that refinement level");
  auto *t = new mgamd_tria;
  t->tria = std::make_shared<Tria>(fine->tria->level_mesh((int)level));
  *out    = t;
  MGAMD_CATCH
}

int
mgamd_dofs_create_level(const mgamd_tria *level_mesh, int degree, int max_brick, mgamd_dofs **out)
{
  MGAMD_TRY
  if (!level_mesh || !out)
    throw std::invalid_argument("null argument");
  if (degree < 1 || degree > MAX_DEGREE)
    throw std::invalid_argument("degree must be in [1," + std::to_string(MAX_DEGREE) + "]");
  auto *d = new mgamd_dofs;
  d->tria = level_mesh->tria;
  try
    {
      d->tables = std::make_shared<LevelTables>(*d->tria, degree, std::max(max_brick, 0), nullptr, false, nullptr, 0, true);
      if (max_brick < 0 && is_small_level(d->tria->n_cells(), degree) && n_nonempty_groups(*d->tables) > 1)
        d->tables = std::make_shared<LevelTables>(*d->tria, degree, 1, nullptr, false, nullptr, 0, true);
    }
  catch (...)
    {
      delete d;
      throw;
    }
  *out = d;
  MGAMD_CATCH
}

int
mgamd_ls_copy_indices(const mgamd_dofs *active_mesh_dofs, const mgamd_dofs *level_dofs, unsigned level, uint64_t *count, uint32_t *global_idx,
                      uint32_t *level_idx)
{
  MGAMD_TRY
  if (!active_mesh_dofs || !level_dofs || !count)
    throw std::invalid_argument("null argument");
  std::vector<uint32_t> g, l;
  ls_copy_indices(*active_mesh_dofs->tables, *level_dofs->tables, (int)level, g, l);
  *count = g.size();
  if (global_idx)
    std::copy(g.begin(), g.end(), global_idx);
  if (level_idx)
    std::copy(l.begin(), l.end(), level_idx);
  MGAMD_CATCH
}

int
mgamd_dofs_get_cell_slots(const mgamd_dofs *d, uint8_t *group, uint32_t *slot)
{
  MGAMD_TRY
  if (!d || !group || !slot)
    throw std::invalid_argument("null argument");
  const LevelTables &L = *d->tables;
  std::copy(L.cell_group.begin(), L.cell_group.end(), group);
  std::copy(L.cell_slot.begin(), L.cell_slot.end(), slot);
  MGAMD_CATCH
}

int
mgamd_dofs_get_keys(const mgamd_dofs *d, int32_t *keys)
{
  MGAMD_TRY
  if (!d || !keys)
    throw std::invalid_argument("null argument");
  std::vector<DofKey> k;
  d->tables->export_dof_keys(k);
  static_assert(sizeof(DofKey) == 5 * sizeof(int32_t), "DofKey layout");
  std::memcpy(keys, k.data(), k.size() * sizeof(DofKey));
  MGAMD_CATCH
}

int
mgamd_dofs_get_cell_dofs(const mgamd_dofs *d, uint32_t *out)
{
  MGAMD_TRY
  if (!d || !out)
    throw std::invalid_argument("null argument");
  std::vector<uint32_t> v;
  d->tables->export_cell_dofs(v);
  std::memcpy(out, v.data(), v.size() * sizeof(uint32_t));
  MGAMD_CATCH
}

int
mgamd_dofs_rhs(const mgamd_dofs *d, int kind, double *out)
{
  MGAMD_TRY
  if (!d || !out)
    throw std::invalid_argument("null argument");
  if (kind != 0 && kind != 1)
    throw std::invalid_argument("SimulationType kind must be 0 (Constant) or 1 (Gaussian)");
  std::vector<double> b;
  d->tables->compute_rhs_function(kind, b);
  std::memcpy(out, b.data(), b.size() * sizeof(double));
  MGAMD_CATCH
}

int
mgamd_dofs_distribute(const mgamd_dofs *d, int kind, double *x)
{
  MGAMD_TRY
  if (!d || !x)
    throw std::invalid_argument("null argument");
  if (kind != 0 && kind != 1)
    throw std::invalid_argument("SimulationType kind must be 0 (Constant) or 1 (Gaussian)");
  std::vector<double> v(x, x + d->tables->n_dofs);
  d->tables->distribute(kind, v);
  std::memcpy(x, v.data(), v.size() * sizeof(double));
  MGAMD_CATCH
}

int
mgamd_dofs_rhs_constant(const mgamd_dofs *d, double *out)
{
  MGAMD_TRY
  if (!d || !out)
    throw std::invalid_argument("null argument");
  std::vector<double> b;
  d->tables->compute_rhs_constant(b);
  std::memcpy(out, b.data(), b.size() * sizeof(double));
  MGAMD_CATCH
}

int
mgamd_partition_create(const mgamd_tria *const *trias, unsigned n_levels, unsigned n_ranks, double hanging_weight, mgamd_partition **out)
{
  return mgamd_partition_create_ex(trias, n_levels, n_ranks, hanging_weight, 0, out);
}

int
mgamd_partition_create_ex(const mgamd_tria *const *trias, unsigned n_levels, unsigned n_ranks, double hanging_weight,
                          uint64_t min_root_cells, mgamd_partition **out)
{
  return mgamd_partition_create_tiered(trias, n_levels, n_ranks, hanging_weight, min_root_cells, 1, 0, out);
}

int
mgamd_partition_create_tiered(const mgamd_tria *const *trias, unsigned n_levels, unsigned n_ranks, double hanging_weight,
                              uint64_t min_root_cells, unsigned group, uint64_t min_sub_root_cells, mgamd_partition **out)
{
  MGAMD_TRY
  if (!trias || !out || n_levels == 0 || n_ranks == 0 || group == 0)
    throw std::invalid_argument("bad argument");
  auto                     *p = new mgamd_partition;
  std::vector<const Tria *> raw;
  for (unsigned l = 0; l < n_levels; ++l)
    {
      if (!trias[l])
        throw std::invalid_argument("null triangulation");
      p->trias.push_back(trias[l]->tria);
      raw.push_back(trias[l]->tria.get());
    }
  try
    {
      p->part = make_partition(raw, (int)n_ranks, hanging_weight, 32, (size_t)min_root_cells, (int)group, (size_t)min_sub_root_cells);
    }
  catch (...)
    {
      delete p;
      throw;
    }
  *out = p;
  MGAMD_CATCH
}

int
mgamd_partition_destroy(mgamd_partition *p)
{
  delete p;
  return MGAMD_OK;
}

int
mgamd_partition_info(const mgamd_partition *p, unsigned *root_level, unsigned *n_ranks)
{
  MGAMD_TRY
  if (!p)
    throw std::invalid_argument("null argument");
  if (root_level)
    *root_level = (unsigned)p->part.root_level;
  if (n_ranks)
    *n_ranks = (unsigned)p->part.n_ranks;
  MGAMD_CATCH
}

int
mgamd_partition_tiers(const mgamd_partition *p, unsigned *sub_root_level, unsigned *group)
{
  MGAMD_TRY
  if (!p)
    throw std::invalid_argument("null argument");
  if (sub_root_level)
    *sub_root_level = (unsigned)p->part.sub_root_level;
  if (group)
    *group = (unsigned)p->part.group;
  MGAMD_CATCH
}

int
mgamd_partition_statistics(const mgamd_partition *p, double stats[5])
{
  MGAMD_TRY
  if (!p || !stats)
    throw std::invalid_argument("null argument");
  std::vector<const Tria *> trias;
  for (const auto &t : p->trias)
    trias.push_back(t.get());
  const PartitionStatistics st = partition_statistics(trias, p->part);
  stats[0] = st.workload_eff;
  stats[1] = st.workload_path_max;
  stats[2] = st.vertical_eff;
  stats[3] = st.horizontal_eff;
  stats[4] = st.mem_total;
  MGAMD_CATCH
}

int
mgamd_partition_get_owner(const mgamd_partition *p, unsigned level, uint16_t *owner)
{
  MGAMD_TRY
  if (!p || !owner || level >= p->trias.size() || (int)level < p->part.sub_root_level)
    throw std::invalid_argument("bad argument");
  const auto &o = p->part.level_owner((int)level);
  std::memcpy(owner, o.data(), o.size() * sizeof(uint16_t));
  MGAMD_CATCH
}

int
mgamd_dofs_create_local(const mgamd_partition *p, unsigned level, unsigned rank, int degree, int max_brick, mgamd_dofs **out)
{
  MGAMD_TRY
  if (!p || !out || level >= p->trias.size() || (int)rank >= p->part.n_ranks)
    throw std::invalid_argument("bad argument");
  if (degree < 1 || degree > MAX_DEGREE)
    throw std::invalid_argument("degree must be in [1," + std::to_string(MAX_DEGREE) + "]");
  auto *d    = new mgamd_dofs;
  d->tria    = p->trias[level];
  // the level's pieces and the one this rank works on (subset levels: parts, shared by the ranks of a group)
  const int np = p->part.n_parts((int)level), my = p->part.part_of((int)level, (int)rank);
  d->n_ranks   = np;
  d->rank      = my;
  try
    {
      if (p->part.replicated((int)level) || p->part.n_ranks == 1)
        {
          d->tables = std::make_shared<LevelTables>(*d->tria, degree, std::max(max_brick, 0));
          if (max_brick < 0 && is_small_level(d->tria->n_cells(), degree) && n_nonempty_groups(*d->tables) > 1)
            d->tables = std::make_shared<LevelTables>(*d->tria, degree, 1);
        }
      else
        {
          const auto &owner = p->part.level_owner((int)level);
          d->owned          = std::make_shared<std::vector<uint8_t>>(owner.size());
          for (size_t c = 0; c < owner.size(); ++c)
            (*d->owned)[c] = owner[c] == my;
          d->shared = std::make_shared<std::map<uint64_t, SharedInfo>>(shared_keys(*d->tria, owner, np, my, degree));
          d->tables = std::make_shared<LevelTables>(*d->tria, degree, std::max(max_brick, 0), d->owned.get(), false, d->shared.get(), my);
          if (max_brick < 0 && is_small_level(d->tria->n_cells() / np, degree) && n_nonempty_groups(*d->tables) > 1)
            d->tables = std::make_shared<LevelTables>(*d->tria, degree, 1, d->owned.get(), false, d->shared.get(), my);
          d->halo   = std::make_shared<HaloPlan>(make_halo_plan(*d->tables, *d->shared, np, my));
        }
    }
  catch (...)
    {
      delete d;
      throw;
    }
  *out = d;
  MGAMD_CATCH
}

int
mgamd_dofs_halo_sizes(const mgamd_dofs *d, uint32_t sizes[4])
{
  MGAMD_TRY
  if (!d || !sizes)
    throw std::invalid_argument("null argument");
  sizes[0] = sizes[1] = sizes[2] = sizes[3] = 0;
  if (d->halo)
    {
      sizes[0] = (uint32_t)d->halo->peers.size();
      sizes[1] = (uint32_t)d->halo->pack_idx.size();
      sizes[2] = (uint32_t)d->halo->sh_tail.size();
      sizes[3] = (uint32_t)d->halo->sh_src.size();
    }
  MGAMD_CATCH
}

int
mgamd_dofs_halo_get(const mgamd_dofs *d, int32_t *peers, uint32_t *peer_offset, uint32_t *pack_idx, uint32_t *sh_tail, uint32_t *sh_ptr,
                    int32_t *sh_src, int32_t *sh_owner_src)
{
  MGAMD_TRY
  if (!d || !d->halo)
    throw std::invalid_argument("not a distributed level");
  const HaloPlan &H   = *d->halo;
  auto            cpy = [](auto *dst, const auto &v) {
    if (dst && !v.empty())
      std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  cpy(peers, H.peers);
  cpy(peer_offset, H.peer_offset);
  cpy(pack_idx, H.pack_idx);
  cpy(sh_tail, H.sh_tail);
  cpy(sh_ptr, H.sh_ptr);
  cpy(sh_src, H.sh_src);
  cpy(sh_owner_src, H.sh_owner_src);
  MGAMD_CATCH
}

int
mgamd_transfer_tables_info(const mgamd_dofs *fine, const mgamd_dofs *coarse, uint64_t n_patches[3], uint32_t nf[3])
{
  MGAMD_TRY
  if (!fine || !coarse)
    throw std::invalid_argument("null argument");
  TransferTables T(*fine->tables, *coarse->tables);
  for (int k = 0; k < 3; ++k)
    {
      if (n_patches)
        n_patches[k] = T.groups[k].n_patches();
      if (nf)
        nf[k] = T.groups[k].nf;
    }
  MGAMD_CATCH
}

int
mgamd_transfer_tables_get(const mgamd_dofs *fine, const mgamd_dofs *coarse, int kind, uint32_t *coarse_idx, uint16_t *coarse_mask,
                          uint32_t *fine_idx)
{
  MGAMD_TRY
  if (!fine || !coarse || kind < 0 || kind > 2)
    throw std::invalid_argument("bad argument");
  TransferTables       T(*fine->tables, *coarse->tables);
  const TransferGroup &g = T.groups[kind];
  if (coarse_idx)
    std::memcpy(coarse_idx, g.coarse_idx.data(), g.coarse_idx.size() * sizeof(uint32_t));
  if (coarse_mask)
    std::memcpy(coarse_mask, g.coarse_mask.data(), g.coarse_mask.size() * sizeof(uint16_t));
  if (fine_idx)
    std::memcpy(fine_idx, g.fine_idx.data(), g.fine_idx.size() * sizeof(uint32_t));
  MGAMD_CATCH
}

} // extern "C"
