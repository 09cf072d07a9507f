// Development check (host only): the ownership plan of the fused level transfers (transfer_tables.hpp).
// Invariant: a DoF owned by a fused brick (SHELL_OWN / interior) is touched by fused bricks only -- any other slot's contribution to
// the residual at that DoF would travel through the tail accumulator into a row of t that no un-fused patch restricts.
//   g++ -O2 -std=c++17 -I dealii_multigrid_amd/csrc tools/fuse_plan_check.cpp -o tools/bin/fuse_plan_check
//   tools/bin/fuse_plan_check annulus 8 4
#include "transfer_tables.hpp"

#include <cstdio>
#include <cstdlib>
#include <string>

using namespace mgamd;

int
main(int argc, char **argv)
{
  if (argc < 4)
    {
      std::printf("usage: fuse_plan_check geometry n_ref degree\n");
      return 2;
    }
  const std::string geo = argv[1];
  const int         L = std::atoi(argv[2]), p = std::atoi(argv[3]);
  std::vector<Tria> trias;
  trias.push_back(Tria::create(geo.c_str(), L, 0));
  while (trias.back().cells.size() > 1 && (int)trias.size() < 12)
    {
      Tria c = trias.back().coarsen_global();
      if (c.cells.size() == trias.back().cells.size())
        break;
      trias.push_back(std::move(c));
    }
  int rc = 0;
  for (size_t l = 0; l + 1 < trias.size(); ++l)
    {
      LevelTables fine(trias[l], p), coarse(trias[l + 1], p);
      int         fuse_group = -1;
      for (size_t gi = 0; gi < fine.groups.size(); ++gi)
        if (fine.groups[gi].N == 17 && !fine.groups[gi].constrained_group && fine.groups[gi].n_slots() > 0)
          fuse_group = (int)gi;
      if (fuse_group < 0)
        continue;
      TransferTables tt(fine, coarse, true, fuse_group);
      if (tt.tail_owned_by_fused.empty())
        continue;
      const SlotGroup  &fg = fine.groups[fuse_group];
      std::vector<bool> slot_fused(fg.n_slots(), false);
      size_t            n_fused = 0;
      for (const BrickTransferGroup &bg : tt.bricks)
        if (bg.fused)
          for (size_t q = bg.n_unfused; q < bg.n_bricks(); ++q, ++n_fused)
            slot_fused[bg.slot[q]] = true;
      size_t    bad = 0, shown = 0;
      const int n = p + 1;
      for (size_t ci = 0; ci < trias[l].cells.size(); ++ci)
        {
          if (fine.cell_group[ci] == fuse_group && slot_fused[fine.cell_slot[ci]])
            continue;
          for (int c = 0; c < n; ++c)
            for (int b = 0; b < n; ++b)
              for (int a = 0; a < n; ++a)
                {
                  const int      loc[3] = {a, b, c};
                  bool           constrained = false, corner = false;
                  const uint32_t idx = fine.cell_node_index(ci, loc, &constrained, &corner);
                  if (idx == INVALID_DOF || idx < fine.n_interior || idx >= fine.n_interior + fine.n_tail)
                    continue;
                  if (tt.tail_owned_by_fused[idx - fine.n_interior])
                    {
                      ++bad;
                      if (shown++ < 5)
                        {
                          const Cell &cc = trias[l].cells[ci];
                          std::printf("   cell %zu (level %d: %u %u %u, group %d B=%d mask 0x%x) node (%d %d %d) constrained %d corner %d -> DoF %u\n", ci,
                                      (int)cc.level, cc.i, cc.j, cc.k, (int)fine.cell_group[ci], fine.groups[fine.cell_group[ci]].B,
                                      (unsigned)trias[l].masks[ci], a, b, c, (int)constrained, (int)corner, idx);
                        }
                    }
                }
        }
      std::printf("%s L=%d p=%d transfer %zu -> %zu: %zu fused bricks of %zu slots, %zu node references from other slots to fused-owned DoFs\n", geo.c_str(),
                  L, p, trias.size() - 1 - l, trias.size() - 2 - l, n_fused, fg.n_slots(), bad);
      if (bad)
        rc = 1;
    }
  return rc;
}
