"""Per-kernel table of the V-cycles between the two markers of tools/vcycle_trace.py, from rocprofv3's kernel trace CSV.
  python tools/vcycle_table.py <kernel_trace.csv> <cycles> [out.csv]
Rows: kernel symbol (template arguments kept, parameter list dropped) x number of workgroups; columns: launches per cycle,
average duration, time per cycle, share.  Persistent kernels launch the same grid on every level: launches of one symbol and
grid whose durations differ by more than ~3x (levels differ by 8x in size) are kept in separate rows.  The last line is the sum = GPU-busy time per cycle (gaps between kernels excluded)
and the wall span per cycle (first start to last end)."""
import csv, math, re, sys
from collections import defaultdict

path, cycles = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
if rows and "Grid_Size" not in rows[0]:  # kernel-trace CSV: per-dimension columns
    for r in rows:
        r["Grid_Size"] = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        r["Workgroup_Size"] = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = [i for i, r in enumerate(rows) if "vec_set_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 304 * 256]
assert len(mark) >= 2, f"markers not found ({len(mark)})"
sel = rows[mark[-2] + 1:mark[-1]]
agg, agg0 = defaultdict(list), defaultdict(list)
for r in sel:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("mgamd::", "").replace("void ", ""))
    wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    agg0[(name, wgs)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, wgs), durs in agg0.items():  # split the launches of one symbol and grid where the sorted durations jump by > 2.5x
    durs.sort()
    c = 0
    for i, d in enumerate(durs):
        if i and d > 2.5 * durs[i - 1]:
            c += 1
        agg[(name, wgs, c)].append(d)
tot = sum(sum(v) for v in agg.values())
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
out = [("kernel", "workgroups", "launches_per_cycle", "avg_us", "us_per_cycle", "share_pct")]
for (n, g, _), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    out.append((n, g, len(v) / cycles, sum(v) / len(v) / 1e3, sum(v) / cycles / 1e3, 100.0 * sum(v) / tot))
out.append(("SUM (GPU busy)", "", sum(len(v) for v in agg.values()) / cycles, "", tot / cycles / 1e3, 100.0))
out.append(("WALL SPAN", "", "", "", span / cycles / 1e3, ""))
w = csv.writer(open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout)
for r in out:
    w.writerow([f"{x:.2f}" if isinstance(x, float) else x for x in r])
if len(sys.argv) > 3:
    for r in out[:16] + out[-2:]:
        print(" ".join(f"{x:.2f}" if isinstance(x, float) else str(x) for x in r)[:200])
