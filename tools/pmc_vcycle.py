"""HBM traffic per kernel of the V-cycles between the markers of tools/vcycle_trace.py, from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as the MI355X guide prescribes).
  python tools/pmc_vcycle.py <fetch counter_collection.csv> <write counter_collection.csv> <cycles> <out.json>
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced
streaming reads -> doubled here; WRITE_SIZE is exact; both in KiB.  Calibrated in the same run on vec_sadd_kernel (reads 2
words, writes 1 word per entry): the corrected numbers must give read/write = 2.0 and read = 16 B x n."""
import csv, json, math, re, sys
from collections import defaultdict

MARK = 304 * 256


def load(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        if "Grid_Size" not in r:
            r["Grid_Size"] = int(r["Grid_Size_X"])
        r["Grid_Size"] = int(r["Grid_Size"])
        r["Dispatch_Id"] = int(r["Dispatch_Id"])
        r["Counter_Value"] = float(r["Counter_Value"])
    rows.sort(key=lambda r: r["Dispatch_Id"])
    return rows


def window(rows):
    mark = [i for i, r in enumerate(rows) if "vec_set_kernel" in r["Kernel_Name"] and r["Grid_Size"] == MARK]
    assert len(mark) >= 2, "markers not found"
    return rows[mark[-2] + 1:mark[-1]]


def name(r):
    return re.sub(r"\(.*", "", r["Kernel_Name"].replace("mgamd::", "").replace("void ", ""))


fetch, write, cycles, out = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
# Persistent kernels launch the same grid on every level: the two passes issue the same dispatch sequence, so dispatch i of
# the fetch window and dispatch i of the write window are the same launch; launches of one symbol and grid whose fetched
# bytes differ by more than ~3x (levels differ by 8x in size) are kept in separate rows.
agg = defaultdict(lambda: [0.0, 0.0, 0])
wf, ww = window(fetch), window(write)
assert len(wf) == len(ww), "the two passes saw different dispatch sequences"
pairs = defaultdict(list)
for rf, rw in zip(wf, ww):
    assert name(rf) == name(rw) and rf["Grid_Size"] == rw["Grid_Size"]
    pairs[(name(rf), rf["Grid_Size"] // 256)].append((rf["Counter_Value"], rw["Counter_Value"]))
for (k, g), lst in pairs.items():  # split where the sorted fetch counts jump by > 2.5x
    lst.sort()
    c = 0
    for i, (f, w) in enumerate(lst):
        if i and f > 2.5 * max(lst[i - 1][0], 1.0):
            c += 1
        a = agg[(k, g, c)]
        a[0] += f * 1024 * 2
        a[1] += w * 1024
        a[2] += 1
cal_r = [r["Counter_Value"] * 2048 for r in fetch if "vec_sadd_kernel" in r["Kernel_Name"]]
cal_w = [r["Counter_Value"] * 1024 for r in write if "vec_sadd_kernel" in r["Kernel_Name"]]
rows = []
for (k, g, _), (rd, wr, n) in sorted(agg.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    rows.append({"kernel": k, "workgroups": g, "launches_per_cycle": n / cycles, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                 "hbm_bytes_per_launch": (rd + wr) / n, "hbm_bytes_per_cycle": (rd + wr) / cycles})
tot = sum(r["hbm_bytes_per_cycle"] for r in rows)
# calibration on the LARGEST vec_sadd launch of the run (y = s y + a x on n doubles reads 16 n bytes and writes 8 n: the
# finest-level update of the eigenvalue-estimation CG, or the 32 M-double calibration launches of vcycle_trace.py)
summary = {"note": __doc__, "cycles": cycles, "hbm_bytes_per_cycle_total": tot,
           "calibration_vec_sadd": {"read_over_write": (max(cal_r) / max(cal_w)) if cal_r and cal_w else None,
                                    "read_bytes": max(cal_r) if cal_r else None, "write_bytes": max(cal_w) if cal_w else None,
                                    "expected": "read = 2 x write = 16 B x n"},
           "kernels": rows}
json.dump(summary, open(out, "w"), indent=1)
print(f"HBM bytes per cycle: {tot/1e9:.3f} GB; calibration {summary['calibration_vec_sadd']}")
for r in rows[:14]:
    print(f'{r["kernel"][:60]:60s} wgs {r["workgroups"]:7d} x{r["launches_per_cycle"]:4.1f}  read {r["read_bytes_per_launch"]/1e6:9.1f} MB write {r["write_bytes_per_launch"]/1e6:9.1f} MB')
