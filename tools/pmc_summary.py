"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic per kernel.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced
streaming reads -> doubled here; WRITE_SIZE is exact.  Calibrated in the same run on vec_sadd_kernel (reads 2 words,
writes 1 word per entry): the corrected numbers must give read/write = 2.0."""
import collections, csv, json, sys

def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return d

fetch, write, out = agg(sys.argv[1]), agg(sys.argv[2]), sys.argv[3]
rows = []
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    f, w = sorted(fetch[k]), sorted(write.get(k, [0.0]))
    fm, wm = f[len(f) // 2] * 1024 * 2, w[len(w) // 2] * 1024
    rows.append({"kernel": k[0], "grid_size": k[1], "launches": len(f), "read_bytes_per_launch": fm, "write_bytes_per_launch": wm,
                 "hbm_bytes_per_launch": fm + wm})
cal = [r for r in rows if "vec_sadd_kernel" in r["kernel"]]
summary = {"note": __doc__, "calibration_vec_sadd_read_over_write": (cal[0]["read_bytes_per_launch"] / cal[0]["write_bytes_per_launch"]) if cal else None,
           # every launch population of the operator kernel is kept (bench.py averages `traffic` over the same launches
           # as `achieved`), of the other kernels the 24 largest
           "kernels": [r for i, r in enumerate(rows) if i < 24 or "lattice_apply_kernel" in r["kernel"]]}
json.dump(summary, open(out, "w"), indent=1)
for r in rows[:10]:
    print(f'{r["kernel"][:70]:70s} grid {r["grid_size"]:9d} read {r["read_bytes_per_launch"]/1e6:9.1f} MB write {r["write_bytes_per_launch"]/1e6:9.1f} MB')
print("calibration read/write of vec_sadd:", summary["calibration_vec_sadd_read_over_write"])
