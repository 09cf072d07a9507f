"""Aggregate a rocprofv3 --kernel-trace results.db by (kernel, number of workgroups): share, launches, average duration."""
import sqlite3, re, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = db.execute("select name, grid_x, workgroup_x, (end-start) from kernels").fetchall()
agg = defaultdict(list)
for n, g, w, d in rows:
    n = re.sub(r"\(.*", "", n.replace("mgamd::", "").replace("void ", ""))
    agg[(n, g // w if w else 0)].append(d)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e6:.2f} ms in {len(rows)} launches")
for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{sum(v)/tot*100:5.1f}%  n={len(v):5d} avg={sum(v)/len(v)/1e3:8.1f}us  wgs={g:7d} {n[:100]}")
