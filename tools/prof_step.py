"""Per-kernel breakdown of the LAST train step in a rocprofv3 rocpd database (kernel-trace run of bench.py).
usage: python tools/prof_step.py gpurun_out/prof/r_results.db [top_n]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(db.execute("select name,start,end,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "adamw" in r[0]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
span = (step[-1][2] - step[0][1]) / 1e6
ev = sorted((r[1], r[2]) for r in step)
tot = 0
cs, ce = ev[0]
for s, e in ev[1:]:
    if s > ce:
        tot += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
tot += ce - cs
print(f"launches {len(step)}  span {span:.2f} ms  sum of kernel durations {sum(r[2]-r[1] for r in step)/1e6:.2f} ms  "
      f"GPU busy (union) {tot/1e6:.2f} ms  idle {span - tot/1e6:.2f} ms")
d = defaultdict(lambda: [0, 0.0])
for r in step:
    n = re.sub(r"\(.*", "", r[0]).replace("void ", "")
    d[n][0] += 1
    d[n][1] += (r[2] - r[1]) / 1e6
for n, (k, t) in sorted(d.items(), key=lambda x: -x[1][1])[:top]:
    print(f"{t:8.2f} ms {k:5d} x {t/k*1000:8.1f} us  {n[:100]}")
