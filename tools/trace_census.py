"""Per-launch census of a rocprofv3 --kernel-trace csv: dispatches grouped by (kernel, grid size, workgroup size) -> count per
step, mean / min / max duration, total per step.  Shows WHICH launches of a kernel family are the slow ones (the --stats summary
averages a family's very different problems together).
    python tools/trace_census.py <..._kernel_trace.csv> <steps> [name filter]"""
import csv
import re
import sys
from collections import defaultdict

path, steps = sys.argv[1], float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
groups = defaultdict(list)
for row in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
    if flt and flt not in name:
        continue
    dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    grid = row.get("Grid_Size") or row.get("Grid_Size_X") or "?"
    wg = row.get("Workgroup_Size") or row.get("Workgroup_Size_X") or "?"
    groups[(name, grid, wg)].append(dur)
rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
total = sum(sum(v) for v in groups.values()) / steps / 1e3
print(f"{total:.2f} ms of kernel time per step in {sum(len(v) for v in groups.values()) / steps:.0f} launches")
print("ms/step  launches/step  mean_us  min_us  max_us  grid  wg  kernel")
for (name, grid, wg), v in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 80]:
    print(f"{sum(v) / steps / 1e3:7.3f}  {len(v) / steps:6.1f}  {sum(v) / len(v):8.1f}  {min(v):7.1f}  {max(v):7.1f}  {grid:>9}  {wg:>4}  {name[:90]}")
