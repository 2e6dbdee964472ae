"""Per-kernel averages of rocprofv3 --pmc passes (one csv per pass: *_counter_collection.csv).
    python tools/pmc_aggregate.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_pmc_traffic_per_launch.csv
    python tools/pmc_aggregate.py mfma <counter_collection.csv> > profiles/rNN_pmc_mfma_busy.csv
traffic: FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch; on gfx950 FETCH_SIZE counts half the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM section), so the corrected column doubles it.
mfma: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): MFMA-busy cycles are summed over the four SIMDs
of a CU, SQ_BUSY_CU_CYCLES counts per-CU busy cycles (both summed over the dispatch's CUs and XCDs)."""
import csv
import re
import sys
from collections import defaultdict


def load(path):
    """kernel -> counter -> [values per dispatch]"""
    d = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        d[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return d


def mean(v):
    return sum(v) / len(v) if v else float("nan")


mode = sys.argv[1]
if mode == "traffic":
    f, w = load(sys.argv[2]), load(sys.argv[3])
    print("kernel,launches,FETCH_SIZE_KB_raw_avg,FETCH_bytes_MB_corrected_x2,WRITE_SIZE_KB_avg,WRITE_MB")
    rows = []
    for k in f:
        fk, wk = mean(f[k]["FETCH_SIZE"]), mean(w.get(k, {}).get("WRITE_SIZE", []))
        rows.append((len(f[k]["FETCH_SIZE"]) * (fk + (wk if wk == wk else 0)), k, len(f[k]["FETCH_SIZE"]), fk, wk))
    for _, k, n, fk, wk in sorted(rows, reverse=True):
        print('"%s",%d,%.1f,%.2f,%.1f,%.2f' % (k, n, fk, 2 * fk / 1024, wk, wk / 1024))
else:
    d = load(sys.argv[2])
    names = sorted({c for k in d for c in d[k]})
    print("kernel,launches," + ",".join(n + "_avg" for n in names) + ",mfma_busy")
    rows = []
    for k in d:
        n = max(len(v) for v in d[k].values())
        busy = mean(d[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        cu = mean(d[k].get("SQ_BUSY_CU_CYCLES", []))
        frac = busy / (4.0 * cu) if cu and cu == cu and busy == busy else float("nan")
        rows.append((n * (busy if busy == busy else 0), k, n, [mean(d[k].get(c, [])) for c in names], frac))
    for _, k, n, vals, frac in sorted(rows, reverse=True):
        print('"%s",%d,%s,%.4f' % (k, n, ",".join("%.1f" % v for v in vals), frac))
