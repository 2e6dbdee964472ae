#!/bin/bash
# Collects the profile set of profiles/README.md on a GPU box: kernel stats (default two-stream run and single-stream),
# HBM traffic and MFMA-busy PMC passes (counters in their own runs, --kernel-trace only), the bench line.
#   [HRSEG_COMMIT=<short hash>] bash tools/collect_profiles.sh <tag>        -> gpurun_out/profiles_<tag>/
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dedup-line --no-bf16-line --no-f32-line"
P="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dedup-line --no-bf16-line --no-f32-line --no-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 $B > $out/ks.log 2>&1
cp $out/ks/ks_kernel_stats.csv $out/${tag}_hrnet_hier_b4_620_kernel_stats.csv
echo "[profiles] kernel stats done"
HRSEG_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks1 -o ks1 -- python3 $B > $out/ks1.log 2>&1
cp $out/ks1/ks1_kernel_stats.csv $out/${tag}_hrnet_hier_b4_620_kernel_stats_single_stream.csv
echo "[profiles] single-stream kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o pf -- python3 $P > $out/pf.log 2>&1
echo "[profiles] FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o pw -- python3 $P > $out/pw.log 2>&1
echo "[profiles] WRITE_SIZE done"
python3 tools/pmc_aggregate.py traffic $out/pf/pf_counter_collection.csv $out/pw/pw_counter_collection.csv > $out/${tag}_pmc_traffic_per_launch.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $out/pm -o pm -- python3 $P > $out/pm.log 2>&1
python3 tools/pmc_aggregate.py mfma $out/pm/pm_counter_collection.csv > $out/${tag}_pmc_mfma_busy.csv
echo "[profiles] MFMA busy done"
HRSEG_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks2 -o ks2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-bf16-line --no-dedup-line --no-f32-line > $out/ks2.log 2>&1
cp $out/ks2/ks2_kernel_stats.csv $out/${tag}_hrnet_hier_b4_620_kernel_stats_single_stream_noprobe.csv
echo "[profiles] single-stream, train steps only: done"
python3 - <<PY > $out/${tag}_meta.json
import json, sys
sys.path.insert(0, ".")
import bench
import os
commit = os.environ.get("HRSEG_COMMIT") or (open("gpurun_out/.head").read().strip() if os.path.exists("gpurun_out/.head") else "?")
print(json.dumps({"csrc_digest": bench._csrc_digest(), "commit": commit}))
PY
rm -rf $out/ks $out/ks1 $out/ks2 $out/pf $out/pw $out/pm
ls -la $out
