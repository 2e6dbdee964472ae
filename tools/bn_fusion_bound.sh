#!/bin/bash
# MEASURED upper bounds of the conv <-> BatchNorm fusions of SURVEY section 7 step 5, on one box, same process settings:
# the step with the launches a fusion would remove simply LEFT OUT (engine.py: HRSEG_EXPERIMENT; results are wrong on purpose,
# the kernels, their bytes and the launch structure are those a perfect fusion would leave).  What a real fusion can save is
# the difference to the baseline MINUS what the fused work costs inside the convolution kernels.
#   bash tools/bn_fusion_bound.sh <outdir>
set -e -o pipefail
out=${1:-gpurun_out/bn_bound}
mkdir -p $out
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probe --no-dedup-line --no-bf16-line --no-f32-line"
for rep in 1 2; do
  for exp in none skip_apply1 skip_stats skip_apply1,skip_stats; do
    e=$exp; [ "$exp" = none ] && e=""
    HRSEG_EXPERIMENT=$e python3 $B > $out/bench_${exp}_$rep.log 2>&1
    ms=$(python3 -c "import json,sys; print([json.loads(l)['ms_per_step'] for l in open('$out/bench_${exp}_$rep.log') if l.startswith('{')][-1])")
    echo "experiment=$exp rep=$rep ms_per_step=$ms" | tee -a $out/summary.txt
  done
done
