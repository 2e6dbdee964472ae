set -e -o pipefail
export HRSEG_COMMIT=$1
bash tools/collect_profiles.sh r04 > gpurun_out/collect.log 2>&1 || { tail -20 gpurun_out/collect.log; exit 1; }
tail -3 gpurun_out/collect.log
cp gpurun_out/profiles_r04/r04_*.csv gpurun_out/profiles_r04/r04_meta.json profiles/
python bench.py > gpurun_out/profiles_r04/r04_bench_n1.json 2> gpurun_out/bench_n1.err; tail -2 gpurun_out/bench_n1.err; cut -c1-400 gpurun_out/profiles_r04/r04_bench_n1.json
python bench.py --model unet --no-cpu-baseline > gpurun_out/profiles_r04/r04_bench_unet_n1.json 2> gpurun_out/bench_unet.err; cut -c1-300 gpurun_out/profiles_r04/r04_bench_unet_n1.json
python bench.py --tree class_tree_tl_extended.json --size 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-dedup-line --no-f32-line > gpurun_out/profiles_r04/r04_bench_cfg4_geometry_1gpu.json 2> gpurun_out/bench_cfg4.err; cut -c1-300 gpurun_out/profiles_r04/r04_bench_cfg4_geometry_1gpu.json
