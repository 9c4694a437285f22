#!/bin/bash
# Run on the GPU box (gpurun): regenerates the measurements committed under profiles/ for the current build.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r02'
# Everything lands in gpurun_out/<tag>_*; tools/summarize_rocprof.py + the notes in profiles/ are written from those files.
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
# 1. the driver's exact command (cold MIOpen state on a fresh box), then the other configurations
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --trace $out/${tag}_bench_c2_trace.json 2>/dev/null | tail -1 > $out/${tag}_bench_c2.json
timeout 600 python bench.py --config c3 --steps 15 --warmup 4 --cpu-seconds 12 2>/dev/null | tail -1 > $out/${tag}_bench_c3_f32.json
timeout 600 python bench.py --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_c3_bf16.json
timeout 600 python bench.py --config c5 --encoder-dtype bf16 --decoder-dtype bf16 --steps 6 --warmup 2 --precondition-max 10 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_c5_bf16.json
# 2. per-kernel micro-benchmarks
OBMAN_KBENCH_C3=1 timeout 600 python tools/kbench.py all 2>/dev/null | grep '^{' > $out/${tag}_kbench.txt
cd /tmp && export TMPDIR=/tmp
# 3. rocprofv3 per-kernel statistics of the configs[1] step
rm -rf /tmp/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_c2_kernel_stats.csv
grep '^{' /tmp/prof.log | tail -1 > $out/${tag}_bench_c2_profiled.json
# 4. the data-parallel machinery on one rank: RCCL all-reduce kernels and the streams they run on
rm -rf /tmp/prof_dp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dp -- python $GRAFT_REPO_ROOT/bench.py --force-dist --steps 12 --warmup 4 --precondition-max 10 --no-cpu-baseline > /tmp/prof_dp.log 2>&1
f=$(find /tmp/prof_dp -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/summarize_dp_trace.py "$f" > $out/${tag}_force_dist_trace.md 2>&1
grep '^{' /tmp/prof_dp.log | tail -1 > $out/${tag}_bench_c2_force_dist.json
# 5. rocprofv3 per-kernel statistics of the bf16 decoder at configs[2] size, and its HBM-side traffic / wave-state counters
rm -rf /tmp/prof_dec
OBMAN_KBENCH_DEC=bf16:25 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /dev/null 2>&1
f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_dec_bf16_c3_kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash tools/pmc_dec.sh bf16:25 > $out/${tag}_pmc_dec_bf16_c3.txt 2>&1
bash tools/pmc_traffic.sh 642 > $out/${tag}_chamfer_pmc_642.txt 2>&1
bash tools/pmc_traffic.sh 16050 > $out/${tag}_chamfer_pmc_16050.txt 2>&1
ls -la $out | tail -20
