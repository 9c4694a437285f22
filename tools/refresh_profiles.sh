#!/bin/bash
# Run on the GPU box (gpurun): regenerates the measurements committed under profiles/ for the current build.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01f'
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py 2>/dev/null | tail -1 > $out/${tag}_bench_c2.json
timeout 300 python bench.py --config c3 --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_c3_f32.json
timeout 300 python bench.py --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_c3_bf16.json
OBMAN_KBENCH_C3=1 timeout 600 python tools/kbench.py all 2>/dev/null | grep '^{' > $out/${tag}_kbench.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_c2_kernel_stats.csv
grep '^{' /tmp/prof.log | tail -1 > $out/${tag}_bench_c2_profiled.json
ls -la $out | tail -8
