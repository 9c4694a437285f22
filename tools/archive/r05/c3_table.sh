#!/bin/bash
# GPU box: per-step kernel table of the configs[2] bf16 step for the FINAL tree (after the contact-backward change); same command as
# step 3 of tools/refresh_profiles_r05.sh.
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
# an untraced run first: MIOpen's find phase fills its user database, so the traced process holds steady-state kernels only
timeout 300 python $GRAFT_REPO_ROOT/bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/prof3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -- python $GRAFT_REPO_ROOT/bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 60 --warmup 6 --no-cpu-baseline > /tmp/prof3.log 2>&1
cp "$(find /tmp/prof3 -name '*kernel_stats.csv' | head -1)" $out/r05_c3_bf16_kernel_stats.csv
grep '^{' /tmp/prof3.log | tail -1 > $out/r05_bench_c3_bf16_profiled.json
python3 -c "import json; d=json.load(open('$out/r05_bench_c3_bf16_profiled.json')); print(d['value'], d['ms_per_step'])"
