#!/bin/bash
# GPU box: the 1-rank RCCL graph bench command 16 times after the watchdog drain (trainer._let_the_watchdog_reap), the two graph
# test files, then the configs[2] / configs[4] legs with the owner-walk contact backward (per-kernel time from rocprofv3 --stats).
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
export OBMAN_MANO_SYNTHETIC=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
bad=0
for i in $(seq 1 16); do
  MASTER_PORT=$((29700 + i)) timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --batch 8 --image-size 128 --precondition-max 9 \
    --force-dist --graph --no-cpu-baseline --secondary-steps 0 > $out/flake2_$i.out 2> $out/flake2_$i.err
  rc=$?; [ $rc -ne 0 ] && bad=$((bad + 1)); echo -n "$rc "
done
echo; echo "aborted runs: $bad of 16"
timeout 900 python -m pytest tests/test_dp_graph_gpu.py tests/test_bench_world2_gpu.py -q -m gpu 2>&1 | tail -3
unset OBMAN_MANO_SYNTHETIC
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $GRAFT_REPO_ROOT/bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 30 --warmup 6 --no-cpu-baseline --secondary-steps 0 > $out/r05_c3_contact.json 2> $out/r05_c3_contact.err
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1)
grep -i "contact_\|contains_" "$f" | cut -c1-220
cd $GRAFT_REPO_ROOT
for leg in c3:bf16:bf16:0 c5:bf16:bf16:0; do
  timeout 400 python bench.py --leg $leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('leg','config','value','ms_per_step','error')})"
done
