#!/bin/bash
# round 6, call 27: MIOpen find modes for the encoder's convolutions (never explored): steady-state step time and first-run cost
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for mode in default 1 3 2; do
  for cfg in "" "--config c3 --encoder-dtype bf16 --decoder-dtype bf16"; do
    rm -rf ~/.config/miopen ~/.cache/miopen 2>/dev/null
    t0=$(date +%s.%N)
    if [ "$mode" = default ]; then
      timeout 900 python bench.py --in-process $cfg --steps 30 --warmup 8 --no-cpu-baseline --secondary-steps 0 2>/dev/null | tail -1 > /tmp/line.json
    else
      MIOPEN_FIND_MODE=$mode timeout 900 python bench.py --in-process $cfg --steps 30 --warmup 8 --no-cpu-baseline --secondary-steps 0 2>/dev/null | tail -1 > /tmp/line.json
    fi
    t1=$(date +%s.%N)
    python3 -c "
import json; d=json.load(open('/tmp/line.json')); print('MIOPEN_FIND_MODE=$mode', '$cfg'[:12], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'gpu median', round(d['step_gpu_ms']['median'],3), 'wall', round($t1-$t0,1))"
  done
done 2>&1 | tee $out/r06_miopen_find_modes.txt
