#!/bin/bash
# round 6, call 3: the wide dW2 tile (tn2w) - parity, per-kernel times A/B, configs[2] / configs[4] step; step-operator tests
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_stepops_gpu.py -x -q -m gpu 2>&1 | tail -15
KNOB=OBMAN_DEC_TN2W VARIANTS="1 0" TAG=tn2w bash tools/archive/r06/dec_bf16.sh
cd $GRAFT_REPO_ROOT
for cfg in c3 c5; do
timeout 600 python bench.py --in-process --config $cfg --encoder-dtype bf16 --decoder-dtype bf16 --steps 30 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>$out/r06_${cfg}_quick.err | tail -1 > $out/r06_${cfg}_tn2w.json
python3 -c "
import json; d=json.load(open('$out/r06_${cfg}_tn2w.json')); print('$cfg bf16', d['ms_per_step'], d['value'], (d.get('decoder_roofline') or {}).get('frac'))"
done
timeout 900 python -m pytest tests/test_handnet_gpu.py tests/test_epoch_and_checkpoint.py tests/test_memory_safety_gpu.py tests/test_dp_nccl_gpu.py tests/test_dp_graph_gpu.py -x -q -m gpu 2>&1 | tail -15
