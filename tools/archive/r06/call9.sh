#!/bin/bash
# round 6, call 9: dA with the mask operand requested before the k loop: parity + per-kernel times; step lines
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
KNOB=OBMAN_DEC_TN2W VARIANTS="1" TAG=dapf bash tools/archive/r06/dec_bf16.sh | grep -v "l1_\|prep_\|gemm_tn\|reduce"
cd $GRAFT_REPO_ROOT
for cfg in c3 c5; do
timeout 600 python bench.py --in-process --config $cfg --encoder-dtype bf16 --decoder-dtype bf16 --steps 40 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>$out/r06_${cfg}_quick.err | tail -1 > $out/r06_${cfg}_call9.json
python3 -c "
import json; d=json.load(open('$out/r06_${cfg}_call9.json')); print('$cfg bf16', d['ms_per_step'], d['value'], (d.get('decoder_roofline') or {}).get('frac'), (d.get('roofline') or {}).get('avg_launch_us'))"
done
timeout 600 python bench.py --in-process --steps 40 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>/dev/null | tail -1 > $out/r06_c2_call9.json
python3 -c "
import json; d=json.load(open('$out/r06_c2_call9.json')); print('c2 f32', d['ms_per_step'], d['value'], (d.get('decoder_roofline') or {}).get('frac'), (d.get('roofline') or {}).get('avg_launch_us'))"
