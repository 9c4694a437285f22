#!/bin/bash
# round 6, call 8: gh3 materialised once (gy2 / dW3 on plain operands): parity + A/B; the narrow train-mode bf16 case in detail
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py -q -m gpu -k "bf16" 2>&1 | grep "^E  \|passed\|failed" | cut -c1-1800 | head -20
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_benchsize_gpu.py -x -q -m gpu 2>&1 | tail -3
SKIP_TESTS=1 KNOB=OBMAN_DEC_GH3 VARIANTS="1 0" TAG=gh3 bash tools/archive/r06/dec_bf16.sh | grep -v "l1_\|prep_\|gemm_tn\|reduce"
cd $GRAFT_REPO_ROOT
for v in 1 0; do
OBMAN_DEC_GH3=$v timeout 600 python bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 40 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>$out/r06_c3_quick.err | tail -1 > $out/r06_c3_gh3_$v.json
python3 -c "
import json; d=json.load(open('$out/r06_c3_gh3_$v.json')); print('c3 bf16 GH3=$v', d['ms_per_step'], d['value'], (d.get('decoder_roofline') or {}).get('frac'))"
done
