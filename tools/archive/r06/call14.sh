#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_stepops_gpu.py -q -m gpu 2>&1 | grep "^E  \|passed\|failed" | cut -c1-700 | head -8; done
timeout 900 python -m pytest tests/test_pairmin_gpu.py tests/test_handnet_gpu.py -x -q -m gpu 2>&1 | tail -3
for cfg in c3 c5; do
timeout 600 python bench.py --in-process --config $cfg --encoder-dtype bf16 --decoder-dtype bf16 --steps 40 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>/dev/null | tail -1 > $out/r06_${cfg}_call14.json
python3 -c "
import json; d=json.load(open('$out/r06_${cfg}_call14.json')); print('$cfg bf16', d['ms_per_step'], d['value'])"
done
