#!/bin/bash
# round 6, call 20: the default bench command three times on one box (GC out of the timed region) + bench world-2 self-test
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/r06_driver_cmd_$i.json
python3 -c "
import json; d=json.load(open('$out/r06_driver_cmd_$i.json')); print('run $i', round(d['value'],1), round(d['ms_per_step'],3), 'gpu median', round(d['step_gpu_ms']['median'],3), 'host', round(d['host_enqueue_ms']['median'],2), round(d['host_enqueue_ms']['max'],2), [(l.get('config'), round(l.get('value',0))) for l in d['secondary']['legs']])"
done
timeout 900 python -m pytest tests/test_bench_world2_gpu.py tests/test_dp_nccl_gpu.py -x -q -m gpu 2>&1 | tail -3
