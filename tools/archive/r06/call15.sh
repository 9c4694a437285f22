#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_stepops_gpu.py tests/test_epoch_and_checkpoint.py tests/test_driver_gpu.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-600
CFG=c3 ENC=bf16 DEC=bf16 timeout 600 python tools/archive/r06/host_profile.py 2>&1 | grep -v "amdgpu.ids" | head -75 | cut -c1-200 | tee $out/r06_host_profile_c3.txt
timeout 600 python bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 40 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>/dev/null | tail -1 > $out/r06_c3_call15.json
python3 -c "
import json; d=json.load(open('$out/r06_c3_call15.json')); print('c3 bf16', d['ms_per_step'], d['value'], d['host_enqueue_ms']['median'], d['step_gpu_ms']['median'])"
