#!/bin/bash
# round 6, call 2: step operators (K11-K14), new parity tests, watchdog probe, c3 bf16 leg quick numbers
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 300 python tools/archive/r06/watchdog_probe.py > $out/r06_watchdog_probe.txt 2>&1; tail -12 $out/r06_watchdog_probe.txt
timeout 900 python -m pytest tests/test_stepops_gpu.py tests/test_handnet_gpu.py tests/test_epoch_and_checkpoint.py tests/test_memory_safety_gpu.py -x -q -m gpu > $out/r06_call2_pytest.log 2>&1; tail -15 $out/r06_call2_pytest.log
timeout 900 python -m pytest tests/test_contact_gpu.py -x -q -m gpu -k adversarial > $out/r06_call2_adv.log 2>&1; tail -15 $out/r06_call2_adv.log
for i in 1 2; do
timeout 600 python bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 40 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>$out/r06_c3_quick.err | tail -1 > $out/r06_c3_quick_$i.json
python3 -c "
import json; d=json.load(open('$out/r06_c3_quick_$i.json')); print('c3 bf16', d['ms_per_step'], d['value'], d.get('step_gpu_ms'))"
done
CFG=c3 ENC=bf16 DEC=bf16 timeout 600 python tools/archive/r06/aten_ops.py > $out/r06_aten_ops_c3_bf16_after.txt 2>$out/r06_aten_ops.err; tail -32 $out/r06_aten_ops_c3_bf16_after.txt
timeout 1500 python -m pytest tests/test_parity_evidence_gpu.py -x -q -m gpu -k "bf16" > $out/r06_call2_bf16par.log 2>&1; tail -15 $out/r06_call2_bf16par.log
