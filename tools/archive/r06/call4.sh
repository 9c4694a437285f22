#!/bin/bash
# round 6, call 4: side-row kernel fixed (dW2), rows4 experiment (one wave per SIMD on h2): parity + per-kernel times per variant; watchdog probe with the flight recorder
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
PROBE_FR=256 timeout 300 python tools/archive/r06/watchdog_probe.py > $out/r06_watchdog_probe_fr.txt 2>&1; grep -v "^\[W\|amdgpu.ids" $out/r06_watchdog_probe_fr.txt | head -14
for v in 2 4; do
OBMAN_DEC_ROWS4=$v timeout 900 python -m pytest tests/test_decoder_gpu.py -m gpu -q --timeout 600 -x -k "bf16" 2>&1 | tail -3
done
SKIP_TESTS=1 KNOB=OBMAN_DEC_ROWS4 VARIANTS="0 2 5 3 4" TAG=rows4 bash tools/archive/r06/dec_bf16.sh | grep -v "l1_\|prep_\|gemm_tn\|l4w\|reduce"
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 30 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>$out/r06_c3_quick.err | tail -1 > $out/r06_c3_call4.json
python3 -c "
import json; d=json.load(open('$out/r06_c3_call4.json')); print('c3 bf16', d['ms_per_step'], d['value'], (d.get('decoder_roofline') or {}).get('frac'))"
