#!/bin/bash
# round 6, call 17: head layers as own kernels (csrc/linear.hip): tests, step lines, aten census
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stepops_gpu.py tests/test_handnet_gpu.py tests/test_benchsize_gpu.py tests/test_driver_gpu.py -x -q -m gpu 2>&1 | tail -5 | cut -c1-500
for cfg in c3 c2; do
extra=""; [ $cfg = c3 ] && extra="--encoder-dtype bf16 --decoder-dtype bf16"
timeout 600 python bench.py --in-process --config $cfg $extra --steps 40 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>/dev/null | tail -1 > $out/r06_${cfg}_call17.json
python3 -c "
import json; d=json.load(open('$out/r06_${cfg}_call17.json')); print('$cfg', d['ms_per_step'], d['value'], d['host_enqueue_ms']['median'], d['step_gpu_ms']['median'])"
done
CFG=c3 ENC=bf16 DEC=bf16 timeout 600 python tools/archive/r06/aten_ops.py 2>/dev/null | tail -30 | tee $out/r06_aten_ops_c3_bf16_final.txt
cd /tmp; rm -rf /tmp/pl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python $GRAFT_REPO_ROOT/bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 30 --warmup 6 --no-cpu-baseline --secondary-steps 0 > /dev/null 2>&1
python3 - "$(find /tmp/pl -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "linear_" in r["Name"] or "Cijk" in r["Name"]:
        print("%-80s calls %5s avg %8.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
