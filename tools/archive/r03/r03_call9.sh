#!/bin/bash
# GPU call: hipGraph step (parity test + host enqueue time eager vs replay on three configurations)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_driver_gpu.py -m gpu -q --timeout 600 > $out/r03_pytest9.log 2>&1
echo "pytest rc=$?" >> $out/r03_pytest9.log
tail -6 $out/r03_pytest9.log
run() {  # tag, flags
  timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $2 2>$out/r03i_$1.err | tail -1 > $out/r03i_$1.json
  python - <<PY
import json
try:
    d = json.load(open("$out/r03i_$1.json"))
    print("$1", "ms/step %.3f" % d["ms_per_step"], "gpu median %.3f" % d["step_gpu_ms"]["median"], "host median %.3f max %.3f" % (d["host_enqueue_ms"]["median"], d["host_enqueue_ms"]["max"]), "img/s %.0f" % d["value"])
except Exception as e:
    print("$1 failed", e); print(open("$out/r03i_$1.err").read()[-1500:])
PY
}
run c2_eager ""
run c2_graph "--graph"
run c2bf16enc_eager "--encoder-dtype bf16"
run c2bf16enc_graph "--encoder-dtype bf16 --graph"
run c3bf16_eager "--config c3 --encoder-dtype bf16 --decoder-dtype bf16"
run c3bf16_graph "--config c3 --encoder-dtype bf16 --decoder-dtype bf16 --graph"
