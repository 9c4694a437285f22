#!/bin/bash
# host-side savings of the step (pooled-output allocation, NHWC average pool, fused loss compositions): tests, then the driver's command
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_handnet_gpu.py tests/test_resnet_golden.py tests/test_bnact_gpu.py tests/test_driver_gpu.py tests/test_benchsize_gpu.py tests/test_epoch_and_checkpoint.py -m gpu -q --timeout 600 -x 2>&1 | tail -4
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$out/r04d_bench_c2.err | tail -1 > $out/r04d_bench_c2.json
python3 - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04d_bench_c2.json"))
print("c2 %.0f img/s %.3f ms dec %s" % (d["value"], d["ms_per_step"], d["decoder_roofline"]["frac"]))
for l in d["secondary"]["legs"]:
    print("  ", l["config"], l["hipgraph"], "%.0f img/s %.3f ms" % (l["value"], l["ms_per_step"]))
PY
CFG=c2 python tools/r04/find_copies.py 2>/dev/null | head -24
