#!/bin/bash
# round 5, call 1: grid-culled inside test - bit-identity vs the all-pairs kernel, contact tests, kernel times, c3 line
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_contains_binned_gpu.py tests/test_contact_gpu.py tests/test_lib_abi.py -m gpu -q --timeout 600 -x 2>&1 | tail -6 | tee $out/r05_call1_pytest.log
timeout 300 python tools/kbench.py contains 2>&1 | grep '^{' | tee $out/r05_kbench_contains.txt
timeout 600 python bench.py --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>$out/r05a_bench_c3.err | tail -1 > $out/r05a_bench_c3_bf16.json
python3 - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05a_bench_c3_bf16.json"))
print("c3 %.0f img/s %.3f ms dec %s" % (d["value"], d["ms_per_step"], d["decoder_roofline"]))
PY
