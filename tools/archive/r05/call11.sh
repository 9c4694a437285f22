#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_dp_graph_gpu.py tests/test_decoder_gpu.py -m gpu -q --timeout 900 2>&1 | tail -30 > $out/r05_call11_dpgraph.log
grep -n "passed\|failed" $out/r05_call11_dpgraph.log | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$out/r05b_bench_c2.err | tail -1 > $out/r05b_bench_c2.json
python3 - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05b_bench_c2.json"))
print("c2 %.0f img/s %.3f ms dec %s roof %s" % (d["value"], d["ms_per_step"], d["decoder_roofline"]["frac"], d["roofline"]["frac"]))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sample"))
for l in d["secondary"]["legs"]:
    if "error" in l: print("  leg error", l); continue
    print("  ", l["config"], l["hipgraph"], "%.0f img/s %.3f ms" % (l["value"], l["ms_per_step"]), l.get("decoder_roofline", {}).get("frac"))
PY
