#!/bin/bash
# GPU box, last call of the round: the whole -m gpu suite as the driver runs it (-x), smoke(), and the default bench line with its wall time.
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 1200 > $out/r05_final_pytest.log 2>&1
tail -4 $out/r05_final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s)
timeout 900 python bench.py > $out/r05_final_bench.json 2> $out/r05_final_bench.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["legs"][-1])
print([(l.get("config"), l.get("value"), l.get("error")) for l in d.get("secondary", {}).get("legs", [])])
PY
