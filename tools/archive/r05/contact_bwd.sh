#!/bin/bash
# GPU box: contact tests (incl. the bit-exact ordered-scatter test) and the kernel time of contact_bwd_kernel inside the configs[2] / configs[4] steps.
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_contact_gpu.py tests/test_benchsize_gpu.py -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5; do
  rm -rf /tmp/prof_$cfg
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -- python $GRAFT_REPO_ROOT/bench.py --in-process --config $cfg --encoder-dtype bf16 --decoder-dtype bf16 --steps 20 --warmup 4 --no-cpu-baseline --secondary-steps 0 > $out/r05_${cfg}_contact.json 2>/dev/null
  f=$(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $cfg <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "contact_" in n or "contains_" in n:
        print(sys.argv[2], n[:60].replace("(anonymous namespace)::", ""), "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
  tail -1 $out/r05_${cfg}_contact.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['ms_per_step'])"
done
