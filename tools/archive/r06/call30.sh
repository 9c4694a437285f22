#!/bin/bash
# round 6, call 30: gh2 pass two rows per step, l1_reduce2 batched loads
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_decoder_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 900 -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for cfg in bf16:25 bf16:25:4; do
  rm -rf /tmp/prof_dec
  OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== $cfg"; grep '^{' /tmp/kb.log || tail -5 /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n and float(r["AverageNs"]) > 15000:
        print("%-86s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $out/r06_gh2_two_rows.txt
