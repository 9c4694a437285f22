#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3; do
  rm -rf /tmp/prof_dec
  OBMAN_F2_DBG=$dbg OBMAN_KBENCH_DEC=f32:25 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  echo "== dbg $dbg"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "rows2f" in n:
        print("%-80s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:80], float(r["AverageNs"]) / 1e3))
PY
done
