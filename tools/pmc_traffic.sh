#!/bin/bash
# GPU box: HBM-side bytes per pairmin_fwd_kernel launch (separate --pmc passes, counters only with --kernel-trace).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/pmc_traffic.sh 642'
n=${1:-642}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  OBMAN_KBENCH_NPRED=$n timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/kbench.py chamfer > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "pairmin_fwd_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print(sys.argv[2], "dispatches", len(vals), "avg_KB", round(sum(vals) / max(len(vals), 1), 1))
PY
done
