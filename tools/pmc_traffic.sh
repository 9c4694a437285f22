#!/bin/bash
# GPU box: HBM-side bytes per Chamfer forward launch (separate --pmc passes, counters only with --kernel-trace).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/pmc_traffic.sh 642 [rotate]'
# rotate > 0: the micro-benchmark cycles through that many input sets (64 x 1.6 MB = 100 MB at 642 points is still inside the
# 256 MB Infinity Cache, 256 sets are not), so re-used inputs cannot be served on-die and FETCH_SIZE is what really reaches the kernel.
n=${1:-642}
rot=${2:-0}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  OBMAN_KBENCH_ROTATE=$rot OBMAN_KBENCH_NPRED=$n timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/kbench.py chamfer > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c $rot <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2]]
for key in ("pairmin_s5_kernel", "pairmin_fwd_kernel", "rowmean2_kernel", "pairmin_bwd_kernel"):
    vals = [float(r["Counter_Value"]) for r in rows if key in r["Kernel_Name"]]
    if vals:
        print(sys.argv[2], "rotate", sys.argv[3], key, "dispatches", len(vals), "avg_KB", round(sum(vals) / len(vals), 1))
PY
done
