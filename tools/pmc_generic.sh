#!/bin/bash
# GPU box: per-kernel averages of a list of PMC counters, one rocprofv3 pass per counter (only --kernel-trace beside --pmc).
#   gpurun -- 'bash tools/pmc_generic.sh "<kernel-name substring>[|<substring>...]" "<COUNTER> <COUNTER> ..." <command ...>'
# ENV passes environment assignments to the profiled command (e.g. ENV="OBMAN_KBENCH_ONE=1").
match=$1; shift
counters=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in $counters; do
  rm -rf /tmp/pmc_g_$c
  env $ENV timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_g_$c -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc_g_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c "$match" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
keys = sys.argv[3].split("|")
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and any(k in r["Kernel_Name"] for k in keys):
        acc[r["Kernel_Name"][:100]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s %-100s n=%4d avg=%.6g" % (sys.argv[2], k, len(v), sum(v) / len(v)))
PY
done
