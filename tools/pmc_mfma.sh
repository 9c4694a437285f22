#!/bin/bash
# GPU box: MFMA-pipe utilisation of the decoder GEMMs (separate --pmc passes, counters only with --kernel-trace).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/pmc_mfma.sh f32:1'
case=${1:-f32:1}
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_$c
  OBMAN_KBENCH_DEC=$case timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and ("gemm_" in r["Kernel_Name"] or "_bf16_kernel" in r["Kernel_Name"] or "rows2f_kernel" in r["Kernel_Name"] or "tn3_kernel" in r["Kernel_Name"]):
        acc[r["Kernel_Name"][:95]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(sys.argv[2], "%-95s n=%3d avg=%.4g" % (k, len(v), sum(v) / len(v)))
PY
done
