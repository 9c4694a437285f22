#!/bin/bash
# round 6, call 26: ablation build - dA's A operand read AS IF fragment-major (OBMAN_R2_ABL=9, wrong results): what would the layout buy?
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
csrc=obman_train_amd/csrc
cp $csrc/libobman_hip.so /tmp/libobman_hip.keep
cp $csrc/libobman_hip_ablation.so $csrc/libobman_hip.so
cd /tmp && export TMPDIR=/tmp
for abl in 0 9; do
for cfg in bf16:25 bf16:25:4; do
  rm -rf /tmp/prof_dec
  OBMAN_R2_ABL=$abl OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== $cfg OBMAN_R2_ABL=$abl"; grep '^{' /tmp/kb.log || tail -5 /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n and float(r["AverageNs"]) > 150000:
        print("%-86s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done 2>&1 | tee $out/r06_abl_fragmajor.txt
cp /tmp/libobman_hip.keep $GRAFT_REPO_ROOT/$csrc/libobman_hip.so
