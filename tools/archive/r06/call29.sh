#!/bin/bash
# round 6, call 29: is gy2's k loop bound by the LDS reads of its per-channel constants (14 KB per k-step and wave)?  Ablation build
# -DOBMAN_ABL_GH3CONST: one channel's constants for all eight elements (wrong results), against the product library on the same box
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
csrc=obman_train_amd/csrc
cp $csrc/libobman_hip.so /tmp/libobman_hip.keep
cd /tmp && export TMPDIR=/tmp
for lib in product ablation; do
  if [ $lib = ablation ]; then cp $GRAFT_REPO_ROOT/$csrc/libobman_hip_ablation.so $GRAFT_REPO_ROOT/$csrc/libobman_hip.so; fi
  rm -rf /tmp/prof_dec
  OBMAN_KBENCH_DEC=bf16:25 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== bf16:25 $lib"; grep '^{' /tmp/kb.log || tail -5 /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n and float(r["AverageNs"]) > 150000:
        print("%-86s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $out/r06_abl_gh3const.txt
cp /tmp/libobman_hip.keep $GRAFT_REPO_ROOT/$csrc/libobman_hip.so
