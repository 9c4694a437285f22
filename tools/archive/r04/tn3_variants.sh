#!/bin/bash
# tn3_kernel with the loads / LDS stores of the next k-tile at different MFMA steps (libraries built into tools/r04/variants/)
cd /tmp && export TMPDIR=/tmp
cp $GRAFT_REPO_ROOT/obman_train_amd/csrc/libobman_hip.so /tmp/lib_keep.so
for lib in $GRAFT_REPO_ROOT/tools/r04/variants/*.so; do
  cp $lib $GRAFT_REPO_ROOT/obman_train_amd/csrc/libobman_hip.so
  for cfg in f32:1 f32:25; do
  rm -rf /tmp/prof_dec
  OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  echo "== $(basename $lib) $cfg"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "tn3" in n or "rows2f" in n:
        print("%-60s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:60], float(r["AverageNs"]) / 1e3))
PY
  done
done
cp /tmp/lib_keep.so $GRAFT_REPO_ROOT/obman_train_amd/csrc/libobman_hip.so
