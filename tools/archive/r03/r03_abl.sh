#!/bin/bash
# GPU box: ablations of the h2 rows2 kernel's k loop (OBMAN_R2_ABL, measurement only).  Needs the ablation library:
#   here: bash tools/ablate_gemm.sh build   (-DOBMAN_ABLATION -> csrc/libobman_hip_ablation.so, git-ignored, travels)
L=$GRAFT_REPO_ROOT/obman_train_amd/csrc; cp $L/libobman_hip.so /tmp/libobman_hip.keep; cp $L/libobman_hip_ablation.so $L/libobman_hip.so
cd /tmp && export TMPDIR=/tmp
for m in ${ABLS:-0 1 2 3 4 5}; do
  rm -rf /tmp/prof_dec
  OBMAN_R2_ABL=$m OBMAN_KBENCH_DEC=bf16:12 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $m <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rows2" in r["Name"] and "BGridFeatPre" in r["Name"]:
        print("ABL", sys.argv[2], "h2 avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/libobman_hip.keep $L/libobman_hip.so
