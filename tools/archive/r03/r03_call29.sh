#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 600 -x -k "decoder" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for m in 1; do
  rm -rf /tmp/prof_dec
  OBMAN_R2_H2MODE=$m OBMAN_KBENCH_DEC=bf16:25 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  grep '^{' /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rows2" in r["Name"] or "tn" in r["Name"][:40] or "gh2" in r["Name"] or "gh3" in r["Name"] or "l4_" in r["Name"] or "prescale" in r["Name"]:
        print(r["Name"][28:80], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
