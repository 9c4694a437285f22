#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py tests/test_benchsize_gpu.py -m gpu -q --timeout 600 -x -k "decoder or pointgen" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-"f32:1" "f32:25"}; do
  rm -rf /tmp/prof_dec
  OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== $cfg"; grep '^{' /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n and float(r["AverageNs"]) > 15000:
        print("%-80s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
