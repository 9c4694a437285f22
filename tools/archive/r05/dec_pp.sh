#!/bin/bash
# round 5: ping-pong k loop (OBMAN_DEC_PP) A/B on one box, rows3 off so that every rows kernel is the rows2 template
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
OBMAN_DEC_ROWS3=${R3:-0} timeout 900 python -m pytest tests/test_decoder_gpu.py -m gpu -q --timeout 600 -x -k "${TESTK:-bf16 or multi_patch}" 2>&1 | tail -3
fi
cd /tmp && export TMPDIR=/tmp
for pp in ${PPS:-0 1}; do
  rm -rf /tmp/prof_dec
  OBMAN_DEC_PP=$pp OBMAN_DEC_ROWS3=${R3:-0} OBMAN_KBENCH_DEC=${CFG:-bf16:25} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== OBMAN_DEC_PP=$pp ROWS3=${R3:-0}"; grep '^{' /tmp/kb.log || tail -5 /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n and float(r["AverageNs"]) > 100000:
        print("%-86s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $out/r05_dec_pp_${TAG:-a}.txt
