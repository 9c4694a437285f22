#!/bin/bash
# round 5: bf16 decoder, rows3 (LDS-DMA A tile) vs rows2 on one box: parity tests, then per-kernel rocprof of tools/kbench.py decoder
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_decoder_gpu.py -m gpu -q --timeout 600 -x -k "${TESTK:-bf16}" 2>&1 | tail -5
fi
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1}; do
for cfg in ${CFGS:-"bf16:25"}; do
  rm -rf /tmp/prof_dec
  OBMAN_DEC_ROWS3=$v OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== $cfg OBMAN_DEC_ROWS3=$v"; grep '^{' /tmp/kb.log || tail -5 /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n and float(r["AverageNs"]) > 15000:
        print("%-86s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done 2>&1 | tee $out/r05_dec_bf16_${TAG:-a}.txt
