#!/bin/bash
# round 6: bf16 decoder A/B on one box: parity tests, then per-kernel rocprof of tools/kbench.py decoder for each knob setting
#   KNOB=OBMAN_DEC_TN2W VARIANTS="1 0" TAG=tn2w bash tools/archive/r06/dec_bf16.sh
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
KNOB=${KNOB:-OBMAN_DEC_TN2W}
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_decoder_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 600 -x -k "${TESTK:-bf16}" 2>&1 | tail -5
fi
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-1 0}; do
for cfg in ${CFGS:-"bf16:25"}; do
  rm -rf /tmp/prof_dec
  env $KNOB=$v OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== $cfg $KNOB=$v"; grep '^{' /tmp/kb.log || tail -5 /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n and float(r["AverageNs"]) > 10000:
        print("%-86s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done 2>&1 | tee $out/r06_dec_bf16_${TAG:-a}.txt
