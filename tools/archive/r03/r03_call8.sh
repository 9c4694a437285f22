#!/bin/bash
# GPU call: rows2 decoder kernels (one fragment per wave, deep request queue): parity + per-kernel durations, A/B vs rows2 off.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 600 -x -k "decoder" > $out/r03_pytest8.log 2>&1
echo "pytest rc=$?" >> $out/r03_pytest8.log
tail -4 $out/r03_pytest8.log
cd /tmp && export TMPDIR=/tmp
for r2 in 1 0; do
  rm -rf /tmp/prof_dec
  OBMAN_DEC_ROWS2=$r2 OBMAN_KBENCH_DEC=bf16:25 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb_$r2.log 2>&1
  grep '^{' /tmp/kb_$r2.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/r03h_dec_bf16_c3_kernel_stats_rows2_$r2.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    print("%-100s calls %5s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
