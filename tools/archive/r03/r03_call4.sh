#!/bin/bash
# GPU call 4: rows2 decoder kernels (parity + A/B + per-kernel durations), Chamfer S5 with the pairwise mean exchange.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py tests/test_fullsize_gpu.py tests/test_pairmin_gpu.py -m gpu -q --timeout 600 -x > $out/r03_pytest4.log 2>&1
echo "pytest rc=$?" >> $out/r03_pytest4.log
tail -15 $out/r03_pytest4.log
for r2 in 1 0; do
  OBMAN_DEC_ROWS2=$r2 OBMAN_KBENCH_DEC=bf16:25 timeout 300 python tools/kbench.py decoder 2>/dev/null | grep '^{' | sed "s/^{/{\"rows2\": $r2, /" >> $out/r03d_kbench_dec.txt
  OBMAN_DEC_ROWS2=$r2 OBMAN_KBENCH_DEC=bf16:1 timeout 300 python tools/kbench.py decoder 2>/dev/null | grep '^{' | sed "s/^{/{\"rows2\": $r2, /" >> $out/r03d_kbench_dec.txt
done
cat $out/r03d_kbench_dec.txt
OBMAN_KBENCH_NPRED=642 timeout 300 python tools/kbench.py chamfer 2>/dev/null | grep '^{'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dec
OBMAN_KBENCH_DEC=bf16:25 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /dev/null 2>&1
f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
cp "$f" $out/r03d_dec_bf16_c3_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:18]:
    print("%-110s calls %5s avg %9.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf /tmp/prof_ch
OBMAN_KBENCH_NPRED=642 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ch -- python $GRAFT_REPO_ROOT/tools/kbench.py chamfer > /dev/null 2>&1
f=$(find /tmp/prof_ch -name "*kernel_stats.csv" | head -1)
grep -i "pairmin\|rowmean" "$f" | cut -c1-200
