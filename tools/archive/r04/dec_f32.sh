#!/bin/bash
# fp32 decoder: parity tests, then per-kernel times (rocprofv3 --kernel-trace --stats) at configs[1] and configs[2] sizes, with
# the second-generation rows kernels (default) and with OBMAN_DEC_ROWS2F=0
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py tests/test_benchsize_gpu.py tests/test_fullsize_gpu.py tests/test_atlas_random.py tests/test_oracle_golden.py -m gpu -q --timeout 600 -x -k "decoder or atlas or pointgen" 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
for cfg in "f32:1" "f32:25"; do
for gen in 1 0; do
  rm -rf /tmp/prof_dec
  OBMAN_DEC_ROWS2F=$gen OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  echo "== $cfg ROWS2F=$gen"; grep '^{' /tmp/kb.log
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  cp $f $out/r04_dec_${cfg/:/_}_rows2f_${gen}_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dec::" in n:
        print("%-90s calls %5s avg %8.1f us" % (n.replace("void dec::", "").replace("dec::", "")[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done
