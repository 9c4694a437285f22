#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE per known-byte stream of 4 / 8 / 12 / 16 bytes per lane (tools/ubench/fetch_calib.hip).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/pmc_fetch_calib.sh'
src=$GRAFT_REPO_ROOT/tools/ubench/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $src 2>/dev/null || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_cal_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_cal_$c -- /tmp/fetch_calib > /tmp/cal_$c.log 2>&1
  f=$(find /tmp/pmc_cal_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
BYTES = 3 << 30
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2]]
acc = collections.OrderedDict()
for r in rows:
    acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    kib = sum(v) / len(v)
    print("%-10s %-48s dispatches %d  counter %.0f KiB  = %.3f x the %d-byte stream" % (sys.argv[2], k[:48], len(v), kib, kib * 1024 / BYTES, BYTES))
PY
done
