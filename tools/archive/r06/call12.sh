#!/bin/bash
# round 6, call 12: pair-min resolve kernel in XCD-aware order + look-before-atomic: parity, kbench, PMC traffic
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pairmin_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/kbench.py chamfer 2>/dev/null | grep '^{' | tee $out/r06_kbench_chamfer_final.txt
bash tools/archive/r05/pmc_pairmin_legs.sh 2>&1 | tee $out/r06_pmc_pairmin_legs.txt
cd /tmp; rm -rf /tmp/pk; OBMAN_KBENCH_NPRED=16050 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python $GRAFT_REPO_ROOT/tools/kbench.py chamfer > /dev/null 2>&1
python3 - "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "pairmin" in r["Name"] or "fill_u32" in r["Name"] or "rowmean" in r["Name"]:
        print("%-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
