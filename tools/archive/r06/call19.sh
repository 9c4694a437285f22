#!/bin/bash
# round 6, call 19: head layers own kernels vs library GEMMs, same box: step lines and per-kernel times
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 1 0; do
for cfg in c3 c2; do
extra=""; [ $cfg = c3 ] && extra="--encoder-dtype bf16 --decoder-dtype bf16"
OBMAN_MLP=$v timeout 600 python bench.py --in-process --config $cfg $extra --steps 60 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>/dev/null | tail -1 > $out/r06_${cfg}_mlp$v.json
python3 -c "
import json; d=json.load(open('$out/r06_${cfg}_mlp$v.json')); print('OBMAN_MLP=$v $cfg', round(d['ms_per_step'],3), round(d['value']), 'host', round(d['host_enqueue_ms']['median'],2), 'gpu', round(d['step_gpu_ms']['median'],3))"
done; done; done
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
rm -rf /tmp/pl; OBMAN_MLP=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python $GRAFT_REPO_ROOT/bench.py --in-process --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 30 --warmup 6 --no-cpu-baseline --secondary-steps 0 > /dev/null 2>&1
echo "== OBMAN_MLP=$v"
python3 - "$(find /tmp/pl -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "small_gemm" in n or "Cijk" in n or "threshold" in n.lower() or "clamp" in n.lower():
        print("%-90s calls %5s avg %8.1f us total %8.1f ms" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
