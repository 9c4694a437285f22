#!/bin/bash
# GPU box: kernel time of the owner-walk contact backward inside the configs[2] and configs[4] steps, and 24 more runs of the 1-rank RCCL graph command.
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5; do
  rm -rf /tmp/prof_$cfg
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -- python $GRAFT_REPO_ROOT/bench.py --in-process --config $cfg --encoder-dtype bf16 --decoder-dtype bf16 --steps 20 --warmup 4 --no-cpu-baseline --secondary-steps 0 > /dev/null 2>&1
  f=$(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $cfg <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "contact_" in n or "contains_" in n:
        print(sys.argv[2], n.split("(")[1 if n.startswith("(") else 0][:40] if False else n[:60].replace("(anonymous namespace)::", ""), "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
done
cd $GRAFT_REPO_ROOT
export OBMAN_MANO_SYNTHETIC=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
bad=0
for i in $(seq 1 24); do
  MASTER_PORT=$((29800 + i)) timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --batch 8 --image-size 128 --precondition-max 9 \
    --force-dist --graph --no-cpu-baseline --secondary-steps 0 > $out/flake3.out 2> $out/flake3_$i.err
  rc=$?; [ $rc -ne 0 ] && bad=$((bad + 1)); echo -n "$rc "
done
echo; echo "aborted runs: $bad of 24"
