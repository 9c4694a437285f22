#!/bin/bash
# round 6, call 6: decoder after the first-generation weight-gradient GEMM was removed (all decoder tests + every remaining knob);
# data-parallel step eager vs hipGraph under rocprofv3 --kernel-trace; PMC of the bf16 decoder kernels
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_decoder_gpu.py tests/test_fallback_kernels_gpu.py tests/test_atlas_random.py tests/test_benchsize_gpu.py -x -q -m gpu 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  rm -rf /tmp/dp_$mode
  extra=""; [ $mode = graph ] && extra="--graph"
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/dp_$mode -- python $GRAFT_REPO_ROOT/bench.py --force-dist $extra --steps 30 --warmup 5 --no-cpu-baseline --secondary-steps 0 > /tmp/dp_$mode.log 2>&1
  grep '^{' /tmp/dp_$mode.log | tail -1 > $out/r06_bench_c2_force_dist_${mode}_traced.json
  f=$(find /tmp/dp_$mode -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/summarize_dp_trace.py "$f" > $out/r06_dp_trace_$mode.md 2>&1
  python3 - "$f" $mode <<'PY' > $out/r06_dp_queues_$mode.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 10 steps: between mano_fwd_kernel launches
marks = [i for i, r in enumerate(rows) if "mano_fwd_kernel" in r["Kernel_Name"]]
lo, hi = marks[-11], marks[-1]
seg = rows[lo:hi]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
print(sys.argv[2], "10 steps: %.3f ms per step (kernel-trace timestamps)" % ((t1 - t0) / 10 / 1e6))
byq = collections.defaultdict(lambda: [0, 0])
for r in seg:
    q = (r.get("Queue_Id"), r.get("Stream_Id"))
    byq[q][0] += 1
    byq[q][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for q, (n, t) in sorted(byq.items(), key=lambda kv: -kv[1][1]):
    print("  queue %s stream %s: %d kernels, busy %.3f ms per step" % (q[0], q[1], n // 10, t / 10 / 1e6))
# idle gaps of the busiest queue
main = max(byq, key=lambda q: byq[q][1])
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg if (r.get("Queue_Id"), r.get("Stream_Id")) == main]
gap = sum(max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1))
print("  main queue: idle between kernels %.3f ms per step" % (gap / 10 / 1e6))
rc = [r for r in seg if "oneRankReduce" in r["Kernel_Name"] or "nccl" in r["Kernel_Name"].lower()]
print("  RCCL kernels per step %d, avg %.1f us" % (len(rc) // 10, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rc) / max(len(rc), 1) / 1e3))
PY
  cat $out/r06_dp_queues_$mode.txt
done
cd $GRAFT_REPO_ROOT
bash tools/pmc_dec.sh bf16:25 > $out/r06_pmc_dec_bf16_c3.txt 2>&1
grep -c "dec::" $out/r06_pmc_dec_bf16_c3.txt
