#!/bin/bash
# GPU call 1 of round 3: full GPU test suite (new parity tests record their measured errors), the 1-rank data-parallel overhead
# three ways (plain / stolen gradients + pack copy / in-place accumulation), per-kernel statistics of the plain and the
# --force-dist step for a by-name diff, and the per-kernel micro-benchmarks before this round's kernel changes.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_measured.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $out/r03_pytest1.log 2>&1
echo "pytest rc=$?" >> $out/r03_pytest1.log
tail -5 $out/r03_pytest1.log
for v in plain dist inplace; do
  case $v in plain) fl="";; dist) fl="--force-dist";; inplace) fl="--force-dist --dp-accumulate-in-place";; esac
  timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $fl --trace $out/r03_dp_${v}_trace.json 2>/dev/null | tail -1 > $out/r03_dp_${v}.json
done
python - <<'PY'
import json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for v in ("plain", "dist", "inplace"):
    try:
        d = json.load(open(os.path.join(out, "r03_dp_%s.json" % v)))
        t = json.load(open(os.path.join(out, "r03_dp_%s_trace.json" % v)))["timed"]
        host = sorted(x["host_ms"] for x in t)[len(t) // 2]
        print(v, "ms/step %.3f" % d["ms_per_step"], "gpu median %.3f" % d["step_gpu_ms"]["median"], "host median %.3f" % host, "img/s %.0f" % d["value"])
    except Exception as e:
        print(v, "failed", e)
PY
timeout 60 tools/ubench/valu_rate > $out/r03_valu_rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in plain dist; do
  case $v in plain) fl="";; dist) fl="--force-dist";; esac
  rm -rf /tmp/prof_$v
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline $fl > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/r03_dpdiff_${v}_kernel_stats.csv
done
ls -la $out | tail -12
