#!/bin/bash
# GPU call 2 of round 3: full GPU suite on the new Chamfer path + DP exchange plan, Chamfer A/B (S5 on / off), rocprofv3 kernel
# durations of the Chamfer launches, PMC traffic with rotating inputs, 1-rank DP overhead again.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_measured.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $out/r03_pytest2.log 2>&1
echo "pytest rc=$?" >> $out/r03_pytest2.log
tail -4 $out/r03_pytest2.log
for s5 in 1 0; do
  OBMAN_PM_S5=$s5 OBMAN_KBENCH_NPRED=642 timeout 300 python tools/kbench.py chamfer 2>/dev/null | grep '^{' >> $out/r03b_kbench_chamfer.txt
done
timeout 300 python tools/kbench.py chamfer 2>/dev/null | grep '^{' >> $out/r03b_kbench_chamfer.txt
cat $out/r03b_kbench_chamfer.txt
for v in plain dist; do
  case $v in plain) fl="";; dist) fl="--force-dist";; esac
  timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $fl --trace $out/r03b_dp_${v}_trace.json 2>/dev/null | tail -1 > $out/r03b_dp_${v}.json
done
python - <<'PY'
import json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for v in ("plain", "dist"):
    try:
        d = json.load(open(os.path.join(out, "r03b_dp_%s.json" % v)))
        t = json.load(open(os.path.join(out, "r03b_dp_%s_trace.json" % v)))["timed"]
        host = sorted(x["host_ms"] for x in t)[len(t) // 2]
        r = d["roofline"]
        print(v, "ms/step %.3f" % d["ms_per_step"], "gpu median %.3f" % d["step_gpu_ms"]["median"], "host median %.3f" % host, "img/s %.0f" % d["value"],
              "chamfer fwd us %.2f bwd us %s frac %.4f valu %.4f" % (r["avg_launch_us"], r["backward"]["avg_launch_us"], r["frac"], r["valu"]["frac"]))
    except Exception as e:
        print(v, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ch
OBMAN_KBENCH_NPRED=642 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ch -- python $GRAFT_REPO_ROOT/tools/kbench.py chamfer > /dev/null 2>&1
f=$(find /tmp/prof_ch -name "*kernel_stats.csv" | head -1)
cp "$f" $out/r03b_chamfer_kernel_stats.csv
grep -i "pairmin\|rowmean" $out/r03b_chamfer_kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh 642 0 > $out/r03b_chamfer_pmc_642.txt 2>&1
bash tools/pmc_traffic.sh 642 256 >> $out/r03b_chamfer_pmc_642.txt 2>&1
cat $out/r03b_chamfer_pmc_642.txt
