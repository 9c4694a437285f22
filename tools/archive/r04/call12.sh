#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export OBMAN_BENCH_TRACE=1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04a_bench_c2.json 2> $out/r04a_bench_c2.err
echo "bench rc=$?"; tail -3 $out/r04a_bench_c2.err
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04a_bench_c2.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["decoder_roofline"]["fwd_us"], d["decoder_roofline"]["bwd_us"], d["decoder_roofline"]["frac"])
for l in d.get("secondary",{}).get("legs",[]):
    print(l["config"], l["dtype"][:12], l["hipgraph"], round(l["value"],1), round(l["ms_per_step"],3), [ (round(e["avg_launch_us"],1), round(e["frac"],4), round(e["valu_frac"],3)) for e in (l.get("roofline") or {}).get("per_launch",[])], (l.get("decoder_roofline") or {}).get("frac"))
print("cpu", d["cpu_baseline"]["value"])
PY
bash tools/pmc_fetch_calib.sh 2>&1 | tee $out/r04_fetch_calib.txt
