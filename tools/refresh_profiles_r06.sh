#!/bin/bash
# Run on the GPU box (gpurun): regenerates the round-6 measurements committed under profiles/ for the current build.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/refresh_profiles_r06.sh r06'
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
b() { name=$1; shift; timeout 1200 python bench.py "$@" 2>$out/${tag}_bench_$name.err | tail -1 > $out/${tag}_bench_$name.json; python3 - $out/${tag}_bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "ms/step %.3f img/s %.0f host %s roof %s dec %s" % (d["ms_per_step"], d["value"], (d.get("host_enqueue_ms") or {}).get("median"),
          (d.get("roofline") or {}).get("frac"), (d.get("decoder_roofline") or {}).get("frac")))
    for l in (d.get("secondary") or {}).get("legs", []):
        if "error" in l:
            print("   secondary FAILED", l)
            continue
        print("   secondary", l["config"], "graph" if l["hipgraph"] else "eager", "img/s %.0f ms %.3f" % (l["value"], l["ms_per_step"]), "dec", (l.get("decoder_roofline") or {}).get("frac"),
              "traffic", (l.get("roofline") or {}).get("traffic"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
# 1. the driver's exact command (cold MIOpen state on a fresh box; carries the secondary legs, each in its own process), then others
b c2 --gpus 1 --steps 20 --warmup 5
b c3_bf16_graph --graph --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 200 --warmup 5 --no-cpu-baseline
b c3_f32 --config c3 --steps 15 --warmup 4 --no-cpu-baseline
b c2_force_dist --force-dist --steps 20 --warmup 5 --no-cpu-baseline --secondary-steps 0
b c2_force_dist_graph --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline --secondary-steps 0
# 2. per-kernel micro-benchmarks + the LDS reference-tile sweep of the pair-min kernel (BASELINE configs[4])
OBMAN_KBENCH_C3=1 timeout 600 python tools/kbench.py all 2>/dev/null | grep '^{' > $out/${tag}_kbench.txt
timeout 600 python tools/kbench.py tiles 2>/dev/null | grep '^{' >> $out/${tag}_kbench.txt
cd /tmp && export TMPDIR=/tmp
# 3. rocprofv3 per-kernel statistics of the FINAL build, steady state: configs[1], configs[2] (bf16 / bf16) and configs[4] (bf16 / bf16)
table() {  # name, bench args...
  name=$1; shift
  timeout 300 python $GRAFT_REPO_ROOT/bench.py --in-process "$@" --steps 5 --warmup 2 --no-cpu-baseline --secondary-steps 0 > /dev/null 2>&1  # fills MIOpen's find database
  rm -rf /tmp/prof_$name
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $GRAFT_REPO_ROOT/bench.py --in-process "$@" --steps 60 --warmup 6 --no-cpu-baseline --secondary-steps 0 > /tmp/prof_$name.log 2>&1
  cp "$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)" $out/${tag}_${name}_kernel_stats.csv
  grep '^{' /tmp/prof_$name.log | tail -1 > $out/${tag}_bench_${name}_profiled.json
}
table c2
table c3_bf16 --config c3 --encoder-dtype bf16 --decoder-dtype bf16
table c5_bf16 --config c5 --encoder-dtype bf16 --decoder-dtype bf16
# 4. decoder kernels per kernel: fp32 at configs[1] size, bf16 at configs[2] size
for cfg in f32:1 bf16:25; do
  rm -rf /tmp/prof_dec
  OBMAN_KBENCH_DEC=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /dev/null 2>&1
  cp "$(find /tmp/prof_dec -name '*kernel_stats.csv' | head -1)" $out/${tag}_dec_${cfg/:/_}_kernel_stats.csv
done
cd $GRAFT_REPO_ROOT
# 5. PMC: Chamfer traffic at 642 x 600 and at the legs' sizes, matrix-pipe busy + cycle breakdown + FETCH / WRITE of the bf16 decoder kernels
bash tools/pmc_traffic.sh 642 256 > $out/${tag}_chamfer_pmc_642.txt 2>&1
bash tools/archive/r05/pmc_pairmin_legs.sh > $out/${tag}_pmc_pairmin_legs.txt 2>&1
bash tools/pmc_mfma.sh bf16:25 > $out/${tag}_pmc_mfma_bf16_c3.txt 2>&1
bash tools/pmc_dec.sh bf16:25 > $out/${tag}_pmc_dec_bf16_c3.txt 2>&1
ls $out | grep "^${tag}_" | wc -l
