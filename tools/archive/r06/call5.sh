#!/bin/bash
# round 6, call 5: fused pair-min sweep (parity, kbench A/B), watchdog wait by flight-recorder status, rows4 s_memtime stamps, step lines
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pairmin_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -8
for v in 1 0; do echo "== OBMAN_PM_FUSED=$v"; OBMAN_PM_FUSED=$v timeout 300 python tools/kbench.py chamfer 2>/dev/null | grep '^{'; done | tee $out/r06_kbench_chamfer_fused.txt
for cfg in c3 c5; do
timeout 600 python bench.py --in-process --config $cfg --encoder-dtype bf16 --decoder-dtype bf16 --steps 30 --warmup 6 --no-cpu-baseline --secondary-steps 0 2>$out/r06_${cfg}_quick.err | tail -1 > $out/r06_${cfg}_call5.json
python3 -c "
import json; d=json.load(open('$out/r06_${cfg}_call5.json')); print('$cfg bf16', d['ms_per_step'], d['value'], (d.get('decoder_roofline') or {}).get('frac'), (d.get('roofline') or {}).get('avg_launch_us'))"
done
for i in 1 2 3; do
  timeout 300 python bench.py --force-dist --graph --steps 10 --warmup 3 --no-cpu-baseline --secondary-steps 0 2>$out/r06_fd_graph_$i.err | tail -1 > $out/r06_fd_graph_$i.json
  python3 -c "
import json,sys
d=json.load(open('$out/r06_fd_graph_$i.json')); print('fd-graph', d['ms_per_step'], d['host_enqueue_ms'].get('watchdog_wait'))"
done
timeout 600 python -m pytest tests/test_dp_graph_gpu.py tests/test_dp_nccl_gpu.py tests/test_bench_world2_gpu.py -x -q -m gpu 2>&1 | tail -4
# rows4: s_memtime stamps (measurement build; the product library is restored by the next gpurun snapshot)
OBMAN_EXTRA_HIPCC_FLAGS=-DOBMAN_R4_TIMING python -m obman_train_amd.build > /dev/null 2>&1
for v in 2 5 4; do OBMAN_DEC_ROWS4=$v OBMAN_KBENCH_DEC=bf16:25 timeout 300 python tools/kbench.py decoder 2>&1 | grep "R4DBG\|^{"; done | tee $out/r06_rows4_stamps.txt
