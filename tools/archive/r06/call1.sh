#!/bin/bash
# round 6, call 1: ADVICE fixes on the GPU (watchdog wait by condition, contact backward), aten-operator census of the configs[2] bf16 step
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dp_graph_gpu.py tests/test_dp_nccl_gpu.py tests/test_contact_gpu.py tests/test_contains_binned_gpu.py -x -q -m gpu > $out/r06_call1_pytest.log 2>&1; tail -3 $out/r06_call1_pytest.log
for i in 1 2 3; do
  timeout 300 python bench.py --force-dist --graph --steps 10 --warmup 3 --no-cpu-baseline --secondary-steps 0 2>$out/r06_fd_graph_$i.err | tail -1 > $out/r06_fd_graph_$i.json
  python3 -c "
import json,sys
d=json.load(open('$out/r06_fd_graph_$i.json')); print('fd-graph', d['ms_per_step'], d['host_enqueue_ms'])"
done
CFG=c3 ENC=bf16 DEC=bf16 timeout 600 python tools/archive/r06/aten_ops.py > $out/r06_aten_ops_c3_bf16.txt 2>$out/r06_aten_ops.err; tail -40 $out/r06_aten_ops_c3_bf16.txt
