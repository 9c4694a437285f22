#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_dp_graph_gpu.py tests/test_bench_world2_gpu.py -m gpu -q --timeout 600 > $out/r05_call23_$i.log 2>&1
grep -n "passed\|failed" $out/r05_call23_$i.log | tail -2
grep -n "terminate\|what()\|Error\b" $out/r05_call23_$i.log | head -5
done
