#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_dp_graph_gpu.py -m gpu -q --timeout 900 2>&1 | tail -40 > $out/r05_call14_dpgraph.log
grep -n "passed\|failed\|Error" $out/r05_call14_dpgraph.log | tail -5
echo "== c3 direct"
timeout 600 python bench.py --config c3 --encoder-dtype bf16 --decoder-dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 direct %.0f img/s %.3f ms pre %d' % (d['value'], d['ms_per_step'], d['precondition_steps']))"
echo "== c3 as a stand-alone leg process"
for i in 1 2; do
timeout 600 python bench.py --leg c3:bf16:bf16:0 --secondary-steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 leg %.0f img/s %.3f ms pre %d' % (d['value'], d['ms_per_step'], d['precondition_steps']))"
done
timeout 600 python bench.py --leg c3:bf16:bf16:0 --secondary-steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 leg30 %.0f img/s %.3f ms pre %d' % (d['value'], d['ms_per_step'], d['precondition_steps']))"
