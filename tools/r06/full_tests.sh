#!/bin/bash
# all -m gpu tests + smoke + the default bench command, on the current tree
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $out/r06_full_pytest_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s); timeout 1500 python bench.py 2>$out/r06_default_bench.err | tail -1 > $out/r06_default_bench.json; t1=$(date +%s)
python3 -c "
import json; d=json.load(open('$out/r06_default_bench.json')); print('default bench wall', $t1-$t0, 's', d['value'], d['ms_per_step'], [(l.get('config'), round(l.get('value',0)), l.get('hipgraph')) for l in d['secondary']['legs']], d['cpu_baseline']['value'])"
