#!/bin/bash
# all -m gpu tests + smoke + the default bench command (with its wall time) + the LDS tile sweep, on the current tree
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
{
echo "== pytest tests -m gpu -x -q"
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== smoke()"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== python bench.py (default flags)"
t0=$(date +%s); timeout 1500 python bench.py 2>$out/r06_default_bench3.err | tail -1 > $out/r06_default_bench3.json; t1=$(date +%s)
python3 -c "
import json; d=json.load(open('$out/r06_default_bench3.json')); print('wall', $t1-$t0, 's; value', round(d['value'],1), 'img/s; ms/step', round(d['ms_per_step'],3), '; legs', [(l.get('config'), round(l.get('value',0)), 'graph' if l.get('hipgraph') else 'eager', (l.get('decoder_roofline') or {}).get('frac')) for l in d['secondary']['legs']], '; cpu', round(d['cpu_baseline']['value'],2), '; roofline', d['roofline']['frac'], d['roofline']['traffic'], '; c3 leg chamfer', [(e['kernel'][:40], round(e['avg_launch_us'],1), round(e['valu_frac'],3), e.get('traffic')) for l in d['secondary']['legs'] if l.get('roofline') for e in l['roofline']['per_launch']][:2])"
echo "== LDS tile sweep (two-sweep path)"
timeout 600 python tools/kbench.py tiles 2>/dev/null | grep '^{'
} 2>&1 | tee $out/r06_final_check3.txt
cp $out/parity_measured.jsonl $out/r06_parity_measured_full.jsonl
