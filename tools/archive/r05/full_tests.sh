#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_measured.jsonl
timeout 3000 python -m pytest tests/ -q -m gpu --timeout 1500 --durations=12 > $out/r05_full_pytest.log 2>&1
grep -n "Fatal\|Segmentation\|passed\|failed\|FAILED\|ERROR" $out/r05_full_pytest.log | head -20
tail -25 $out/r05_full_pytest.log
cp $out/parity_measured.jsonl $out/r05_parity_measured_full.jsonl 2>/dev/null
