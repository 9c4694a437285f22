#!/bin/bash
# round 5, call 8: parity evidence tests (bs 64 whole models vs the host oracle), contact tests with the 1e-5 graze margin
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_measured.jsonl
timeout 2400 python -m pytest tests/test_parity_evidence_gpu.py tests/test_contact_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 1500 --durations=8 2>&1 | tail -40 | tee $out/r05_call8_pytest.log
cp $out/parity_measured.jsonl $out/r05_parity_measured_a.jsonl 2>/dev/null
free -g | head -2
