#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_evidence_gpu.py -q -m gpu -k "downstream" 2>&1 | grep "^E  \|passed\|failed" | cut -c1-3000 | head -12
