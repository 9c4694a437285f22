#!/bin/bash
# round 6, call 21: the one-direction swapped-role pair-min (contact term): parity + kbench A/B
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pairmin_gpu.py tests/test_contact_gpu.py -x -q -m gpu 2>&1 | tail -5
{ echo "== new"; timeout 300 python tools/kbench.py contactmin; echo "== OBMAN_PM_FUSED=0 (r05 path)"; OBMAN_PM_FUSED=0 timeout 300 python tools/kbench.py contactmin; } 2>/dev/null | tee $out/r06_kbench_contactmin.txt
