#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pairmin_gpu.py tests/test_contact_gpu.py tests/test_oracle_golden.py tests/test_fullsize_gpu.py -m gpu -q --timeout 600 -x -k "pairmin or chamfer or contact or golden" 2>&1 | tail -4
timeout 300 python tools/kbench.py chamfer 2>&1 | grep '^{' | tee gpurun_out/r04_kbench_chamfer.txt
