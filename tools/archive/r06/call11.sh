#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_driver_gpu.py tests/test_graph_robustness_gpu.py tests/test_memory_safety_gpu.py tests/test_parity_evidence_gpu.py tests/test_dp_graph_gpu.py -x -q -m gpu -k "not bs16_256 and not dec_bf16" 2>&1 | tail -6 | cut -c1-600
