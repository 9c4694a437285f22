#!/bin/bash
# round 5, call 9: encoder-half evidence test, data-parallel graph tests, fallback-knob tests, fullsize per-term record
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_measured.jsonl
timeout 2400 python -m pytest tests/test_parity_evidence_gpu.py::test_configs1_gradients_through_the_encoder_bs64 tests/test_dp_graph_gpu.py tests/test_fallback_kernels_gpu.py "tests/test_fullsize_gpu.py::test_configs2_model_matches_oracle_at_full_size" tests/test_dp_two_ranks_gpu.py tests/test_dp_nccl_gpu.py tests/test_driver_gpu.py -m gpu -q --timeout 1500 --durations=8 2>&1 | tail -60 | tee $out/r05_call9_pytest.log
cp $out/parity_measured.jsonl $out/r05_parity_measured_b.jsonl 2>/dev/null
