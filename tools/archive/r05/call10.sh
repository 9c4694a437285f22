#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rm -f $out/parity_measured.jsonl
timeout 1200 python -m pytest tests/test_dp_graph_gpu.py -m gpu -q --timeout 900 2>&1 | tail -150 > $out/r05_call10_dpgraph.log
tail -5 $out/r05_call10_dpgraph.log
timeout 2400 python -m pytest "tests/test_fullsize_gpu.py::test_configs2_model_matches_oracle_at_full_size" tests/test_dp_two_ranks_gpu.py tests/test_dp_nccl_gpu.py tests/test_driver_gpu.py tests/test_decoder_gpu.py tests/test_fallback_kernels_gpu.py -m gpu -q --timeout 1500 2>&1 | tail -40 | tee $out/r05_call10_pytest.log
cp $out/parity_measured.jsonl $out/r05_parity_measured_c.jsonl 2>/dev/null
SKIP_TESTS=1 VARIANTS=1 TAG=d bash tools/r05/dec_bf16.sh 2>&1 | grep "rows\|kernel\|tn2\|gh2"
