#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_benchsize_gpu.py tests/test_bench_world2_gpu.py tests/test_dp_two_ranks_gpu.py tests/test_decoder_gpu.py -m gpu -q --timeout 900 --durations=15 2>&1 | tail -40 > $out/r04_pytest_new.log
tail -40 $out/r04_pytest_new.log
