#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=12 2>&1 | tail -40 > $out/r04_pytest_full.log
tail -25 $out/r04_pytest_full.log
