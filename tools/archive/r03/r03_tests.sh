#!/bin/bash
# GPU box: the full -m gpu suite + smoke() with the final build
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 > $out/r03_pytest_final.log; tail -6 $out/r03_pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
