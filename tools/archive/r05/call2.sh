#!/bin/bash
# round 5, call 2: inside test with the wide-triangle list; where the time of the binned kernel goes (OBMAN_MC_DBG stops)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_contains_binned_gpu.py tests/test_contact_gpu.py -m gpu -q --timeout 600 -x 2>&1 | tail -3 | tee $out/r05_call2_pytest.log
for d in 0 1 2 3 4; do
  echo "== OBMAN_MC_DBG=$d"
  OBMAN_MC_DBG=$d timeout 300 python tools/kbench.py contains 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('  %-5s F=%-6d p=%-2d %8.1f us (all pairs %7.1f) same=%s' % (r['geometry'], r['F'], r['patches'], r['us'], r['all_pairs_us'], r['identical']))
"
done 2>&1 | tee $out/r05_contains_phases.txt
