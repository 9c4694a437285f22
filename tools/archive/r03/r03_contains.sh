#!/bin/bash
# GPU box: inside test after a change - parity (contact tests, whole-model goldens, full size), then the kernel's timings
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_contact_gpu.py tests/test_handnet_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 900 -x -k "not decoder" 2>&1 | tail -4
timeout 300 python tools/kbench.py contains 2>/dev/null | grep '^{'
