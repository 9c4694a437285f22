#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_stepops_gpu.py -q -m gpu -k "shadow_and_records" 2>&1 | grep -B30 "^E  " | cut -c1-300 | tail -60
