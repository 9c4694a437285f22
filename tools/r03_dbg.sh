#!/bin/bash
# GPU box: in-kernel phase timing of the h2 rows2 kernel (OBMAN_R2_ABL=8, measurement only)
cd /tmp && export TMPDIR=/tmp
OBMAN_R2_ABL=8 OBMAN_KBENCH_DEC=${1:-bf16:25} timeout 300 python $GRAFT_REPO_ROOT/tools/kbench.py decoder 2>&1 | grep -a "R2DBG\|kernel" | head -5
