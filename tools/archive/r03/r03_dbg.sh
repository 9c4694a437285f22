#!/bin/bash
# GPU box: in-kernel phase timing of the h2 rows2 kernel (OBMAN_R2_ABL=8, measurement only).  Needs the ablation library
# (bash tools/ablate_gemm.sh build):
L=$GRAFT_REPO_ROOT/obman_train_amd/csrc; cp $L/libobman_hip.so /tmp/libobman_hip.keep; cp $L/libobman_hip_ablation.so $L/libobman_hip.so
cd /tmp && export TMPDIR=/tmp
OBMAN_R2_ABL=8 OBMAN_KBENCH_DEC=${1:-bf16:25} timeout 300 python $GRAFT_REPO_ROOT/tools/kbench.py decoder 2>&1 | grep -a "R2DBG\|kernel" | head -5
cp /tmp/libobman_hip.keep $L/libobman_hip.so
