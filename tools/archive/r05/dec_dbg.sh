#!/bin/bash
# round 5: in-kernel phase timing (OBMAN_R2_ABL=8, -DOBMAN_ABLATION library) of the h2 and dA rows2 kernels; then tests + kernel times
# of the wide layer-4 kernels
L=$GRAFT_REPO_ROOT/obman_train_amd/csrc; out=$GRAFT_REPO_ROOT/gpurun_out
cp $L/libobman_hip.so /tmp/libobman_hip.keep; cp $L/libobman_hip_ablation.so $L/libobman_hip.so
cd /tmp && export TMPDIR=/tmp
for abl in 8 3 4 1; do
echo "== OBMAN_R2_ABL=$abl"
OBMAN_DEC_ROWS3=0 OBMAN_R2_ABL=$abl OBMAN_KBENCH_DEC=bf16:25 timeout 300 python $GRAFT_REPO_ROOT/tools/kbench.py decoder 2>&1 | grep -a "R2DBG\|kernel" | head -6
done 2>&1 | tee $out/r05_dec_dbg.txt
cp /tmp/libobman_hip.keep $L/libobman_hip.so
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py -m gpu -q --timeout 600 -x 2>&1 | tail -4
SKIP_TESTS=1 VARIANTS=1 TAG=c bash tools/r05/dec_bf16.sh 2>&1 | grep "l4\|kernel\|colsum"
