#!/bin/bash
# round 6, call 13: measurement build - dA without its per-tile barrier (wrong Q sums): what does the lockstep cost?
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
SKIP_TESTS=1 KNOB=OBMAN_DEC_TN2W VARIANTS="1" TAG=da_ref bash tools/archive/r06/dec_bf16.sh | grep "EpiL1B2\|^{"
cd $GRAFT_REPO_ROOT
OBMAN_EXTRA_HIPCC_FLAGS=-DOBMAN_DA_NOBARRIER python -m obman_train_amd.build > /dev/null 2>&1
SKIP_TESTS=1 KNOB=OBMAN_DEC_TN2W VARIANTS="1" TAG=da_nobarrier bash tools/archive/r06/dec_bf16.sh | grep "EpiL1B2\|^{"
