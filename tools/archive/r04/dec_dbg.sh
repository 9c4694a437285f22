#!/bin/bash
# measurement build of the fp32 rows2f kernels with s_memtime stamps (not the product library: built into a scratch copy)
cd $GRAFT_REPO_ROOT
cp obman_train_amd/csrc/libobman_hip.so /tmp/lib_product.so
OBMAN_EXTRA_HIPCC_FLAGS=-DOBMAN_F2_TIMING python -m obman_train_amd.build --force > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
OBMAN_KBENCH_DEC=f32:25 timeout 300 python tools/kbench.py decoder 2>&1 | grep "F2DBG" | head -4
OBMAN_KBENCH_DEC=f32:1 timeout 300 python tools/kbench.py decoder 2>&1 | grep "F2DBG" | head -4
cp /tmp/lib_product.so obman_train_amd/csrc/libobman_hip.so
