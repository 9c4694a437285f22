#!/bin/bash
# Which resource bounds the fp32 rows-GEMM main loop?  Subtractive builds of gemm_rows_kernel (-DOBMAN_ABLATION, template DBG):
#   1 = no operand transform   2 = no LDS staging of the next tile   4 = no global loads in the loop   8 = no fragment reads
# (combinations add up; results are wrong by design, only the time is read).
#   here:        bash tools/ablate_gemm.sh build      -> obman_train_amd/csrc/libobman_hip_ablation.so (git-ignored, travels)
#   on the box:  bash tools/ablate_gemm.sh run f32:25 -> one kbench line per variant
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/obman_train_amd/csrc
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DOBMAN_ABLATION -c $csrc/decoder.hip -o /tmp/decoder_ablation.o
  objs=$(ls $csrc/build/*.o | grep -v '/decoder.o')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $csrc/libobman_hip_ablation.so /tmp/decoder_ablation.o $objs
  ls -la $csrc/libobman_hip_ablation.so
else
  case=${2:-f32:1}
  cp $csrc/libobman_hip.so /tmp/libobman_hip.keep
  cp $csrc/libobman_hip_ablation.so $csrc/libobman_hip.so
  for v in ${OBMAN_ABLATE_SET:-0 1 2 3 4 6 8 10 14}; do
    OBMAN_GEMM_DBG=$v OBMAN_GEMM_VARIANT=$v OBMAN_KBENCH_DEC=$case python $root/tools/kbench.py decoder 2>/dev/null | grep '^{'
  done
  cp /tmp/libobman_hip.keep $csrc/libobman_hip.so
fi
