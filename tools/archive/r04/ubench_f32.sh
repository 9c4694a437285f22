#!/bin/bash
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_loop $GRAFT_REPO_ROOT/tools/ubench/mfma_f32_loop.hip 2>/dev/null && /tmp/mfma_f32_loop | tee $GRAFT_REPO_ROOT/gpurun_out/r04_mfma_f32_loop.txt
