#!/bin/bash
# round 6, call 7: narrow / wide bf16 decoder cases with full output; pair-min with the inline-asm DPP minima; head GEMM micro-benchmark; graph robustness
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decoder_gpu.py -q -m gpu -k "bf16" 2>&1 | grep -v "^$" | tail -40 | cut -c1-1500
timeout 900 python -m pytest tests/test_pairmin_gpu.py -x -q -m gpu 2>&1 | tail -5
for v in 1; do echo "== OBMAN_PM_FUSED=$v"; OBMAN_PM_FUSED=$v timeout 300 python tools/kbench.py chamfer 2>/dev/null | grep '^{'; done | tee $out/r06_kbench_chamfer_fused_asm.txt
timeout 300 python tools/archive/r06/heads_bench.py 2>&1 | grep -v amdgpu | tee $out/r06_heads_bench.txt
timeout 1200 python -m pytest tests/test_graph_robustness_gpu.py tests/test_fallback_kernels_gpu.py -x -q -m gpu 2>&1 | tail -8
