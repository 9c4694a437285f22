#!/bin/bash
# GPU box: (1) the contact tests with the owner-walk backward, (2) the 1-rank RCCL graph bench command of
# tests/test_bench_world2_gpu.py eight times with its stderr kept (it aborted once at teardown in the -x run of the final tree).
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_contact_gpu.py -q -m gpu -k "contact" 2>&1 | tail -5
export OBMAN_MANO_SYNTHETIC=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
for i in 1 2 3 4 5 6 7 8; do
  MASTER_PORT=$((29600 + i)) timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --batch 8 --image-size 128 --precondition-max 9 \
    --force-dist --graph --no-cpu-baseline --secondary-steps 0 > $out/flake_$i.out 2> $out/flake_$i.err
  echo "run $i rc=$?"
done
for i in 1 2 3 4 5 6 7 8; do if ! grep -q '"value"' $out/flake_$i.out; then echo "--- run $i no line"; fi; done
grep -l "terminate\|Aborted\|what()" $out/flake_*.err | head
