#!/bin/bash
# GPU box: epilogue ablations of the rows2 kernels (variant libraries built with -DR2_EPI_ABL=n, measurement only).
# Build the variants first, in the container (they travel with the snapshot; csrc/build/ is git-ignored):
#   cd obman_train_amd/csrc && mkdir -p build/variants && for v in 1 2 3; do
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DR2_EPI_ABL=$v -c decoder.hip -o build/variants/decoder_$v.o
#     hipcc --offload-arch=gfx950 -fPIC -shared -o build/variants/lib_$v.so $(ls build/*.o | grep -v build/decoder.o) build/variants/decoder_$v.o; done
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/obman_train_amd/csrc
cp $L/libobman_hip.so /tmp/lib_0.so
for m in 0 1 2 3; do
  if [ $m = 0 ]; then cp /tmp/lib_0.so $L/libobman_hip.so; else cp $L/build/variants/lib_$m.so $L/libobman_hip.so; fi
  rm -rf /tmp/prof_dec
  OBMAN_SKIP_BUILD=1 OBMAN_KBENCH_DEC=bf16:12 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/kb.log 2>&1
  f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $m <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rows2" in r["Name"]:
        print("EPI_ABL", sys.argv[2], r["Name"][28:62], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/lib_0.so $L/libobman_hip.so
