#!/bin/bash
# GPU box: where do the decoder kernels spend their cycles?  Separate --pmc passes (counters only with --kernel-trace).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_dec.sh bf16:25 > gpurun_out/pmc_dec.txt'
case=${1:-bf16:25}
cd /tmp && export TMPDIR=/tmp
passes=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  OBMAN_KBENCH_DEC=$case timeout 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d /tmp/pmc_$i -- python $GRAFT_REPO_ROOT/tools/kbench.py decoder > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $i ($p): no counter file"; tail -3 /tmp/pmc_$i.log; continue; fi
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "dec::" not in k:
        continue
    acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
done
