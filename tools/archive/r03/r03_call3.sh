#!/bin/bash
# GPU call 3: Chamfer S5 after the balance fix (tests + A/B + rocprof durations), ds_read_b64_tr_b16 lane-mapping probe,
# 1-rank DP overhead with the collectives skipped (what hooks + packing alone cost).
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pairmin_gpu.py tests/test_handnet_gpu.py tests/test_contact_gpu.py -m gpu -q --timeout 600 > $out/r03_pytest3.log 2>&1
echo "pytest rc=$?" >> $out/r03_pytest3.log
tail -3 $out/r03_pytest3.log
timeout 60 tools/ubench/tr_probe > $out/r03_tr_probe.txt 2>&1
head -70 $out/r03_tr_probe.txt
for s5 in 1 0; do
  OBMAN_PM_S5=$s5 OBMAN_KBENCH_NPRED=642 timeout 300 python tools/kbench.py chamfer 2>/dev/null | grep '^{' >> $out/r03c_kbench_chamfer.txt
done
cat $out/r03c_kbench_chamfer.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ch
OBMAN_KBENCH_NPRED=642 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ch -- python $GRAFT_REPO_ROOT/tools/kbench.py chamfer > /dev/null 2>&1
f=$(find /tmp/prof_ch -name "*kernel_stats.csv" | head -1)
cp "$f" $out/r03c_chamfer_kernel_stats.csv
grep -i "pairmin\|rowmean" $out/r03c_chamfer_kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
for v in plain dist noreduce; do
  case $v in plain) fl="";; dist) fl="--force-dist";; noreduce) fl="--force-dist"; export OBMAN_DP_DEBUG=noreduce;; esac
  timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $fl 2>/dev/null | tail -1 > $out/r03c_dp_${v}.json
  python -c "
import json,sys
d=json.load(open('$out/r03c_dp_${v}.json')); r=d['roofline']
print('$v', 'ms/step %.3f'%d['ms_per_step'], 'img/s %.0f'%d['value'], 'chamfer fwd us %.2f bwd %.2f'%(r['avg_launch_us'], r['backward']['avg_launch_us']), 'frac %.4f valu %.4f'%(r['frac'], r['valu']['frac']))"
done
