#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export OBMAN_MANO_SYNTHETIC=1
run() { # name, args...
  name=$1; shift
  timeout 900 python tools/efence/efence.py "$@" > $out/r04_efence_$name.log 2>&1
  echo "$name rc=$?"; grep -v "obman-launch\|efence\] malloc\|efence\] free" $out/r04_efence_$name.log | tail -12; grep "obman-launch" $out/r04_efence_$name.log | tail -2
}
OBMAN_EFENCE_VERBOSE=2 OBMAN_EFENCE_KEEP_GOING=1 run dbg16 --config c3 --batch 16 --steps 0
OBMAN_EFENCE_VERBOSE=2 OBMAN_EFENCE_KEEP_GOING=1 run dbg64 --config c3 --batch 64 --steps 0
