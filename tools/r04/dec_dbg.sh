#!/bin/bash
cd $GRAFT_REPO_ROOT
OBMAN_F2_DBGPRINT=1 OBMAN_KBENCH_DEC=f32:25 timeout 300 python tools/kbench.py decoder 2>&1 | grep "F2DBG" | head -8
OBMAN_F2_DBGPRINT=1 OBMAN_KBENCH_DEC=f32:1 timeout 300 python tools/kbench.py decoder 2>&1 | grep "F2DBG" | head -8
