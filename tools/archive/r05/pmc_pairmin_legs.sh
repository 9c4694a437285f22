#!/bin/bash
# round 5: HBM-side bytes per ChamferLoss-forward launch at the secondary legs' sizes (16 050 and 64 050 predicted points x 600, bs 64),
# per kernel instance and grid size (= per direction).  Separate --pmc passes, counters only beside --kernel-trace.
cd /tmp && export TMPDIR=/tmp
for n in 16050 64050; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  OBMAN_KBENCH_ROTATE=${ROT:-4} OBMAN_KBENCH_NPRED=$n timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/kbench.py chamfer > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c $n <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2]]
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "pairmin" in k:
        name = k.split("(")[0].replace("void (anonymous namespace)::", "")
        acc[(name, r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
for (name, grid), v in sorted(acc.items()):
    print("n_pred", sys.argv[3], sys.argv[2], name, "grid", grid, "dispatches", len(v), "avg_KB", round(sum(v) / len(v), 1))
PY
done
done
