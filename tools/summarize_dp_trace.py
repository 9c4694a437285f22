#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV of `bench.py --force-dist` -> markdown: which kernels ran on which stream/queue, and how much of
the RCCL all-reduce time overlaps compute kernels of the training stream (interval intersection on the kernel timestamps)."""
import csv
import sys
from collections import defaultdict


def is_rccl(name):
    n = name.lower()
    return ("nccl" in n or "rccl" in n or "onerankreduce" in n) and "rocclr" not in n


def main(path):
    rows = list(csv.DictReader(open(path)))
    if not rows:
        print("empty trace")
        return
    cols = rows[0].keys()
    qcol = "Stream_Id" if "Stream_Id" in cols else ("Queue_Id" if "Queue_Id" in cols else None)
    by_q = defaultdict(list)
    for r in rows:
        by_q[r.get(qcol, "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    print("# rocprofv3 --kernel-trace of `python bench.py --force-dist` (1 rank, nccl = RCCL)\n")
    print("column used to separate streams: `%s`\n" % qcol)
    print("| %s | kernels | busy ms | RCCL kernels | example kernels |" % qcol)
    print("|---|---|---|---|---|")
    rccl = []
    other = []
    for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        n_rccl = [k for k in ks if is_rccl(k[2])]
        names = sorted({k[2][:48] for k in ks})[:3]
        print("| %s | %d | %.2f | %d | %s |" % (q, len(ks), sum(e - s for s, e, _ in ks) / 1e6, len(n_rccl), "; ".join(names)))
        rccl += n_rccl
        other += [k for k in ks if not is_rccl(k[2])]
    if not rccl:
        print("\nno RCCL kernels found in the trace")
        return
    other.sort()
    tot = sum(e - s for s, e, _ in rccl)
    ov = 0
    for s, e, _ in rccl:
        for os_, oe, _ in other:
            if oe <= s:
                continue
            if os_ >= e:
                break
            ov += min(e, oe) - max(s, os_)
    names = defaultdict(lambda: [0, 0])
    for s, e, n in rccl:
        names[n[:80]][0] += 1
        names[n[:80]][1] += e - s
    print("\nRCCL kernels (a 1-rank group reduces with `oneRankReduce`, a copy-like kernel; N ranks run the ring kernels in the same "
          "place): %d launches, %.3f ms total; %.1f %% of that time overlaps a compute kernel of another stream.\n"
          % (len(rccl), tot / 1e6, 100.0 * min(ov, tot) / max(tot, 1)))
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print("* `%s` x %d, avg %.1f us" % (n, c, t / c / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
