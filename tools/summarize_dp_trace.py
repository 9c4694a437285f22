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
    print("| %s | hardware queue(s) | kernels | busy ms | RCCL kernels | example kernels |" % qcol)
    print("|---|---|---|---|---|---|")
    hwq = defaultdict(set)
    for r in rows:
        hwq[r.get(qcol, "?")].add(r.get("Queue_Id", "?"))
    rccl = []
    other = []
    for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        n_rccl = [k for k in ks if is_rccl(k[2])]
        names = sorted({k[2][:48] for k in ks})[:3]
        print("| %s | %s | %d | %.2f | %d | %s |" % (q, ",".join(sorted(hwq[q])), len(ks), sum(e - s for s, e, _ in ks) / 1e6, len(n_rccl),
                                                  "; ".join(names)))
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
    # placement inside the last full step: a step = the compute kernels between two consecutive `mano_fwd_kernel` launches
    marks = [s for s, e, n in other if "mano_fwd_kernel" in n]
    last = max(s for s, e, n in rccl)
    marks = [m for m in marks if m <= last] + [m for m in marks if m > last][:1]  # later steps (PCIe legs) run without buckets
    if len(marks) >= 3:
        t0, t1 = marks[-2], marks[-1]
        comp = [(s, e, n) for s, e, n in other if t0 <= s < t1]
        print("\nLast full step (%.2f ms between two `mano_fwd_kernel` launches): RCCL kernels by start time, with the compute "
              "kernel running at that moment\n" % ((t1 - t0) / 1e6))
        print("| start, % of step | duration us | concurrent compute kernel |")
        print("|---|---|---|")
        for s, e, n in sorted(rccl):
            if not (t0 <= s < t1):
                continue
            live = [cn for cs, ce, cn in comp if cs < e and ce > s]
            print("| %.1f | %.1f | %s |" % (100.0 * (s - t0) / (t1 - t0), (e - s) / 1e3, "; ".join(x[:60] for x in live[:3]) or "(none: gap between compute kernels)"))


if __name__ == "__main__":
    main(sys.argv[1])
