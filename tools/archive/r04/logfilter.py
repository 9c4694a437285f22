"""stdin -> stdout: keep the lines that match the pattern (argv[1]) and, at the end, the last N (argv[2]) lines whatever they are."""
import collections
import re
import sys

pat = re.compile(sys.argv[1])
ring = collections.deque(maxlen=int(sys.argv[2]))
n = 0
for line in sys.stdin:
    n += 1
    ring.append((n, line))
    if pat.search(line):
        sys.stdout.write("%d: %s" % (n, line))
sys.stdout.write("==== last %d of %d lines ====\n" % (len(ring), n))
for k, line in ring:
    sys.stdout.write("%d: %s" % (k, line))
