"""[evt] lines of a development build (MCRX_EVT_DUMP=1) -> per-push timeline of the five timed stages (us from the first event)"""
import sys, collections
names = ["chan", "acq", "place", "workers", "decode"]
ev = collections.defaultdict(list)
for l in sys.stdin:
    if l.startswith("[evt]"):
        _, w, a, b = l.split(); ev[int(w)].append((float(a), float(b)))
n = min(len(v) for v in ev.values())
last = None
for i in range(max(0, n - 14), n):
    row = "push %3d " % i
    for w in range(5):
        a, b = ev[w][i]
        row += " %s %8.1f +%6.1f |" % (names[w], a, b - a)
    if last is not None: row += "  period %.1f" % (ev[1][i][0] - last)
    last = ev[1][i][0]
    print(row)
