"""Per-kernel breakdown of ONE step out of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes
NAME_results.db): the kernels between the last two launches of `marker` (default: the J-assembly kernel), their counts,
total and average durations, and the idle gaps between consecutive kernels.
  python tools/rocpd_step.py path/to/NAME_results.db [marker]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "tet_lhs_slot"
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = idx[-2], idx[-1]
seg = rows[a:b + 1]


def short(n):
    m = re.search(r"(\w+)(<[^(]*>)?\(", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:50]


d = collections.defaultdict(lambda: [0, 0])
for n, s, e in seg[:-1]:
    k = short(n)
    d[k][0] += 1
    d[k][1] += e - s
busy = sum(v[1] for v in d.values())
print("one step: wall %.3f ms, %d kernels, busy %.3f ms" % ((seg[-1][1] - seg[0][1]) / 1e6, len(seg) - 1, busy / 1e6))
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-62s %4d %8.3f ms  avg %6.1f us" % (k[:62], v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
gaps = collections.Counter()
for (n1, s1, e1), (n2, s2, e2) in zip(seg[:-1], seg[1:]):
    gaps[(short(n1)[:34], short(n2)[:34])] += max(0, s2 - e1)
print("idle between kernels: %.3f ms" % (sum(gaps.values()) / 1e6))
for k, v in gaps.most_common(8):
    print("  %8.1f us  %s -> %s" % (v / 1e3, k[0], k[1]))
