"""Union-busy breakdown of one step out of a rocprofv3 rocpd database whose kernels run on MORE than one stream (the
partitioned solve launches its boundary rows on the halo stream): kernels that overlap in time are counted once, the idle
time is what no kernel covers.  Steps are delimited by launches of `marker`; `which` picks the step (negative = from the end),
so the two ranks of tools/probe_rank_local.py can be looked at separately.
  python tools/rocpd_union.py NAME_results.db [marker] [which ...]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "tet_lhs_slot"
which = [int(a) for a in sys.argv[3:]] or [-2]
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if marker in r[0]]


def short(n):
    m = re.search(r"(\w+)(<[^(]*>)?\(", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:50]


for w in which:
    a, b = idx[w], idx[w + 1] if w + 1 != 0 else len(rows) - 1
    seg = rows[a:b]
    t0, t1 = seg[0][1], rows[b][1]
    d = collections.defaultdict(lambda: [0, 0])
    for n, s, e, st, q in seg:
        k = (short(n)[:44], st, q)
        d[k][0] += 1
        d[k][1] += e - s
    # union of busy intervals, and who borders every idle gap
    ev = sorted(seg, key=lambda r: r[1])
    busy, gaps, cur_end, cur_name = 0, collections.defaultdict(lambda: [0, 0]), None, None
    for n, s, e, st, q in ev:
        if cur_end is None:
            cur_s, cur_end, cur_name = s, e, n
            continue
        if s > cur_end:
            busy += cur_end - cur_s
            g = gaps[(short(cur_name)[:30], short(n)[:30])]
            g[0] += 1
            g[1] += s - cur_end
            cur_s, cur_end, cur_name = s, e, n
        elif e > cur_end:
            cur_end, cur_name = e, n
    busy += cur_end - cur_s
    wall = t1 - t0
    print("step %d: wall %.1f us, union busy %.1f us, idle %.1f us, kernels %d" % (w, wall / 1e3, busy / 1e3, (wall - busy) / 1e3, len(seg)))
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:24]:
        print("   %-44s stream %-3s queue %-3s %4d %8.1f us  avg %6.1f" % (k[0], k[1], k[2], v[0], v[1] / 1e3, v[1] / v[0] / 1e3))
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:10]:
        print("   gap %8.1f us (%3d x %5.1f)  %s -> %s" % (v[1] / 1e3, v[0], v[1] / v[0] / 1e3, k[0], k[1]))
