OUT=gpurun_out/r3u; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $REPO/$OUT/trace -o f -- python3 $REPO/tools/rhs_wavetime.py 119 > $REPO/$OUT/run.txt 2>&1
cd $REPO
python3 - <<'PY'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob("gpurun_out/r3u/trace/**/*_results.db", recursive=True)[0])
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "tet_rhs_lane" in r[0]]
a = idx[-3]
t0 = None
for n, s, e in rows[a - 6:idx[-2] + 1]:
    if t0 is None: t0 = s
    m = re.search(r"(\w+)(<[^(]*>)?\(", n)
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (m.group(1) if m else n[:40])))
PY
