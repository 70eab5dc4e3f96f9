"""Per bench step (from one J-assembly launch to the next) the mean duration of the hot kernels, out of a rocprofv3
--kernel-trace CSV: python3 tools/trace_step_means.py <..._kernel_trace.csv>.  The --stats average of the SpMV kernel also
covers the few hundred launches of the work-space calibration (slow candidates included); this table is the one to hold
against bench.py's roofline.avg_launch_ms."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
steps = [i for i, r in enumerate(rows) if "tet_lhs_slot" in r["Kernel_Name"]]
names = {"spmv": "bcsr_spmv", "J": "tet_lhs_slot", "F": "tet_rhs_lane", "dots": "cgs_dots_stage1", "update": "cgs_update_kernel<true", "pc": "pc_apply_kernel"}
print("step   t[s]   " + "  ".join("%-14s" % k for k in names))
for si, i in enumerate(steps):
    j = steps[si + 1] if si + 1 < len(steps) else len(rows)
    seg = rows[i:j]
    out = []
    for k, pat in names.items():
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in seg if pat in r["Kernel_Name"]]
        out.append("%4d x %.4f" % (len(d), statistics.mean(d)) if d else "   0          ")
    print("%4d  %6.2f   %s" % (si, (int(seg[0]["Start_Timestamp"]) - t0) / 1e9, "  ".join("%-14s" % o for o in out)))
