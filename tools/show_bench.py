import json,sys
d=json.load(open(sys.argv[1]))
print("value %.4g DOF/s  ms/step %.2f  setup %.1fs colors %s" % (d["value"], d["ms_per_step"], d["setup_s"], d["config"].get("colors")))
for k,v in d["kernels"].items(): print("%-9s n=%-5d avg %.4f ms  tot/step %.3f ms  %.0f GB/s  frac %.3f" % (k, v["launches"], v["avg_ms"], v["total_ms_per_step"], v["GBps"], v["frac_of_8TBps"]))
print("roofline", d["roofline"])
c=d.get("cpu_baseline")
if c: print("cpu", {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k!="sample"})
