import json,sys
r=json.load(open(sys.argv[1]))
print(sys.argv[2], r["value"], r["ms_per_step"], r["ms_per_step_forces_only"], "direct", r["config"]["direct_kernel_ms"], "recip", r["config"]["reciprocal_ms"], " | ", " ".join("%s=%.1f" % (k["kernel"].split(" ")[0][2:], k["avg_launch_us"]) for k in r["roofline_pme"]))
