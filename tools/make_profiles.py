"""Copies the judged artefacts of a round from gpurun_out/ (scratch) into profiles/ (tracked):
    tools/final_profiles.sh  -> gpurun_out/final/{bench.json, kernel_stats.csv, pmc_sq.txt}
    tools/pmc_hbm.sh hb_rNN  -> gpurun_out/hb_rNN_hbm.json   (FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes)
usage: python tools/make_profiles.py r02 [config]"""
import json, os, shutil, sys
tag = sys.argv[1]; cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "final"); dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_%s_default_bench.json" % (tag, cfg)))
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "%s_%s_default_bench_kernel_stats.csv" % (tag, cfg)))
shutil.copy(os.path.join(src, "pmc_sq.txt"), os.path.join(dst, "%s_%s_pmc_sq.txt" % (tag, cfg)))
for extra, name in (("bench_serial.json", "serial_bench.json"), ("kernel_stats_overlap.csv", "overlapped_bench_kernel_stats.csv")):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, "%s_%s_%s" % (tag, cfg, name)))
if os.path.exists(os.path.join(src, "pmc_mfma.txt")):
    shutil.copy(os.path.join(src, "pmc_mfma.txt"), os.path.join(dst, "%s_%s_pmc_mfma.txt" % (tag, cfg)))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
hb = json.load(open(os.path.join(root, "gpurun_out", "hb_%s_hbm.json" % tag)))
def pick(deriv):
    want = "true, true, false, false" if deriv else "true, false, false, false"
    names = [k for k in hb if ("k_directPacked" in k and want in k) or "k_direct<double" in k]
    return names[0] if names else None
def record(name):
    rec = hb[name]
    fetch, write = rec["FETCH_SIZE_per_launch_raw"], rec["WRITE_SIZE_per_launch_raw"]
    return {"kernel": name.split("(")[0], "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
            "formula": "2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, KiB -> bytes",
            "traffic_bytes_per_launch": int((2 * fetch + write) * 1024), "tiles": bench["config"]["tiles_32x32"]}
out = {"command": "tools/pmc_hbm.sh hb_%s --steps 40 --warmup 5  (rocprofv3 --kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE, over bench.py)" % tag,
       "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, averaged per launch",
       "all_kernels_raw_KiB_per_launch": {k.split("(")[0]: v for k, v in hb.items()}}
for key, deriv in (("k_direct_forces", False), ("k_direct_derivatives", True)):
    name = pick(deriv)
    if name:
        out[key] = record(name)
for key in ("roofline", "roofline_other_step"):
    r = bench.get(key, {})
    tgt = "k_direct_derivatives" if "true, true" in r.get("kernel", "") else "k_direct_forces"
    if tgt in out:
        out[tgt]["algorithmic_bytes_per_launch"] = r.get("algorithmic_bytes")
json.dump(out, open(os.path.join(dst, "%s_%s_pmc_hbm.json" % (tag, cfg)), "w"), indent=1, sort_keys=True)
for key in ("k_direct_forces", "k_direct_derivatives"):
    if key in out:
        print("profiles/%s_%s_*: %s traffic %d B per launch vs algorithmic %s" % (tag, cfg, key, out[key]["traffic_bytes_per_launch"], out[key].get("algorithmic_bytes_per_launch")))
