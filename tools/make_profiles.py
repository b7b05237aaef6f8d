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
if os.path.exists(os.path.join(src, "pmc_mfma.txt")):
    shutil.copy(os.path.join(src, "pmc_mfma.txt"), os.path.join(dst, "%s_%s_pmc_mfma.txt" % (tag, cfg)))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
hb = json.load(open(os.path.join(root, "gpurun_out", "hb_%s_hbm.json" % tag)))
name = [k for k in hb if "k_directPacked" in k and "true, false, false, false" in k or "k_direct<double" in k][0]
rec = hb[name]
fetch, write = rec["FETCH_SIZE_per_launch_raw"], rec["WRITE_SIZE_per_launch_raw"]
out = {"command": "tools/pmc_hbm.sh hb_%s --steps 40 --warmup 5  (rocprofv3 --kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE, over bench.py)" % tag,
       "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, averaged per launch",
       "k_direct_forces": {"kernel": name.split("(")[0], "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                           "formula": "2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, KiB -> bytes",
                           "traffic_bytes_per_launch": int((2 * fetch + write) * 1024), "tiles": bench["config"]["tiles_32x32"],
                           "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes"]},
       "all_kernels_raw_KiB_per_launch": {k.split("(")[0]: v for k, v in hb.items()}}
json.dump(out, open(os.path.join(dst, "%s_%s_pmc_hbm.json" % (tag, cfg)), "w"), indent=1, sort_keys=True)
print("profiles/%s_%s_*: traffic %d B per launch vs algorithmic %d" % (tag, cfg, out["k_direct_forces"]["traffic_bytes_per_launch"], bench["roofline"]["algorithmic_bytes"]))
