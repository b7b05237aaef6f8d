"""(GPU box) GPU time of every step around the neighbour rebuilds: an event behind each snb_execute, elapsed time between consecutive events.
usage: python tools/step_series.py [config]   (REBUILD_EVERY=20; prints two rebuild periods, un-profiled, forces-only graph steps)"""
import sys, os, importlib
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_target, Lbox, nsub, method, grid, dgrid, precision = bench.CONFIGS[name]
w = bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED))
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
every = int(os.environ.get("REBUILD_EVERY", "20"))
eng = bench.Engine(pkg, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, every, stream=st.cuda_stream)
tdt = torch.float64 if precision == "double" else torch.float32
pos = torch.tensor(w["pos"], dtype=tdt, device="cuda"); forces = torch.zeros((len(w["q"]), 3), dtype=tdt, device="cuda")
eng.set_force_output(forces.data_ptr(), precision == "double"); eng.set_positions_device(pos.data_ptr(), precision == "double"); eng.set_timing_interval(0)
for _ in range(3 * every + 5):
    eng.execute(False)
eng.sync()
n = 3 * every
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record(st)
for i in range(n):
    eng.execute(False); ev[i + 1].record(st)
eng.sync()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("per-step GPU time (us), %d steps, a rebuild every %d:" % (n, every))
for r in range(0, n, every):
    print(" ".join("%5.0f" % (1e3 * m) for m in ms[r:r + every]))
print("mean %.1f us per step; median %.1f" % (1e3 * np.mean(ms), 1e3 * np.median(ms)))
