"""(GPU box) would a neighbour rebuild that runs BESIDE the steps (own stream, results unused) cost less than one that runs between them?
Engine 1 runs forces-only c3 steps with no rebuilds; engine 2 (same workload, second stream) does nothing but rebuild every
PROBE_EVERY-th iteration (a step with neither half: gather + finish only).  Compare the per-step time of engine 1 with and without engine 2,
and with its own in-line rebuilds (tools/host_launch_cost.py rebuild).  usage: python tools/async_rebuild_probe.py [config]"""
import sys, os, time, importlib, ctypes
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_target, Lbox, nsub, method, grid, dgrid, precision = bench.CONFIGS[name]
w = bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED))
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream(priority=int(os.environ.get("PROBE_PRIO", "0")))
every = int(os.environ.get("PROBE_EVERY", "20"))
N = len(w["q"])
pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
f1 = torch.zeros((N, 3), dtype=torch.float32, device="cuda"); f2 = torch.zeros((N, 3), dtype=torch.float32, device="cuda")
eng1 = bench.Engine(pkg, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, 1 << 30, stream=s1.cuda_stream)
eng2 = bench.Engine(pkg, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, 1, stream=s2.cuda_stream)
for e, f in ((eng1, f1), (eng2, f2)):
    e.set_force_output(f.data_ptr(), False); e.set_positions_device(pos.data_ptr(), False); e.set_timing_interval(0)
def rebuild_only():
    eng2.ok(eng2.L.snb_execute(eng2.h, 1, 0, 0, 0, None))
for _ in range(50):
    eng1.execute(False)
for _ in range(3):
    rebuild_only()
eng1.sync(); eng2.sync()
for beside in (False, True, False, True):
    for rep in range(2):
        n = 300
        t0 = time.perf_counter()
        for i in range(n):
            eng1.execute(False)
            if beside and i % every == every // 2:
                rebuild_only()
        eng1.sync(); eng2.sync(); t1 = time.perf_counter()
        print("%s: %.4f ms per step of engine 1 (%s)" % (name, (t1 - t0) * 1e3 / n, "a rebuild beside it every %d steps" % every if beside else "alone, no rebuilds"))
