"""(GPU box) cost of a parameter change on c3: snb_set_particle_parameters-free path -- global parameter values change every step
(an alchemical loop), the engine re-evaluates the offset parameters and their sums on the device.  Prints ms per step with and
without the per-step change; run under rocprofv3 --kernel-trace --stats for the kernels behind the difference."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench, torch, importlib
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
n_target, Lbox, nsub, method, grid, dgrid, _ = bench.CONFIGS["c3"]
w = bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED))
n = len(w["q"])
eng = bench.Engine(snb, w, method, grid, dgrid, "single", 0, 0, 1, 0.1, -100)
L = eng.L
sel = np.nonzero(w["subset"] == 1)[0].astype(np.int32)
glob = np.zeros(len(sel), dtype=np.int32); delta = np.zeros((len(sel), 3)); delta[:, 0] = 0.01
ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int)); dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
eng.ok(L.snb_set_parameter_offsets(eng.h, 1, len(sel), ip(sel), ip(glob), dp(delta), 0, None, None, None))
pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
eng.set_positions_device(pos.data_ptr(), False)
def run(change, steps=200):
    for i in range(steps + 20):
        if i == 20:
            eng.sync(); t0 = time.perf_counter()
        if change:
            v = np.array([0.001 * (i % 7)]); eng.ok(L.snb_set_global_parameters(eng.h, 1, dp(v)))
        eng.execute(False, False)
    eng.sync(); return (time.perf_counter() - t0) / steps * 1e3
a = run(False); b = run(True)
print("ms/step fixed parameters %.4f   global parameter changed every step %.4f   (%d offset particles of %d)" % (a, b, len(sel), n))
