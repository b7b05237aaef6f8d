"""(GPU box) host time of snb_execute against the GPU time of the step: is a replayed step launch-bound?  Runs the c3 workload, N executes
back to back without synchronising (host time per call), then synchronises (total).  usage: python tools/host_launch_cost.py [config]"""
import sys, os, time, importlib
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_target, Lbox, nsub, method, grid, dgrid, precision = bench.CONFIGS[name]
w = bench.build_workload(n_target, Lbox, nsub, np.random.default_rng(bench.SEED))
torch.cuda.set_stream(torch.cuda.Stream())
eng = bench.Engine(pkg, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, 1 << 30, stream=torch.cuda.current_stream().cuda_stream)
pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda"); forces = torch.zeros((len(w["q"]), 3), dtype=torch.float32, device="cuda")
eng.set_force_output(forces.data_ptr(), False); eng.set_positions_device(pos.data_ptr(), False)
eng.set_timing_interval(0)
for _ in range(50):
    eng.execute(False)
eng.sync()
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"      # plain | move (a torch kernel per step) | deriv (derivative steps, all slices) | derivsel (the bench's slices) | rebuild (a rebuild every 20 steps) | all
walk = torch.tensor(np.random.default_rng(1).normal(0.0, 0.0015, (len(w["q"]), 3)), dtype=torch.float32, device="cuda")
sel = (np.abs(w["lam"] - 1.0).max(axis=1) > 0).astype(np.int32) if mode.endswith("sel") else np.ones(nsub * (nsub + 1) // 2, dtype=np.int32)      # (…sel: the slices bench.py asks derivatives for)
eng.set_energy_slices(sel)
if mode in ("rebuild", "all"):
    eng.close()
    eng = bench.Engine(pkg, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, int(os.environ.get("REBUILD_EVERY", "20")), stream=torch.cuda.current_stream().cuda_stream)
    eng.set_force_output(forces.data_ptr(), False); eng.set_positions_device(pos.data_ptr(), False); eng.set_timing_interval(0)
    eng.set_energy_slices(np.ones(nsub * (nsub + 1) // 2, dtype=np.int32))
    for _ in range(50):
        eng.execute(False)
    eng.sync()
for rep in range(3):
    N = 300
    t0 = time.perf_counter(); host = []
    for i in range(N):
        a = time.perf_counter()
        if mode in ("move", "all"):
            pos.add_(walk, alpha=1.0 if i % 2 else -1.0); eng.set_positions_device(pos.data_ptr(), False)
        if mode in ("deriv", "derivsel", "all"):
            eng.execute(2, fetch=False)
        else:
            eng.execute(False)
        host.append(time.perf_counter() - a)
    t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    host = np.array(host) * 1e6
    print("%s: host per execute median %.1f us, p90 %.1f, max %.1f; enqueue loop %.3f ms per step, with the final wait %.3f ms per step" % (
        os.environ.get("SNB_OVERLAP", "0") + " " + mode, np.median(host), np.quantile(host, 0.9), host.max(), (t1 - t0) * 1e3 / N, (t2 - t0) * 1e3 / N))
