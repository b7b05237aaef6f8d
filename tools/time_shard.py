"""Per-rank compute time of a sharded engine, rehearsed on ONE GPU: python3 tools/time_shard.py [world] [config]  (no all-reduce included)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
torch.cuda.set_stream(torch.cuda.Stream())      # not the legacy default stream (see bench.py)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "c4"
n_target, L, nsub, method, grid, dgrid, precision = bench.CONFIGS[cfg]
w = bench.build_workload(n_target, L, nsub, np.random.default_rng(bench.SEED))
n = len(w["q"])
pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
forces = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
balance = "--balance" in sys.argv
def measure(rank, rng=None):
    eng = bench.Engine(snb, w, method, grid, dgrid, precision, 0, rank, world, 0.1, 20, stream=torch.cuda.current_stream().cuda_stream)
    if rng is not None:
        eng.set_shard_blocks(rng[0][0], rng[0][1], rng[1])
    noise = torch.zeros_like(pos)
    def step(i):
        noise.normal_(0, 0.002); pos.add_(noise)
        eng.set_positions_device(pos.data_ptr(), False); eng.execute(False); eng.forces_to(forces.data_ptr(), False)
    for i in range(25): step(i)
    eng.sync(); torch.cuda.synchronize(); eng.reset_timers()
    t0 = time.perf_counter()
    for i in range(100): step(i)
    eng.sync(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 10
    st = eng.stats()
    d = st.sum_direct_ms / max(st.n_timed, 1)
    print("world %d rank %d%s: %.3f ms/step  (direct %.3f, recip %.3f, tiles %d)" % (world, rank, (" blocks %s/%d" % (rng[0], rng[1])) if rng else "", ms, d, st.sum_recip_ms / max(st.n_timed, 1), st.n_tiles), flush=True)
    eng.close()
    return d, max(ms - d, 0.0)

if balance:
    # the procedure of bench.py --gpus N, one rank after the other on this GPU: measure, balance, measure again (two rounds)
    ranges, period = snb.sharding.default_block_ranges(world)
    for rnd in range(3):
        times = [measure(r, (ranges[r], period)) for r in range(world)]
        print("round %d: slowest rank %.3f ms" % (rnd, max(a + b for a, b in times)), flush=True)
        ranges, period = snb.sharding.balance_block_ranges([t[0] for t in times], [t[1] for t in times])
    sys.exit(0)
for rank in sorted(set([0, world - 1])):
    eng = bench.Engine(snb, w, method, grid, dgrid, precision, 0, rank, world, 0.1, 20, stream=torch.cuda.current_stream().cuda_stream)
    noise = torch.zeros_like(pos)
    def step(i):
        noise.normal_(0, 0.002); pos.add_(noise)
        eng.set_positions_device(pos.data_ptr(), False); eng.execute(False); eng.forces_to(forces.data_ptr(), False)
    for i in range(25): step(i)
    eng.sync(); torch.cuda.synchronize(); eng.reset_timers()
    t0 = time.perf_counter()
    for i in range(100): step(i)
    eng.sync(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 10
    st = eng.stats()
    print("world %d rank %d: %.3f ms/step  (direct %.3f, recip %.3f, tiles %d)" % (world, rank, ms, st.sum_direct_ms / max(st.n_timed, 1), st.sum_recip_ms / max(st.n_timed, 1), st.n_tiles))
    eng.close()
