import importlib, sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
n_target, L, nsub, method, grid, dgrid, precision = bench.CONFIGS["c3"]
w = bench.build_workload(n_target, L, nsub, np.random.default_rng(bench.SEED))
n = len(w["q"])
pos = torch.tensor(w["pos"], dtype=torch.float32, device="cuda")
eng = bench.Engine(snb, w, method, grid, dgrid, precision, 0, 0, 1, 0.1, 20, stream=torch.cuda.current_stream().cuda_stream)
eng.set_positions_device(pos.data_ptr(), False); eng.execute(False); eng.sync()
st = eng.stats()
print("tiles", st.n_tiles, "masked tiles", st.n_exclusion_tiles, "blocks", st.n_blocks, "exclusions", st.n_exclusions)
