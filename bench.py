#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SlicedNonbondedForce hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3|c2|c4|c5|small]

A "step" is ONE force evaluation (direct-space sliced pair loop + exclusion/1-4 lists + sliced PME reciprocal
pipeline) of the synthetic periodic box named by BASELINE.json (default c3: 300k atoms, 4 subsets with lambda
scaling, PME 120^3, single precision), with positions already resident in HBM.  metric = ns/day at dt = 2 fs
(force evaluation only: no integrator, no bonded forces -- OpenMM is not available), SURVEY.md section 8(d).

N > 1 (launched by torch.distributed.run): ONE system, strong scaling -- PME subset grids and direct-space
work items are sharded over the ranks (snb_config.shard_rank/shard_count) and the per-rank partial forces are
summed with one RCCL all-reduce per step.

Extra objects on the JSON line: "roofline" for the dominant kernel (the direct-space tile kernel, timed with HIP
events on the engine's own stream; algorithmic bytes 52*N + 1792*T) and "cpu_baseline" (the CPU
oracle timed on a bounded, down-scaled sample of the same workload, rank 0 / N=1 only).
"""
import argparse
import ctypes
import importlib
import json
import math
import os
import sys
import time

# the GPU box reports every host core but grants a 16-core share: keep the CPU-baseline leg's OpenMP team inside it
if "OMP_NUM_THREADS" not in os.environ:
    os.environ["OMP_NUM_THREADS"] = str(min(16, len(os.sched_getaffinity(0))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = 20251212
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); 6.29 TB/s is the measured copy ceiling


# --------------------------------------------------------------------------------------------------------------------
# Synthetic workloads: cubic box, density 100 atoms/nm^3, cutoff 1.0 nm, alpha 2.6283/nm.
# --------------------------------------------------------------------------------------------------------------------
def build_workload(n_target, L, nsub, rng, blob_atoms=(12000, 3000, 300)):
    """Bulk 3-site water-like molecules (subset 0, or radial shells when nsub > 4) + up to three spherical solute blobs
    of chain-bonded charged LJ sites (1-2/1-3 excluded, 1-4 scaled 0.8333/0.5)."""
    scale = n_target / 300000.0
    blob_atoms = [max(40, int(round(b * scale))) for b in blob_atoms]
    centres = np.array([[0.30, 0.30, 0.35], [0.70, 0.62, 0.55], [0.45, 0.80, 0.20]]) * L
    nblobs = min(3, max(0, nsub - 1)) if nsub <= 4 else 3
    pos, q, sig, eps, sub, exc = [], [], [], [], [], []
    radii = []
    a = 0.2154
    for b in range(nblobs):
        nb = blob_atoms[b]
        r = (3.0 * nb / (4.0 * math.pi * 100.0)) ** (1.0 / 3.0) * 1.02
        m = int(math.ceil(2 * r / a)) + 1
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3) * a - r
        g = g[np.argsort(np.linalg.norm(g, axis=1), kind="stable")][:nb]
        g = g[np.lexsort((g[:, 2], g[:, 1], g[:, 0]))]          # chain runs along z, then y, then x
        radii.append(np.linalg.norm(g, axis=1).max() + 0.25)
        base = sum(len(p) for p in pos)
        pos.append(g + centres[b] + rng.uniform(-0.03, 0.03, g.shape))
        qq = rng.uniform(0.1, 0.5, nb) * np.where(np.arange(nb) % 2 == 0, 1.0, -1.0); qq -= qq.mean()
        q.append(qq); sig.append(rng.uniform(0.16, 0.19, nb)); eps.append(rng.uniform(0.2, 0.8, nb))
        sub.append(np.full(nb, (nsub - nblobs + b) if nsub > 4 else (b + 1), dtype=np.int32))
        for i in range(nb - 1):
            exc.append((base + i, base + i + 1, 0.0, 1.0, 0.0))
        for i in range(nb - 2):
            exc.append((base + i, base + i + 2, 0.0, 1.0, 0.0))
        for i in range(nb - 3):
            s14 = 0.5 * (sig[-1][i] + sig[-1][i + 3]); e14 = 0.5 * math.sqrt(eps[-1][i] * eps[-1][i + 3])
            exc.append((base + i, base + i + 3, 0.8333 * qq[i] * qq[i + 3], s14, e14))
    n_solute = sum(len(p) for p in pos)
    n_mol = (n_target - n_solute) // 3
    # molecular lattice, sites inside the blobs removed
    spacing = (L ** 3 / (n_mol * 1.08 + sum(4.19 * r ** 3 for r in radii) * 33.4)) ** (1.0 / 3.0)
    while True:
        m = int(L / spacing)
        g = (np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3) + 0.5) * (L / m)
        keep = np.ones(len(g), dtype=bool)
        for b in range(nblobs):
            d = g - centres[b]; d -= L * np.round(d / L)
            keep &= np.linalg.norm(d, axis=1) > radii[b]
        g = g[keep]
        if len(g) >= n_mol:
            break
        spacing *= 0.99
    g = g[np.sort(rng.choice(len(g), n_mol, replace=False))]
    g = g + rng.uniform(-0.02, 0.02, g.shape)
    # random orientations, rigid geometry r_OH = 0.09572 nm, HOH = 104.52 deg
    u = rng.standard_normal((n_mol, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.standard_normal((n_mol, 3)); v -= (v * u).sum(1, keepdims=True) * u; v /= np.linalg.norm(v, axis=1, keepdims=True)
    half = math.radians(104.52 / 2)
    h1 = g + 0.09572 * (math.cos(half) * u + math.sin(half) * v)
    h2 = g + 0.09572 * (math.cos(half) * u - math.sin(half) * v)
    w = np.stack([g, h1, h2], 1).reshape(-1, 3)
    base = n_solute
    pos.append(w)
    q.append(np.tile([-0.834, 0.417, 0.417], n_mol)); sig.append(np.tile([0.315075, 1.0, 1.0], n_mol)); eps.append(np.tile([0.635968, 0.0, 0.0], n_mol))
    if nsub > 4:   # c4: bulk split into (nsub-3) radial shells around the box centre
        nshell = nsub - nblobs
        d = g - 0.5 * L
        rr = np.linalg.norm(d, axis=1)
        edges = np.quantile(rr, np.linspace(0, 1, nshell + 1)[1:-1])
        shell = np.searchsorted(edges, rr).astype(np.int32)
        sub.append(np.repeat(shell, 3))
    elif nsub == 2 and nblobs == 1:
        sub.append(np.zeros(3 * n_mol, dtype=np.int32))
    else:
        sub.append(np.zeros(3 * n_mol, dtype=np.int32))
    for k in range(n_mol):
        o = base + 3 * k
        exc.append((o, o + 1, 0.0, 1.0, 0.0)); exc.append((o, o + 2, 0.0, 1.0, 0.0)); exc.append((o + 1, o + 2, 0.0, 1.0, 0.0))
    pos = np.concatenate(pos); q = np.concatenate(q); sig = np.concatenate(sig); eps = np.concatenate(eps); sub = np.concatenate(sub)
    q -= q.mean()
    exc = np.array(exc, dtype=np.float64).reshape(-1, 5)
    S = nsub * (nsub + 1) // 2
    lam = np.ones((S, 2))
    vals = [0.7, 0.9, 0.5, 1.0, 0.3, 0.6, 0.8]
    sl = lambda i, j: (i * (i + 1) // 2 + j) if i > j else (j * (j + 1) // 2 + i)
    if nsub <= 4:
        k = 0
        for s in range(1, nsub):
            lam[sl(0, s), 0] = vals[k % 7]; lam[sl(0, s), 1] = vals[(k + 1) % 7]; k += 2
        if nsub >= 3:
            lam[sl(1, 2), 0] = vals[6]
    else:
        for b in range(nblobs):
            s_b = nsub - nblobs + b
            for s in range(nsub - nblobs):
                lam[sl(s, s_b), 0] = vals[(2 * b) % 7]; lam[sl(s, s_b), 1] = vals[(2 * b + 1) % 7]
    return dict(pos=np.ascontiguousarray(pos), q=q, sigma=sig, epsilon=eps, subset=np.ascontiguousarray(sub, dtype=np.int32),
                exc_pairs=np.ascontiguousarray(exc[:, :2].astype(np.int32)), exc_qq=np.ascontiguousarray(exc[:, 2]),
                exc_sigma=np.ascontiguousarray(exc[:, 3]), exc_eps=np.ascontiguousarray(exc[:, 4]), lam=lam, L=L, nsub=nsub)


CONFIGS = {
    # name: (atoms, L, subsets, method, grid, dgrid, precision)
    "small": (24000, 6.2145, 4, 4, 54, 0, "single"),
    "c2": (96000, 9.865, 2, 4, 80, 0, "single"),
    "c3": (300000, 14.42, 4, 4, 120, 0, "single"),
    "c4": (300000, 14.42, 8, 4, 120, 0, "single"),
    "c5": (1000000, 21.54, 4, 5, 180, 90, "double"),
    "c3t": (300000, 14.42, 4, 4, 120, 0, "single"),      # c3 in a triclinic cell (shear_workload)
    "c3l": (300000, 14.42, 4, 5, 120, 60, "single"),     # c3 with LJPME (dispersion mesh 60^3)
}
ALPHA = 2.6283
CUTOFF = 1.0
# configs whose BASELINE.json line names energy-parameter derivatives ("300k-atom solvated protein, 4 subsets with lambda_elec/lambda_vdW derivatives"):
# their headline step is the derivative-carrying one
DERIVATIVE_CONFIGS = ("c3", "c3t", "c3l")


def workload_box(w):
    """Box vectors (rows a, b, c, flattened) of a workload: cubic, or the sheared cell of the triclinic variants."""
    return np.ascontiguousarray(w.get("box", np.diag([w["L"]] * 3)), dtype=np.float64).reshape(9)


def shear_workload(w):
    """Triclinic variant of a cubic workload: the same atoms in a cell of the same volume with OpenMM-reduced box vectors
    a = (L,0,0), b = (L/3,L,0), c = (-L/4,L/4,L); positions are sheared with the cell (a synthetic perf/parity workload)."""
    L = w["L"]
    H = np.array([[L, 0.0, 0.0], [L / 3.0, L, 0.0], [-L / 4.0, L / 4.0, L]])
    w = dict(w)
    w["pos"] = np.ascontiguousarray((w["pos"] / L) @ H)
    w["box"] = H
    return w


class ForceView:
    """Duck-typed SlicedNonbondedForce over the workload arrays, for the oracle front-end (cpu_baseline / checks)."""

    def __init__(self, w, method, grid, dgrid):
        self.w, self.method, self.grid, self.dgrid = w, method, grid, dgrid
    def getNumParticles(self): return len(self.w["q"])
    def getNumSubsets(self): return self.w["nsub"]
    def getParticleParameters(self, i): return (self.w["q"][i], self.w["sigma"][i], self.w["epsilon"][i])
    def getParticleSubset(self, i): return int(self.w["subset"][i])
    def getNumParticleParameterOffsets(self): return 0
    def getNumExceptionParameterOffsets(self): return 0
    def getNumExceptions(self): return len(self.w["exc_qq"])
    def getExceptionParameters(self, k): return (int(self.w["exc_pairs"][k, 0]), int(self.w["exc_pairs"][k, 1]), self.w["exc_qq"][k], self.w["exc_sigma"][k], self.w["exc_eps"][k])
    def getNumScalingParameters(self): return 0
    def getNumEnergyParameterDerivatives(self): return 0
    def getNumGlobalParameters(self): return 0
    def getNonbondedMethod(self): return self.method
    def getCutoffDistance(self): return CUTOFF
    def getUseSwitchingFunction(self): return False
    def getSwitchingDistance(self): return -1.0
    def getReactionFieldDielectric(self): return 78.3
    def getPMEParameters(self): return (ALPHA, self.grid, self.grid, self.grid)
    def getLJPMEParameters(self): return (ALPHA, self.dgrid, self.dgrid, self.dgrid)
    def getExceptionsUsePeriodicBoundaryConditions(self): return False
    def getUseDispersionCorrection(self): return False
    def getIncludeDirectSpace(self): return True


def oracle_eval(w, method, grid, dgrid, include_direct=1, include_reciprocal=1):
    import oracle
    fv = ForceView(w, method, grid, dgrid)
    # resolve() would loop in Python over every atom; feed the C entry point directly instead
    L = oracle.lib()
    cfg = oracle.OrcConfig()
    n = len(w["q"])
    cfg.n_atoms = n; cfg.n_subsets = w["nsub"]; cfg.method = method; cfg.cutoff = CUTOFF; cfg.rf_dielectric = 78.3
    cfg.alpha = ALPHA; cfg.grid[0] = cfg.grid[1] = cfg.grid[2] = grid
    cfg.alpha_d = ALPHA; cfg.dgrid[0] = cfg.dgrid[1] = cfg.dgrid[2] = max(dgrid, 1)
    cfg.include_direct = int(include_direct); cfg.include_reciprocal = int(include_reciprocal); cfg.background_term = 1; cfg.correct_q1 = 1
    S = w["nsub"] * (w["nsub"] + 1) // 2
    forces = np.zeros((n, 3)); sliceE = np.zeros((S, 2))
    box = workload_box(w)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    t0 = time.perf_counter()
    rc = L.orc_evaluate(ctypes.byref(cfg), dp(w["pos"]), dp(box), dp(w["q"]), dp(w["sigma"]), dp(w["epsilon"]), ip(w["subset"]), len(w["exc_qq"]),
                        ip(w["exc_pairs"]), dp(w["exc_qq"]), dp(w["exc_sigma"]), dp(w["exc_eps"]), dp(np.ascontiguousarray(w["lam"])), None, dp(forces), dp(sliceE))
    dt = time.perf_counter() - t0
    assert rc == 0
    return forces, sliceE, dt, int(L.orc_last_pair_count())


class Engine:
    """Thin ctypes driver of the C ABI for the bench (device-resident positions, torch only for memory/streams)."""

    def __init__(self, pkg, w, method, grid, dgrid, precision, device, rank, world, padding, rebuild_interval, stream=None):
        self.capi = pkg.capi; self.L = pkg.capi.lib(); self.h = ctypes.c_void_p()
        cfg = self.capi.SnbConfig()
        cfg.abi_version = self.capi.SNB_ABI_VERSION; cfg.n_atoms = len(w["q"]); cfg.n_subsets = w["nsub"]; cfg.method = method
        cfg.precision = {"single": 0, "double": 1, "mixed": 2}[precision]; cfg.device = device; cfg.cutoff = CUTOFF; cfg.rf_dielectric = 78.3
        cfg.alpha = ALPHA; cfg.grid[0] = cfg.grid[1] = cfg.grid[2] = grid
        cfg.alpha_d = ALPHA; cfg.dgrid[0] = cfg.dgrid[1] = cfg.dgrid[2] = max(dgrid, 1)
        cfg.neighbor_padding = padding; cfg.rebuild_interval = rebuild_interval
        cfg.shard_rank = rank; cfg.shard_count = world
        cfg.stream = stream
        self.ok(self.L.snb_create(ctypes.byref(cfg), ctypes.byref(self.h)), create=True)
        dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        self.ok(self.L.snb_set_particles(self.h, dp(w["q"]), dp(w["sigma"]), dp(w["epsilon"]), ip(w["subset"])))
        self.ok(self.L.snb_set_exceptions(self.h, len(w["exc_qq"]), ip(w["exc_pairs"]), dp(w["exc_qq"]), dp(w["exc_sigma"]), dp(w["exc_eps"]), None))
        self.ok(self.L.snb_set_lambdas(self.h, dp(np.ascontiguousarray(w["lam"]))))
        box = workload_box(w)
        self.ok(self.L.snb_set_box(self.h, dp(box)))

    def close(self):
        if self.h:
            self.L.snb_destroy(self.h); self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ok(self, st, create=False):
        if st != 0:
            raise RuntimeError("snb error %d: %s" % (st, (self.L.snb_last_error(None if create else self.h) or b"").decode()))

    def set_positions_device(self, ptr, is_double):
        self.ok(self.L.snb_set_positions(self.h, ctypes.c_void_p(ptr), 1, int(is_double), 0))

    def execute(self, energy=False, fetch=True):
        """One evaluation.  energy: accumulate the raw per-slice energies too (the step of a force with energy-parameter derivatives);
        fetch=False leaves them on the device (no synchronisation: snb_get_slice_energies reads them when wanted)."""
        if energy and not fetch:
            self.ok(self.L.snb_execute(self.h, 1, 2 if energy == 2 else 1, 1, 1, None))      # 2: derivative-only step (slices of set_energy_slices)
            return None
        e = ctypes.c_double(0.0)
        self.ok(self.L.snb_execute(self.h, 1, int(energy), 1, 1, ctypes.byref(e)))
        return e.value

    def set_energy_slices(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.int32)
        self.ok(self.L.snb_set_energy_slices(self.h, m.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))

    def set_timing_interval(self, n):
        self.ok(self.L.snb_set_timing_interval(self.h, int(n)))

    def set_shard_blocks(self, begin, end, period):
        self.ok(self.L.snb_set_shard_blocks(self.h, int(begin), int(end), int(period)))

    def set_force_output(self, ptr, is_double, accumulate=0):
        self.ok(self.L.snb_set_force_output(self.h, ctypes.c_void_p(ptr), int(is_double), int(accumulate)))

    def forces_to(self, ptr, is_double):
        self.ok(self.L.snb_get_forces(self.h, ctypes.c_void_p(ptr), 1, int(is_double), 0))

    def slice_energies(self, S):
        out = np.zeros((S, 2)); self.ok(self.L.snb_get_slice_energies(self.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))); return out

    def stats(self):
        st = self.capi.SnbStats(); self.ok(self.L.snb_get_stats(self.h, ctypes.byref(st))); return st

    def reset_timers(self): self.ok(self.L.snb_reset_timers(self.h))
    def rebuild(self): self.ok(self.L.snb_rebuild_neighbors(self.h))
    def sync(self): self.ok(self.L.snb_synchronize(self.h))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None)
    ap.add_argument("--padding", type=float, default=0.1, help="neighbour-list skin in nm")
    ap.add_argument("--rebuild-interval", type=int, default=20, help="re-sort atoms and rebuild the tile lists every this many steps (inside the timed region)")
    ap.add_argument("--precision", default=None, choices=["single", "mixed", "double"], help="override the config's precision (mixed: single-precision arithmetic, 64-bit fixed-point force accumulation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-auto-leg", action="store_true", help="skip the leg with displacement-triggered rebuilds (the `displacement_triggered_rebuilds` object of the JSON line)")
    ap.add_argument("--no-balance", action="store_true", help="N > 1: keep the even i-block split instead of balancing direct-space work against the ranks' reciprocal work")
    ap.add_argument("--check", action="store_true", help="also compare forces/energies with the CPU oracle (slow at full size)")
    ap.add_argument("--no-double", action="store_true", help="skip the double-precision leg (the `double_precision` object of the JSON line)")
    args = ap.parse_args()

    # --gpus N > 1 without a launcher: start the N ranks ourselves (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1)
    # BEFORE anything in this process touches the GPU, relay rank 0's JSON line and exit with the launcher's code.  Never fall through to a
    # one-GPU run that would print n_gpus: 1, never re-exec a process that has initialised HIP.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d does not match --gpus %d (launch with --nproc-per-node %d, or drop the launcher and let bench.py start the ranks)" % (world, args.gpus, args.gpus))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SNB_BENCH_REHEARSE=1: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (ranks share devices, gloo carries the
        # collectives through the host) -- exercises the code path, its numbers mean nothing.  The measured run is RCCL, one rank per GPU.
        if os.environ.get("SNB_BENCH_REHEARSE"):
            local = local % torch.cuda.device_count()
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    # a stream of our own for the harness kernels and the engine, as any application has: on the legacy default stream every transition
    # between a harness kernel and the engine's work costs ~17 us of idle GPU (measured on c3: 0.508 vs 0.483 ms per step)
    if not os.environ.get("SNB_BENCH_NULL_STREAM"):
        torch.cuda.set_stream(torch.cuda.Stream())
    dev = torch.device("cuda", local)
    # N = 1: the headline workload c3 (BASELINE.json config 3).  N > 1: BASELINE.json config 4 -- the same 300k-atom box cut into 8 subsets,
    # whose 8 subset grids shard over the ranks (strong scaling; `one_gpu_same_workload` is the unsharded engine on the same workload and step kind, measured by rank 0 in the same run)
    cfg_name = args.config or ("c3" if world == 1 else "c4")
    n_target, Lbox, nsub, method, grid, dgrid, precision = CONFIGS[cfg_name]
    precision = args.precision or precision
    pkg = importlib.import_module("openmm-nonbonded-slicing_amd")
    pkg.capi.build()
    w = build_workload(n_target, Lbox, nsub, np.random.default_rng(SEED))
    if cfg_name == "c3t":
        w = shear_workload(w)
    N = len(w["q"])
    is_double = precision == "double"
    # the engine enqueues on torch's current stream: one stream, no host synchronisation inside the timed loop
    eng = Engine(pkg, w, method, grid, dgrid, precision, local, rank, world, args.padding, args.rebuild_interval, stream=torch.cuda.current_stream().cuda_stream)
    tdtype = torch.float64 if is_double else torch.float32
    pos0 = torch.tensor(w["pos"], dtype=tdtype, device=dev).contiguous()
    pos = pos0.clone()
    forces = torch.zeros((N, 3), dtype=tdtype, device=dev)
    eng.set_force_output(forces.data_ptr(), is_double)      # the step graph ends with the user-order force write; forces_to() below is then free
    # The coordinates do a RANDOM WALK (one in-place update kernel per step, standing where an integrator would): sixteen fixed
    # Gaussian displacement fields of sigma = 0.0015 nm are added with a pseudo-random sign each, so the displacement since the last
    # rebuild grows as sqrt(steps) -- 0.007 nm rms per coordinate after the 20 steps of a list's life, far inside skin/2 = 0.05 nm -- and the list
    # really ages: atoms drift across tile, column and mesh-cell boundaries between rebuilds (snb_stats.n_list_overruns must stay 0).
    walk_rng = np.random.default_rng(SEED + 1)
    walk = [torch.tensor(walk_rng.normal(0.0, 0.0015, (N, 3)), dtype=tdtype, device=dev) for _ in range(16)]
    walk_sign = walk_rng.choice([-1.0, 1.0], size=1 << 16)

    def move(i):
        pos.add_(walk[i % 16], alpha=float(walk_sign[i % len(walk_sign)]))

    def fenced_step(i, derivatives=False):
        move(i)
        eng.set_positions_device(pos.data_ptr(), is_double)
        eng.execute(2, fetch=False) if derivatives else eng.execute(False)
        eng.forces_to(forces.data_ptr(), is_double)
        if world > 1:
            dist.all_reduce(forces)          # RCCL, ordered after the engine's kernels on the same stream

    t_rebuild0 = time.perf_counter()
    fenced_step(0); eng.sync()
    first_ms = (time.perf_counter() - t_rebuild0) * 1e3
    block_ranges = None
    if world > 1 and not args.no_balance:
        # untimed, before the warm-up: two rounds of load balancing.  Each rank times its own compute (no all-reduce), the times are
        # all-gathered, and every rank derives the same uneven i-block ranges (sharding.balance_block_ranges): ranks that carry a PME
        # grid hand direct-space blocks to ranks that do not.
        for _ in range(2):
            eng.sync(); torch.cuda.synchronize(); eng.reset_timers()
            tb = time.perf_counter()
            for i in range(48):
                move(i)
                eng.set_positions_device(pos.data_ptr(), is_double); eng.execute(False)
            eng.sync(); torch.cuda.synchronize()
            ms = (time.perf_counter() - tb) * 1e3 / 48
            stb = eng.stats()
            d = stb.sum_direct_ms / max(stb.n_timed, 1)
            mine = torch.tensor([d, max(ms - d, 0.0)], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            times = [[float(x) for x in t.tolist()] for t in every]
            block_ranges, period = pkg.sharding.balance_block_ranges([t[0] for t in times], [t[1] for t in times])
            eng.set_shard_blocks(block_ranges[rank][0], block_ranges[rank][1], period)
    # Untimed pre-conditioning before the W warm-up steps: a fresh process starts with the GPU at idle clocks (kernel stamps of the first
    # ~100 ms read 15 % long: 0.239 ms for the pair kernel against 0.204 sustained) and with list buffers that still grow at the first few
    # rebuilds; 250 steps (13 rebuilds, ~0.12 s) bring both to their steady state.  Then W warm-up steps, then exactly K timed steps.
    # (250 steps before the timed region, not 240: with a rebuild every 20 steps the region would otherwise START with a rebuild, issued into
    # a GPU the synchronisation in front of the region has just drained -- a host-bound millisecond at idle clocks that a long run pays once
    # and a 20-step region pays in full: measured 1.7 ms per region on c4, 0.2 ms on c3.  Now the region's rebuilds fall mid-region, with the
    # queue full, as every rebuild of a long run does; their number per 20 steps is unchanged.)
    precondition = max(0, 250 - args.warmup) if not os.environ.get("SNB_BENCH_NO_PRECONDITION") else 0
    for i in range(precondition):
        fenced_step(1 << 20 | i)
    for i in range(1, args.warmup + 1):
        fenced_step(i)
    eng.sync(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.set_timing_interval(int(os.environ.get("SNB_BENCH_TIMING_INTERVAL", max(1, min(32, (args.steps + 2) // 3)))))      # >= 3 timed (eager, serial) steps with kernel stamps even in a short region (an eager stamped step costs ~85 us more than a replayed, overlapped one)
    # derivatives are requested for the scaling parameters of the workload: the slices whose lambda differs from 1
    deriv_slices = (np.abs(w["lam"] - 1.0).max(axis=1) > 0).astype(np.int32)
    eng.set_energy_slices(deriv_slices)

    def timed_region(first, derivatives):
        """EXACTLY K steps bracketed by barrier + synchronize on both sides; MAX over ranks.  Returns (ms per step, engine stats of the region)."""
        eng.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        eng.reset_timers()
        before = int(eng.stats().n_rebuilds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            fenced_step(first + i, derivatives=derivatives)
        eng.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        stx = eng.stats()
        return elapsed * 1e3 / args.steps, stx, int(stx.n_rebuilds) - before

    # Two regions of K steps each, same walk, same rebuild cadence:
    #  * forces only -- snb_execute(include_energy = 0);
    #  * the step BASELINE.json config 3 names, "with lambda_elec / lambda_vdW derivatives": a force that requests energy-parameter derivatives
    #    accumulates the raw per-slice energies on EVERY step (the reference does, CommonNonbondedSlicingKernels.cpp:712-718); the sums stay on
    #    the device (read once at the end, outside the region).
    # `value` is the region the config names (c3 family: derivatives; the others: forces only); the other one is reported beside it.
    plain_ms, st, rebuilds_plain = timed_region(args.warmup + 1, False)
    for i in range(4):
        fenced_step(args.warmup + args.steps + 1 + i, derivatives=True)
    deriv_ms, st_d, rebuilds_deriv = timed_region(args.warmup + args.steps + 5, True)
    headline_deriv = cfg_name in DERIVATIVE_CONFIGS
    ms_per_step = deriv_ms if headline_deriv else plain_ms
    st_head = st_d if headline_deriv else st
    rebuilds_in_region = rebuilds_deriv if headline_deriv else rebuilds_plain
    resident_ms = None
    allreduce_ms = None
    if world > 1:
        # the exchange alone: K all-reduces of the force array back to back (its share of ms_per_step, for the record)
        torch.cuda.synchronize(); dist.barrier()
        ta = time.perf_counter()
        for _ in range(args.steps):
            dist.all_reduce(forces)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - ta) * 1e3 / args.steps
    overruns = int(eng.stats().n_list_overruns)
    one_gpu_same = None; world_facts = None
    if world > 1:
        # The curve's own anchor, measured in THIS run: rank 0 runs the UNSHARDED engine of the same workload, the same
        # step kind (the one `value` quotes), walk and rebuild cadence for K steps on its GPU while the other ranks wait at the barrier.
        if rank == 0:
            e1 = Engine(pkg, w, method, grid, dgrid, precision, local, 0, 1, args.padding, args.rebuild_interval, stream=torch.cuda.current_stream().cuda_stream)
            p1 = pos0.clone(); f1 = torch.zeros((N, 3), dtype=tdtype, device=dev)
            e1.set_force_output(f1.data_ptr(), is_double); e1.set_energy_slices(deriv_slices)

            def step1(i):
                p1.add_(walk[i % 16], alpha=float(walk_sign[i % len(walk_sign)]))
                e1.set_positions_device(p1.data_ptr(), is_double)
                e1.execute(2, fetch=False) if headline_deriv else e1.execute(False)
                e1.forces_to(f1.data_ptr(), is_double)
            for i in range(90):      # (not a multiple of the rebuild interval: the region's rebuilds fall mid-region, as in the main one)
                step1(i)
            e1.sync(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step1(90 + i)
            e1.sync(); torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t1) * 1e3 / args.steps
            e1.close()
            one_gpu_same = {"value": round(86.4 * 2.0 / ms1, 3), "unit": "ns/day", "ms_per_step": round(ms1, 4), "steps": args.steps, "measured_in_this_run": True,
                            "workload": cfg_name, "step": "with derivatives" if headline_deriv else "forces only", "device": torch.cuda.get_device_name(local)}
        torch.cuda.synchronize(); dist.barrier()
        # did the collective backend see N ranks on N devices?  Answerable from the line itself.
        mine = {"rank": rank, "local_rank": local, "device_index": int(torch.cuda.current_device()), "device_name": torch.cuda.get_device_name(local),
                "pci_bus_id": getattr(torch.cuda.get_device_properties(local), "pci_bus_id", None), "host": os.uname().nodename}
        every = [None] * world
        dist.all_gather_object(every, mine)
        world_facts = {"world_size": int(dist.get_world_size()), "backend": str(dist.get_backend()), "ranks": every,
                       "distinct_devices": len({(r_["host"], r_["device_index"]) for r_ in every})}
    if world == 1:
        # For the record (never `value`): the same evaluations fed from two coordinate sets generated beforehand, i.e. nothing but
        # force evaluations on the stream.  The coordinate-update kernel of the main region stands where an integrator would.
        ring = [pos0 + 0.5 * walk[j] for j in (1, 2)]
        n_res = max(20, min(args.steps, 100))
        eng.rebuild()      # (the walk has carried the atoms away from pos0: fresh list for these coordinates)
        for i in range(4):
            eng.set_positions_device(ring[i % 2].data_ptr(), is_double); eng.execute(False); eng.forces_to(forces.data_ptr(), is_double)
        eng.sync(); torch.cuda.synchronize()
        tr = time.perf_counter()
        for i in range(n_res):
            eng.set_positions_device(ring[i % 2].data_ptr(), is_double); eng.execute(False); eng.forces_to(forces.data_ptr(), is_double)
        eng.sync(); torch.cuda.synchronize()
        resident_ms = (time.perf_counter() - tr) * 1e3 / n_res
    ns_day = 86.4 * 2.0 / ms_per_step
    T = int(st_head.n_tiles)
    itemsize = 8 if is_double else 4
    G = grid ** 3; Gh = grid * grid * (grid // 2 + 1)

    def valu_issue(kern, tiles, launch_ms):
        """The bound that applies to the pair kernel: SQ_INSTS_VALU per launch from the newest committed --pmc pass
        (profiles/rNN_<config>_pmc_sq.txt, taken at the tile count of profiles/rNN_<config>_default_bench.json) scaled to this run's tile
        count, x 4 cycles per wave instruction (packed fp32: 64 lanes x 2 on a 32-lane-pair VALU), over SIMDs x clock x the launch time."""
        here = os.path.dirname(os.path.abspath(__file__))
        for tag in ("r04", "r03"):
            fsq = os.path.join(here, "profiles", "%s_%s_pmc_sq.txt" % (tag, cfg_name)); fb = os.path.join(here, "profiles", "%s_%s_default_bench.json" % (tag, cfg_name))
            if not (os.path.exists(fsq) and os.path.exists(fb)) or launch_ms <= 0:
                continue
            try:
                prof_tiles = json.loads(open(fb).read().strip().splitlines()[-1])["config"]["tiles_32x32"]
                short = kern.replace("snb::", "")
                for line in open(fsq):
                    if short in line and "SQ_INSTS_VALU" in line:
                        insts = float(line.split("'SQ_INSTS_VALU':")[1].split(",")[0].split("}")[0])
                        scaled = insts * tiles / max(prof_tiles, 1)
                        simds, clock_ghz = 1024, 2.4
                        issue_ms = scaled * 4.0 / (simds * clock_ghz * 1e9) * 1e3
                        return {"insts_per_launch": int(scaled), "source": "profiles/%s: SQ_INSTS_VALU %d at %d tiles, scaled to %d tiles" % (os.path.basename(fsq), int(insts), prof_tiles, tiles),
                                "cycles_per_inst": 4, "simds": simds, "clock_ghz": clock_ghz, "issue_ms": round(issue_ms, 4), "frac": round(issue_ms / launch_ms, 4), "measured_in_this_run": False}
            except Exception:
                continue
        return None

    def pair_roofline(stx, derivatives):
        """N*(posq+sigeps+subset) + N*24 force write + T*32*(posq+sigeps+subset+index) + T*32*24 j-force scatter, over the
        average launch duration of the pair kernel on the region's eager steps (begin/end stamps of hipExtLaunchKernelGGL)."""
        Tx = int(stx.n_tiles)
        d_ms = stx.sum_direct_ms / max(stx.n_timed, 1)
        nbytes = N * ((4 + 2) * itemsize + 4) + N * 24 + Tx * 32 * ((4 + 2) * itemsize + 4 + 4) + Tx * 32 * 24
        ach = nbytes / (d_ms * 1e-3) / 1e9 if d_ms > 0 else 0.0
        if is_double:
            kern = "snb::k_direct<double, %d, false, %s>" % (2 if method == 4 else 3, "true" if derivatives else "false")
        else:
            kern = "snb::k_directPacked<%d, true, %s, false, %s>" % (2 if method == 4 else 3, "true" if derivatives else "false", "true" if precision == "mixed" else "false")
        valu = valu_issue(kern, Tx, d_ms)
        return {"bound": "valu" if valu else "hbm", "bound_note": "VALU-issue-bound in practice (DESIGN.md 4.1); achieved / peak / frac are the HBM figures the bench contract asks for (algorithmic bytes over the launch time), valu_issue is the bound that applies",
                "valu_issue": valu, "kernel": kern, "step": "with derivatives (energies on the tiles of the bound slices)" if derivatives else "forces only",
                "timing": "kernel begin/end stamps of hipExtLaunchKernelGGL on the eager steps of the timed region (snb_set_timing_interval: at least 3 of them; these steps run serially, every kernel alone)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_bytes": int(nbytes), "tiles": Tx, "avg_launch_ms": round(d_ms, 4), "timed_launches": int(stx.n_timed)}

    def pme_rooflines(stx, nheld):
        """The reciprocal pipeline kernel by kernel, from the engine's own per-kernel stamps (snb_stats.sum_kernel_ms): algorithmic bytes of
        DESIGN.md section 4 / SURVEY 8(d) over the mean launch duration, against the same HBM peak."""
        rows = []
        def add(slot, name, nbytes):
            cnt = int(stx.n_kernel_timed[slot])
            if cnt <= 0:
                return
            ms = stx.sum_kernel_ms[slot] / cnt
            rows.append({"kernel": name, "algorithmic_bytes": int(nbytes), "avg_launch_us": round(ms * 1e3, 2),
                         "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None, "timed_launches": cnt})
        r = itemsize
        add(0, "k_gatherPositions (+ force clear + mesh cells)", N * (3 * r + 4 + 4 * r + 3 * r + 3 * r + 4))
        meshes = [(0, grid, "")] + ([(8, dgrid, " [dispersion mesh]")] if method == 5 else [])
        for off, g, tag in meshes:
            Gm = g ** 3; Ghm = g * g * (g // 2 + 1)
            # charge spreading: either one kernel that also runs the forward z FFT (scanning brick spreader), or the own-atoms spreader's
            # two kernels (k_spreadOwn leaves per-work-group regions, k_spreadMerge sums them and runs the forward z FFT).  Algorithmic
            # bytes are the pipeline's: atoms in, half-complex mesh out -- the regions in between are not credited.
            fused = int(stx.n_kernel_timed[off + 2]) == 0
            if fused:
                add(off + 1, "k_spreadBrick (+ forward z FFT)" + tag, N * (4 * r + 4) + nheld * Ghm * 2 * r)
            else:
                add(off + 1, "k_spreadOwn (atoms -> per-work-group regions)" + tag, N * (4 * r + 4))
                add(off + 2, "k_spreadMerge (regions -> mesh, + forward z FFT)" + tag, nheld * Ghm * 2 * r)
            # plane path (square single-precision meshes whose (subset, kz) plane fits LDS): no y passes were stamped -- k_planeXY does
            # FFT_y, FFT_x, the convolution and both inverses of a plane in LDS (one read and one write of the complex meshes, + the kernel
            # table), k_fftZInvMix mixes on the matrix cores and runs the inverse z FFT
            plane = int(stx.n_kernel_timed[off + 3]) == 0 and int(stx.n_kernel_timed[off + 5]) == 0 and int(stx.n_kernel_timed[off + 4]) > 0
            if plane:
                add(off + 4, "k_planeXY (FFT_y + FFT_x + convolution + slice energies + inverse FFT_x + FFT_y, one (subset, kz) plane in LDS)" + tag, nheld * Ghm * 4 * r + Ghm * r)
                add(off + 6, "k_fftZInvMix (lambda mix on the matrix cores + inverse z FFT)" + tag, nheld * (Gm * r + Ghm * 2 * r))
            else:
                add(off + 3, "k_fftStrided (y, forward)" + tag, nheld * Ghm * 4 * r)
                add(off + 4, "k_convolveX (x FFT + slice energies + lambda mix + inverse x FFT)" + tag, nheld * Ghm * 4 * r)
                add(off + 5, "k_fftStrided (y, inverse)" + tag, nheld * Ghm * 4 * r)
                add(off + 6, "k_fftZ inverse" + tag, nheld * (Gm * r + Ghm * 2 * r))
            add(off + 7, "k_interpolateBricks (+ user-order force write)" + tag, N * (4 * r + 3 * r + 4 + 3 * r + 3 * r) + nheld * Gm * r)
        return rows

    nheld = len([s_ for s_ in range(nsub) if s_ % world == rank]) if world > 1 else nsub
    roof_plain = pair_roofline(st, False)
    roof_deriv = pair_roofline(st_d, True)
    roof = roof_deriv if headline_deriv else roof_plain
    direct_ms = st_head.sum_direct_ms / max(st_head.n_timed, 1)
    recip_ms = st_head.sum_recip_ms / max(st_head.n_timed, 1)
    gpu_ms = st_head.sum_total_ms / max(st_head.n_timed, 1)

    # one energy evaluation for the record (per-slice energies)
    eng.rebuild()
    eng.set_positions_device(pos0.data_ptr(), is_double)
    torch.cuda.synchronize()
    energy_ms = 1e30
    for _ in range(4):      # best of four: the first may coincide with a scheduled list rebuild
        t1 = time.perf_counter(); e_total = eng.execute(True); eng.sync(); energy_ms = min(energy_ms, (time.perf_counter() - t1) * 1e3)
    out = {
        "metric": "ns/day (force evaluation only, dt = 2 fs)", "value": round(ns_day, 3), "unit": "ns/day", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f64" if is_double else "f32", "data": "synthetic",
        "value_step": "with lambda derivatives (snb_execute include_energy = 2 every step)" if headline_deriv else "forces only",
        "value_forces_only": round(86.4 * 2.0 / plain_ms, 3), "ms_per_step_forces_only": round(plain_ms, 4),
        "value_with_derivatives": round(86.4 * 2.0 / deriv_ms, 3), "ms_per_step_with_derivatives": round(deriv_ms, 4),
        "config": {"workload": ("%s: %d-atom " + ("triclinic cell a=(L,0,0) b=(L/3,L,0) c=(-L/4,L/4,L)" if "box" in w else "cubic box") + " L=%.3f nm, %d subsets, %s %d^3%s, cutoff 1.0 nm, alpha 2.6283/nm, %d exclusions, list skin %.2f nm")
                   % (cfg_name, N, Lbox, nsub, "PME" if method == 4 else "LJPME", st.grid[0], (" + dispersion %d^3" % st.dgrid[0]) if method == 5 else "",
                      len(w["exc_qq"]), args.padding),
                   "tiles_32x32": T, "blocks": int(st.n_blocks), "rebuild_interval": args.rebuild_interval, "rebuilds_in_timed_region": rebuilds_in_region, "host_rebuilds": int(st.n_host_rebuilds), "neighbor_rebuild_ms": round(st.last_rebuild_ms, 2),
                   "ms_per_step_resident_coordinates": round(resident_ms, 4) if resident_ms is not None else None,
                   "derivative_slices": [int(i) for i in np.nonzero(deriv_slices)[0]], "list_overruns": overruns, "preconditioning_steps": precondition, "allreduce_ms": round(allreduce_ms, 4) if allreduce_ms is not None else None,
                   "first_call_ms": round(first_ms, 1), "energy_step_ms": round(energy_ms, 3), "energy_step_gpu_ms": round(eng.stats().last_total_ms, 3),
                   "gpu_ms_per_step": round(gpu_ms, 4), "direct_kernel_ms": round(direct_ms, 4), "reciprocal_ms": round(recip_ms, 4),
                   # replayed steps run the PME chain beside a CU-limited resident launch of the pair kernel (engine.hip overlapMode; the eager stamped
                   # steps behind `roofline` stay serial, so every kernel is still timed alone): the engine's own rule, restated
                   "overlap": os.environ.get("SNB_OVERLAP", "1") != "0" and T >= int(os.environ.get("SNB_OVERLAP_MIN_TILES", "100000")),
                   # the list of the next interval is built on a stream of its own while the last steps of this one run (engine.hip startSideBuild): restated rule
                   "rebuild_beside_steps": os.environ.get("SNB_SIDE_REBUILD", "1") != "0" and args.rebuild_interval > int(os.environ.get("SNB_SIDE_LEAD", "3")) + 1 and args.padding > 0,
                   "parallelism": ("subset-grid + i-block sharding x%d, RCCL all-reduce of forces" % world) if world > 1 else "1 GPU"},
        "roofline": roof,
        "roofline_other_step": roof_plain if headline_deriv else roof_deriv,
        "roofline_pme": pme_rooflines(st_head, nheld),
    }
    # HBM-side traffic of the pair kernel: PMC passes cannot run inside this process (rocprofv3 --pmc wraps the whole command, in separate
    # FETCH_SIZE / WRITE_SIZE passes: tools/pmc_hbm.sh).  `traffic` therefore stays null here; the figure of the committed profile of this
    # command is quoted under its own key, with the tile count it was taken at.
    for tag in ("r04", "r03", "r02"):
        pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "%s_%s_pmc_hbm.json" % (tag, cfg_name))
        if os.path.exists(pmc_file) and world == 1:
            try:
                allrec = json.load(open(pmc_file))
                rec = allrec.get("k_direct_derivatives" if headline_deriv else "k_direct_forces") or allrec["k_direct_forces"]
                out["roofline"]["traffic_from_profile"] = {"bytes_per_launch": int(rec["traffic_bytes_per_launch"]), "source": "profiles/" + os.path.basename(pmc_file) + ": " + rec["formula"],
                                                           "tiles_in_that_run": rec.get("tiles"), "measured_in_this_run": False}
            except Exception as exc:   # a malformed summary must not hide the measurement
                out["roofline"]["traffic_from_profile"] = {"error": "unreadable %s (%s)" % (pmc_file, exc)}
            break
    if block_ranges is not None:
        out["config"]["i_block_ranges_of_128"] = [list(r) for r in block_ranges]
    if world > 1:
        out["one_gpu_same_workload"] = one_gpu_same
        out["distributed"] = world_facts
    if rank == 0 and world == 1 and not args.no_auto_leg:
        # For the record (never `value`): the same workload, walk and step kind with DISPLACEMENT-TRIGGERED rebuilds (rebuild_interval < 0: the mode
        # the plugin adapter of INTEGRATION.md runs the engine in -- the reference relies on OpenMM's padded list doing the same).  These rebuilds
        # run in line.
        e2 = Engine(pkg, w, method, grid, dgrid, precision, local, 0, 1, args.padding, -100, stream=torch.cuda.current_stream().cuda_stream)
        p2 = pos0.clone(); f2 = torch.zeros((N, 3), dtype=tdtype, device=dev)
        e2.set_force_output(f2.data_ptr(), is_double); e2.set_energy_slices(deriv_slices); e2.set_positions_device(p2.data_ptr(), is_double); e2.set_timing_interval(0)

        def step2(i):
            p2.add_(walk[i % 16], alpha=float(walk_sign[i % len(walk_sign)]))
            e2.execute(2, fetch=False) if headline_deriv else e2.execute(False)
        for i in range(110):
            step2(i)
        e2.sync(); torch.cuda.synchronize()
        before2 = int(e2.stats().n_rebuilds)
        n2 = max(40, min(args.steps, 200))
        t2 = time.perf_counter()
        for i in range(n2):
            step2(110 + i)
        e2.sync(); torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t2) * 1e3 / n2
        st2 = e2.stats()
        out["displacement_triggered_rebuilds"] = {"ms_per_step": round(ms2, 4), "value": round(86.4 * 2.0 / ms2, 3), "unit": "ns/day", "steps": n2, "rebuilds": int(st2.n_rebuilds) - before2,
                                                  "list_overruns": int(st2.n_list_overruns), "padded_atoms": int(st2.n_padded_atoms), "step": "with derivatives" if headline_deriv else "forces only",
                                                  "note": "snb_config.rebuild_interval = -100: rebuild when an atom has moved 0.8 * skin / 2; same atoms, walk and step kind as `value`"}
        e2.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The CPU oracle on the FULL workload (round 4: its pair list and pair loops are threaded, one evaluation of c3 takes seconds, so
        # nothing is scaled any more): one evaluation to size the sample, then as many as fit ~20 s, at least 2, at most 6.
        import oracle as _o
        cores = int(_o.lib().orc_num_threads())
        _, _, dt0, pairs = oracle_eval(w, method, grid, dgrid)
        reps = int(max(2, min(6, round(20.0 / max(dt0, 1e-3))))); tsum = 0.0
        for _ in range(reps):
            _, _, dt, pairs = oracle_eval(w, method, grid, dgrid)
            tsum += dt
        cpu_full_ms = tsum / reps * 1e3
        out["cpu_baseline"] = {"value": round(86.4 * 2.0 / cpu_full_ms, 5), "unit": "ns/day", "cores": cores, "kind": "port",
                               "sample": "CPU oracle (C restatement of the Reference platform; pair list, pair loop and PME passes on %d OpenMP threads; the reference itself is single-threaded): "
                                         "%d evaluations (forces + slice energies) of the full workload, %d atoms, %d pairs inside the cutoff: %.0f ms per evaluation" % (cores, reps, N, pairs, cpu_full_ms),
                               "ms_per_step": round(cpu_full_ms, 1)}
        # the 24k-atom box of the same generator (density, cutoff, alpha, 54^3 mesh): parity sample of the precision legs below
        ws = build_workload(24000, 6.2145, min(nsub, 4), np.random.default_rng(SEED))
        ws["pos"] = np.ascontiguousarray(ws["pos"].astype(np.float32).astype(np.float64))      # float-representable coordinates: the oracle and engines of either precision see identical inputs
        ms24, ds24 = (4 if method == 4 else 5), (27 if method == 5 else 0)
        fo_s, so_s, _, _ = oracle_eval(ws, ms24, 54, ds24)
    if rank == 0 and world == 1 and not args.no_double and not is_double:
        # north_star couples the roofline target with "per-slice energies to 1e-5", which only double precision delivers: the same workload in
        # SNB_DOUBLE beside the headline (its own K-step regions, same walk and cadence), the pair kernel against the double-precision byte
        # model (76 N + 2560 T), and the worst slice-energy / force error of both precisions against the oracle on the 24k-atom sample.
        eng.close()
        K2 = max(10, min(args.steps, 60))
        engd = Engine(pkg, w, method, grid, dgrid, "double", local, 0, 1, args.padding, args.rebuild_interval, stream=torch.cuda.current_stream().cuda_stream)
        posd = pos0.double().clone(); forcesd = torch.zeros((N, 3), dtype=torch.float64, device=dev)
        walkd = [x.double() for x in walk[:4]]
        engd.set_force_output(forcesd.data_ptr(), True)
        engd.set_energy_slices(deriv_slices)

        def dstep(i, derivatives):
            posd.add_(walkd[i % 4], alpha=float(walk_sign[i % len(walk_sign)]))
            engd.set_positions_device(posd.data_ptr(), True)
            engd.execute(2, fetch=False) if derivatives else engd.execute(False)
            engd.forces_to(forcesd.data_ptr(), True)
        res = {}
        engd.set_timing_interval(max(1, K2 // 5))
        for name, derivatives in (("forces_only", False), ("with_derivatives", True)):
            for i in range(25):
                dstep(i, derivatives)
            engd.sync(); torch.cuda.synchronize(); engd.reset_timers()
            td0 = time.perf_counter()
            for i in range(K2):
                dstep(25 + i, derivatives)
            engd.sync(); torch.cuda.synchronize()
            ms = (time.perf_counter() - td0) * 1e3 / K2
            sd = engd.stats()
            d_ms = sd.sum_direct_ms / max(sd.n_timed, 1); Td = int(sd.n_tiles)
            nb = N * (6 * 8 + 4) + N * 24 + Td * 32 * (6 * 8 + 4 + 4) + Td * 32 * 24
            res[name] = {"ms_per_step": round(ms, 4), "ns_day": round(86.4 * 2.0 / ms, 2), "direct_kernel_ms": round(d_ms, 4), "reciprocal_ms": round(sd.sum_recip_ms / max(sd.n_timed, 1), 4),
                         "pair_kernel_algorithmic_bytes": int(nb), "pair_kernel_frac_of_hbm_peak": round(nb / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if d_ms > 0 else None, "tiles": Td}
        engd.close()
        out["double_precision"] = {"workload": cfg_name + " in SNB_DOUBLE (same atoms, mesh, walk and rebuild cadence)", "steps": K2, **res,
                                   "byte_model": "76*N + 2560*T (SURVEY 8d with 32-byte positions and 8-byte reals)"}
        if not args.no_cpu_baseline and fo_s is not None:
            par = {}
            rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))
            for pr in ("double", "single", "mixed"):
                e24 = Engine(pkg, ws, ms24, 54, ds24, pr, local, 0, 1, 0.1, 1 << 30, stream=torch.cuda.current_stream().cuda_stream)
                dtp = torch.float64 if pr == "double" else torch.float32
                p24 = torch.tensor(ws["pos"], dtype=dtp, device=dev); f24 = torch.zeros((len(ws["q"]), 3), dtype=dtp, device=dev)
                e24.set_positions_device(p24.data_ptr(), pr == "double"); e24.execute(True); e24.forces_to(f24.data_ptr(), pr == "double"); e24.sync()
                se = e24.slice_energies(so_s.shape[0]); f = f24.double().cpu().numpy()
                par[pr] = {"max_slice_energy_rel_err": rel(se, so_s),
                           "max_force_rel_err": float(np.max(np.linalg.norm(f - fo_s, axis=1) / np.maximum(np.linalg.norm(fo_s, axis=1), 1.0)))}
                if pr != "double":
                    # where the single-precision slice-energy error sits: each half of the Ewald split against the oracle's
                    for half, (d_, r_) in (("direct_space_only", (1, 0)), ("reciprocal_only", (0, 1))):
                        _, so_h, _, _ = oracle_eval(ws, ms24, 54, ds24, d_, r_)
                        eh = ctypes.c_double(); e24.ok(e24.L.snb_execute(e24.h, 1, 1, d_, r_, ctypes.byref(eh)))
                        # (each half's absolute error over the magnitude of the FULL slice energy: the two figures add up to the total's at worst)
                        par[pr]["slice_energy_err_of_%s_over_full_magnitude" % half] = float(np.max(np.abs(e24.slice_energies(so_h.shape[0]) - so_h) / np.maximum(np.abs(so_s), 1.0)))
                e24.close()
            out["double_precision"]["parity_vs_oracle_24k_atoms"] = par
        eng = None
    if args.check and rank == 0 and world == 1:
        # parity at full size, as tests/test_gpu_fullsize.py does it: the oracle on the coordinates the engine was given (float-rounded in
        # single precision), the pairs within float rounding of the cutoff accounted for one by one (tests/parity_tools.py)
        import parity_tools as pt
        wc = pt.float_positions(w) if not is_double else w
        fo, so, _, _ = oracle_eval(wc, method, grid, dgrid)
        fa, ea, nband = pt.band_allowance(wc, method, grid, dgrid, pt.band_rel(wc, precision))
        tol = 1e-5 if is_double else 1e-3
        if eng is None:
            eng = Engine(pkg, w, method, grid, dgrid, precision, local, rank, world, args.padding, args.rebuild_interval, stream=torch.cuda.current_stream().cuda_stream)
        eng.rebuild()
        eng.set_positions_device(pos0.data_ptr(), is_double); eng.execute(True); eng.forces_to(forces.data_ptr(), is_double); eng.sync()
        rec_e = pt.compare(forces.double().cpu().numpy(), eng.slice_energies(so.shape[0]), fo, so, tol, fa, ea)
        eng.set_positions_device(pos0.data_ptr(), is_double); eng.execute(False); eng.forces_to(forces.data_ptr(), is_double); eng.sync()
        rec_f = pt.compare(forces.double().cpu().numpy(), None, fo, so, tol, fa, ea)
        out["check"] = {"tolerance": tol, "band_pairs": nband, "energy_step": rec_e, "forces_only_step": rec_f}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
