import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import systems
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle
F = snb.SlicedNonbondedForce
n, L = 13824, 6.0
force, pos, box = systems.random_box(F, n, 4, 4, L, 1.0, pme=(2.6283, 48, 48, 48), derivatives=False)
system = snb.System()
for _ in range(n): system.addParticle(1.0)
system.setDefaultPeriodicBoxVectors(*box); system.addForce(force)
ctx = snb.Context(system, precision="double", neighbor_padding=0.1, rebuild_interval=10)
ctx.setPositions(pos)
fr = ctx.getState(getForces=True).getForces()
fo = oracle.evaluate(force, pos, box, None, True, True)["forces"]
err = np.linalg.norm(fo - fr, axis=1) / np.maximum(np.linalg.norm(fo, axis=1), 1.0)
print("double forces-only (polynomial Ewald) vs oracle: max rel err %.3e, median %.3e" % (err.max(), np.median(err)))
