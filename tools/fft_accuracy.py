"""(GPU box) accuracy of the engine's single-precision 3D FFT on the bench meshes: max and rms error of the spectrum against NumPy
(double), relative to the rms spectrum amplitude, and of the unnormalised round trip."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
snb = importlib.import_module("openmm-nonbonded-slicing_amd")
L = snb.capi.lib(); dp = ctypes.POINTER(ctypes.c_double)
rng = np.random.default_rng(5)
for n in (80, 90, 120, 180):
    a = rng.standard_normal((1, n, n, n))
    spec = np.zeros((1, n, n, n // 2 + 1, 2)); rt = np.zeros_like(a)
    assert L.snb_test_fft3d(0, 0, 1, n, n, n, a.ctypes.data_as(dp), spec.ctypes.data_as(dp), rt.ctypes.data_as(dp)) == 0
    ref = np.fft.rfftn(a[0]); got = spec[0, ..., 0] + 1j * spec[0, ..., 1]
    rms = np.sqrt(np.mean(np.abs(ref) ** 2)); err = np.abs(got - ref)
    print("n=%d  spectrum: max err / rms %.3g   rms err / rms %.3g    round trip: max err %.3g rms err %.3g" % (n, err.max() / rms, np.sqrt(np.mean(err ** 2)) / rms, np.abs(rt[0] / n ** 3 - a[0]).max(), np.sqrt(np.mean((rt[0] / n ** 3 - a[0]) ** 2))), flush=True)
