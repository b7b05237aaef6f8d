"""MI355X-native SlicedNonbondedForce hot path (direct-space sliced pair loop + sliced PME).

Host-side mirror of the reference interface (``SlicedNonbondedForce``, the
``CalcSlicedNonbondedForceKernel`` boundary) over the C-ABI engine in ``csrc/`` (``include/snb.h``).
The directory name carries hyphens (it is fixed by the build contract), so import it with
``importlib.import_module("openmm-nonbonded-slicing_amd")``.
"""
from .force import OpenMMException, SlicedNonbondedForce, sliceIndex  # noqa: F401

ONE_4PI_EPS0 = 138.93545764438198  # OpenMM 8.3 SimTKOpenMMRealType.h (third-party constant)


def __getattr__(name):
    # lazy: these need the built HIP library
    if name in ("HipCalcSlicedNonbondedForceKernel", "Context", "System", "State", "capi"):
        import importlib
        if name == "capi":
            return importlib.import_module(__name__ + "._capi")
    if name in ("sharding", "serialization"):
        import importlib
        return importlib.import_module(__name__ + "." + name)
    if name == "XmlSerializer":
        import importlib
        return importlib.import_module(__name__ + ".serialization").XmlSerializer
    if name in ("HipCalcSlicedNonbondedForceKernel", "Context", "System", "State"):
        import importlib
        mod = importlib.import_module(__name__ + ".context")
        return getattr(mod, name)
    raise AttributeError(name)
