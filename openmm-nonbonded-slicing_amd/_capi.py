"""ctypes binding of the C ABI in ``include/snb.h`` (``libsnb_hip.so``).

This is all the FFI there is: plain pointers and sizes.  There is deliberately no CPU fallback -- if the HIP
library is missing, loading fails loudly; if there is no GPU, ``snb_create`` returns ``SNB_ERR_HIP``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNB_LIB_PATH") or os.path.join(_HERE, "libsnb_hip.so")      # (SNB_LIB_PATH: an experimental build of the same ABI, tools/ only)
SNB_ABI_VERSION = 6

# every symbol include/snb.h declares (tests check the library exports each one)
SYMBOLS = [
    "snb_create", "snb_destroy", "snb_last_error", "snb_set_particles", "snb_set_exceptions", "snb_set_parameter_offsets", "snb_set_global_parameters", "snb_set_energy_slices", "snb_set_lambdas",
    "snb_set_dispersion_coefficients", "snb_compute_dispersion_coefficients", "snb_set_box", "snb_set_positions",
    "snb_rebuild_neighbors", "snb_execute", "snb_get_forces", "snb_set_force_output", "snb_set_shard_blocks", "snb_get_slice_energies", "snb_slice_energies_device", "snb_synchronize",
    "snb_get_pme_parameters", "snb_get_ljpme_parameters", "snb_get_stats", "snb_reset_timers", "snb_set_timing_interval", "snb_legal_grid_size", "snb_abi_version",
    "snb_test_fft3d",
]

SNB_OK, SNB_ERR_INVALID_ARGUMENT, SNB_ERR_HIP, SNB_ERR_BOX_TOO_SMALL, SNB_ERR_NOT_PME, SNB_ERR_STATE, SNB_ERR_UNSUPPORTED = range(7)


class SnbConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("n_atoms", ctypes.c_int32), ("n_subsets", ctypes.c_int32), ("method", ctypes.c_int32),
        ("precision", ctypes.c_int32), ("use_switch", ctypes.c_int32), ("exceptions_periodic", ctypes.c_int32), ("device", ctypes.c_int32),
        ("cutoff", ctypes.c_double), ("switch_distance", ctypes.c_double), ("rf_dielectric", ctypes.c_double), ("alpha", ctypes.c_double),
        ("grid", ctypes.c_int32 * 3), ("kmax", ctypes.c_int32 * 3), ("alpha_d", ctypes.c_double), ("dgrid", ctypes.c_int32 * 3),
        ("neighbor_padding", ctypes.c_double), ("rebuild_interval", ctypes.c_int32), ("shard_rank", ctypes.c_int32),
        ("shard_count", ctypes.c_int32), ("disable_graph", ctypes.c_int32), ("host_neighbor_build", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SnbStats(ctypes.Structure):
    _fields_ = [
        ("n_tiles", ctypes.c_int64), ("n_blocks", ctypes.c_int64), ("n_padded_atoms", ctypes.c_int64), ("n_exclusion_tiles", ctypes.c_int64),
        ("n_exclusions", ctypes.c_int64), ("n_14", ctypes.c_int64), ("n_rebuilds", ctypes.c_int64), ("grid", ctypes.c_int32 * 3),
        ("dgrid", ctypes.c_int32 * 3), ("last_direct_ms", ctypes.c_double), ("last_recip_ms", ctypes.c_double),
        ("last_total_ms", ctypes.c_double), ("last_rebuild_ms", ctypes.c_double), ("sum_direct_ms", ctypes.c_double),
        ("sum_recip_ms", ctypes.c_double), ("sum_total_ms", ctypes.c_double), ("n_timed", ctypes.c_int64), ("n_host_rebuilds", ctypes.c_int64), ("n_list_overruns", ctypes.c_int64),
        ("sum_kernel_ms", ctypes.c_double * 16), ("n_kernel_timed", ctypes.c_int64 * 16), ("n_spread_strays", ctypes.c_int64),
    ]
KERNEL_SLOTS = ("gather", "spread", "fft_z_forward", "fft_y_forward", "convolve_x", "fft_y_inverse", "fft_z_inverse", "interpolate")


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``libsnb_hip.so`` (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))] + [os.path.join(_HERE, "..", "include", "snb.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", csrc])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsnb_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP engine has no CPU fallback)")
    L = ctypes.CDLL(LIB_PATH)
    dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int32); vp = ctypes.c_void_p; i32 = ctypes.c_int32
    L.snb_create.argtypes = [ctypes.POINTER(SnbConfig), ctypes.POINTER(vp)]
    L.snb_destroy.argtypes = [vp]; L.snb_destroy.restype = None
    L.snb_last_error.argtypes = [vp]; L.snb_last_error.restype = ctypes.c_char_p
    L.snb_set_particles.argtypes = [vp, dp, dp, dp, ip]
    L.snb_set_exceptions.argtypes = [vp, i32, ip, dp, dp, dp, ip]
    L.snb_set_lambdas.argtypes = [vp, dp]
    L.snb_set_energy_slices.argtypes = [vp, ip]
    L.snb_set_parameter_offsets.argtypes = [vp, i32, i32, ip, ip, dp, i32, ip, ip, dp]
    L.snb_set_global_parameters.argtypes = [vp, i32, dp]
    L.snb_set_dispersion_coefficients.argtypes = [vp, dp]
    L.snb_compute_dispersion_coefficients.argtypes = [i32, i32, dp, dp, ip, ctypes.c_double, i32, ctypes.c_double, dp]
    L.snb_set_box.argtypes = [vp, dp]
    L.snb_set_positions.argtypes = [vp, vp, i32, i32, i32]
    L.snb_rebuild_neighbors.argtypes = [vp]
    L.snb_execute.argtypes = [vp, i32, i32, i32, i32, dp]
    L.snb_get_forces.argtypes = [vp, vp, i32, i32, i32]
    L.snb_set_force_output.argtypes = [vp, vp, i32, i32]
    L.snb_set_shard_blocks.argtypes = [vp, i32, i32, i32]
    L.snb_get_slice_energies.argtypes = [vp, dp]
    L.snb_synchronize.argtypes = [vp]
    L.snb_slice_energies_device.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p)]
    L.snb_get_pme_parameters.argtypes = [vp, dp, ip]
    L.snb_get_ljpme_parameters.argtypes = [vp, dp, ip]
    L.snb_get_stats.argtypes = [vp, ctypes.POINTER(SnbStats)]
    L.snb_reset_timers.argtypes = [vp]
    L.snb_set_timing_interval.argtypes = [vp, i32]
    L.snb_legal_grid_size.argtypes = [i32]; L.snb_legal_grid_size.restype = i32
    L.snb_abi_version.restype = i32
    L.snb_test_fft3d.argtypes = [i32, i32, i32, i32, i32, i32, dp, dp, dp]
    for name in SYMBOLS:
        getattr(L, name)  # AttributeError if the header and the library ever diverge
    _lib = L
    return L
