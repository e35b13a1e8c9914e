"""ctypes binding of libvcmi.so (the C-ABI declared in include/vcmi.h).

The HIP library is the product: if it is missing this module raises at import time -- there is no
CPU fallback anywhere in the package (the CPU oracle under oracle/ is test infrastructure only).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvcmi.so")
# A/B and probe builds (tools/ab_swap.sh, tools/convert_ab.py, tools/*_prof.sh) are SELECTED through this variable, never
# copied over the in-tree library (an interrupted tool left tests and bench running a probe build).  bench.py
# marks a line produced with it (`probe_library`).
PROBE = os.environ.get("LIBVCMI_PROBE")
if PROBE:
    LIB_PATH = os.path.abspath(PROBE)


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (src/gmmmap.jl:102, src/trajectory_gmmmap.jl:68, src/align.jl:11-13)."""


class PosDefException(ArithmeticError):
    """Julia's PosDefException, raised by MvNormal's Cholesky in src/gmm.jl:17."""


class VCMIError(RuntimeError):
    """Any other libvcmi failure (HIP error, out of memory, bad argument, no device)."""


VCMI_OK, VCMI_ERR_DIM, VCMI_ERR_NOT_PD, VCMI_ERR_HIP, VCMI_ERR_OOM, VCMI_ERR_ARG, VCMI_ERR_NO_DEVICE = range(7)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C voiceconversion.jl_amd/csrc). There is no CPU fallback."
    )

# PyTorch-ROCm bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Two HIP
# runtimes in one process cannot both own the GPU, so torch is imported FIRST: libvcmi's NEEDED
# libamdhip64.so.7 then binds to the runtime torch already loaded, and device pointers / streams are shared.
# (A Julia or C++ host that never loads torch binds to /opt/rocm/lib instead.)
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_dpp = C.POINTER(_dp)
_ipp = C.POINTER(_ip)
_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# name -> (restype, argtypes); every symbol include/vcmi.h declares
SIGNATURES = {
    "vcmi_last_error": (C.c_char_p, []),
    "vcmi_version": (C.c_char_p, []),
    "vcmi_device_count": (_int, [C.POINTER(_int)]),
    "vcmi_set_device": (_int, [_int]),
    "vcmi_set_devices": (_int, [C.POINTER(_int), _int]),
    "vcmi_get_devices": (_int, [C.POINTER(_int), _int, C.POINTER(_int)]),
    "vcmi_host_register": (_int, [_vp, C.c_size_t]),
    "vcmi_host_unregister": (_int, [_vp]),
    "vcmi_host_is_registered": (_int, [_vp, C.c_size_t, C.POINTER(_int)]),
    "vcmi_gmmmap_create": (_int, [_dp, _dp, _dp, _int, _int, _int, C.POINTER(_vp)]),
    "vcmi_gmmmap_destroy": (_int, [_vp]),
    "vcmi_gmmmap_dim": (_int, [_vp]),
    "vcmi_gmmmap_ncomponents": (_int, [_vp]),
    "vcmi_gmmmap_get_A": (_int, [_vp, _dp]),
    "vcmi_gmmmap_convert": (_int, [_vp, _dp, _i64, _i64, _dp, _i64]),
    "vcmi_gmmmap_convert_dev": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "vcmi_vc_frames": (_int, [_vp, _dp, _i64, _dp]),
    "vcmi_vc_frames_postf": (_int, [_vp, _dp, _i64, _dp, _dp]),
    "vcmi_gmmmap_posterior": (_int, [_vp, _dp, _i64, _i64, _dp]),
    "vcmi_gmmmap_posterior_dev": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "vcmi_gmmmap_predict": (_int, [_vp, _dp, _i64, _i64, _ip]),
    "vcmi_gmmmap_predict_dev": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "vcmi_gmmmap_set_kernel": (_int, [_vp, _int]),
    "vcmi_gmmmap_set_prune": (_int, [_vp, C.c_double]),
    "vcmi_gmmmap_prune_stats": (_int, [_vp, _int, _ip]),
    "vcmi_gmmmap_convert_plan": (_int, [_vp, _ip, C.POINTER(_int), _dp, _dp]),
    "vcmi_dtw_fit": (_int, [_dp, _i64, _dp, _i64, _int, _int, _int, _ip, _dp, _ip]),
    "vcmi_dtw_fit_batch": (_int, [_i64, _dpp, _ip, _dpp, _ip, _int, _int, _int, _ipp]),
    "vcmi_dtw_fit_batch_dev": (_int, [_i64, _vp, _ip, _ip, _ip, _ip, _int, _int, _int, _vp, _ip, _vp]),
    "vcmi_align": (_int, [_dp, _i64, _dp, _i64, _int, _dp, _ip]),
    "vcmi_align_batch": (_int, [_i64, _dpp, _ip, _dpp, _ip, _int, _dpp]),
    "vcmi_estep_diag": (_int, [_dp, _i64, _int, _int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "vcmi_estep_stats_len": (_i64, [_int, _int]),
    "vcmi_estep_set_path": (_int, [_int]),
    "vcmi_estep_get_path": (_int, [C.POINTER(_int)]),
    "vcmi_estep_diag_dev": (_int, [_vp, _i64, _int, _int, _dp, _dp, _dp, _vp, _vp]),
    "vcmi_estep_full_stats_len": (_i64, [_int, _int]),
    "vcmi_estep_full": (_int, [_dp, _i64, _int, _int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "vcmi_estep_full_dev": (_int, [_vp, _i64, _int, _int, _dp, _dp, _dp, _vp, _vp]),
    "vcmi_gmm_em_create": (_int, [_int, _int, _dp, _dp, _dp, C.c_double, C.POINTER(_vp)]),
    "vcmi_gmm_em_destroy": (_int, [_vp]),
    "vcmi_gmm_em_estep_dev": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "vcmi_gmm_em_mstep": (_int, [_vp, _vp, _vp, _dp]),
    "vcmi_gmm_em_get": (_int, [_vp, _dp, _dp, _dp]),
    "vcmi_traj_create": (_int, [_vp, _i64, C.POINTER(_vp)]),
    "vcmi_traj_destroy": (_int, [_vp]),
    "vcmi_traj_length": (_i64, [_vp]),
    "vcmi_traj_convert": (_int, [_vp, _dp, _i64, _dp]),
    "vcmi_traj_convert_batch": (_int, [_vp, _i64, _dpp, _ip, _dpp]),
    "vcmi_traj_convert_batch_dev": (_int, [_vp, _i64, _vp, _ip, _ip, _vp, _ip, _vp]),
    "vcmi_vc_traj": (_int, [_vp, _dp, _i64, _dp]),
    "vcmi_push_delta": (_int, [_dp, _int, _i64, _dp]),
    "vcmi_push_delta_dev": (_int, [_vp, _i64, _int, _i64, _vp, _i64, _vp]),
    "vcmi_vc_traj_postf": (_int, [_vp, _dp, _i64, _dp, _dp]),
    "vcmi_trajgv_create": (_int, [_vp, _dp, _dp, C.POINTER(_vp)]),
    "vcmi_trajgv_destroy": (_int, [_vp]),
    "vcmi_trajgv_convert": (_int, [_vp, _dp, _i64, _int, C.c_double, _dp]),
    "vcmi_trajgv_convert_batch": (_int, [_vp, _i64, _dpp, _ip, _int, C.c_double, _dpp]),
    "vcmi_trajgv_convert_batch_dev": (_int, [_vp, _i64, _vp, _ip, _ip, _int, C.c_double, _vp, _ip, _vp]),
    "vcmi_variance_scaling": (_int, [_dp, _int, _i64, _dp, _dp]),
    "vcmi_variance_scaling_dev": (_int, [_vp, _i64, _int, _i64, _dp, _vp, _i64, _vp]),
    "vcmi_diffgmm": (_int, [_dp, _dp, _int, _int, _dp, _dp]),
    "vcmi_mc2e": (_int, [_dp, _int, _i64, C.c_double, _int, _dp]),
    "vcmi_align_mcep": (_int, [_dp, _i64, _dp, _i64, _int, C.c_double, _int, C.c_double, _int, _dp, _dp, _ip]),
    "vcmi_parallel_dataset_dev": (_int, [_i64, _dpp, _ip, _dpp, _ip, _int, _int, C.c_double, _int, C.c_double, _int, _int, _int,
                                         _int, _vp, _i64, _ip, _ip]),
    "vcmi_gv_dataset": (_int, [_i64, _dpp, _ip, _int, _int, _int, _dp, _ip]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)      # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.vcmi_last_error().decode("utf-8", "replace")


def check(status):
    """Map a vcmi_status to the exception type the reference would have thrown."""
    if status == VCMI_OK:
        return
    msg = last_error()
    if status == VCMI_ERR_DIM:
        raise DimensionMismatch(msg)
    if status == VCMI_ERR_NOT_PD:
        raise PosDefException(msg)
    raise VCMIError(f"libvcmi status {status}: {msg}")


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def device_count():
    n = _int(0)
    check(lib.vcmi_device_count(C.byref(n)))
    return n.value


def set_device(i):
    check(lib.vcmi_set_device(int(i)))


def set_devices(devices):
    """One host process, several GPUs: the host-pointer entry points shard over `devices` (see include/vcmi.h).
    An empty list restores the single-device behaviour."""
    devices = [int(d) for d in devices]
    arr = (_int * max(len(devices), 1))(*devices)
    check(lib.vcmi_set_devices(arr, len(devices)))


def get_devices():
    n = _int(0)
    arr = (_int * 64)()
    check(lib.vcmi_get_devices(arr, 64, C.byref(n)))
    return [arr[i] for i in range(n.value)]


def pin(a):
    """Page-lock a host array once (vcmi_host_register): calls that pass it afterwards DMA straight from / into it.
    The array must stay alive and must be unpinned before it is freed; returns the array."""
    if not (a.flags.c_contiguous or a.flags.f_contiguous):
        raise ValueError("pin: the array must be contiguous (a.nbytes would not describe the memory of a strided view)")
    check(lib.vcmi_host_register(a.ctypes.data, a.nbytes))
    return a


def unpin(a):
    check(lib.vcmi_host_unregister(a.ctypes.data))


def is_pinned(a):
    f = _int(0)
    check(lib.vcmi_host_is_registered(a.ctypes.data, a.nbytes, C.byref(f)))
    return bool(f.value)


# Test hook (NOT part of include/vcmi.h; inert unless VCMI_TEST_HOOKS=1 is in the environment when the library first sees
# the call): force the fallback kernels a shape would not select by itself.
DBG_TRAJ_GENERIC, DBG_TRAJ_G_SCALAR, DBG_GV_ONE_TEAM, DBG_PREDICT_TWO_PASS, DBG_ESTEP_GENERIC, DBG_DTW_TWO_KERNELS = 1, 2, 4, 8, 16, 32
DBG_PREDICT_NO_EARLY_EXIT = 64
DBG_TRAJ_ONE_WG_PER_CU = 128
DBG_CONVERT_NO_GROUPING = 2048
DBG_CONVERT_SHAPE_BROAD, DBG_CONVERT_SHAPE_PEAKED = 4096, 8192
DBG_DTW_NO_SEGMENTS, DBG_DTW_GRID_ORDER = 256, 512
DBG_DTW_TWO_SEGMENTS, DBG_DTW_WHOLE_FIRST = 16384, 32768
DBG_ESTEP_FULL_NO_LISTS = 65536
DBG_CONVERT_WIDE_TILES = 131072
DBG_CONVERT_SHAPE_SCREENED = 262144
DBG_SCREEN_ROWS4, DBG_SCREEN_ROWS2, DBG_SCREEN_ROWS1 = 2097152, 524288, 1048576
DBG_ESTEP_WAVE_KERNEL = 4194304
DBG_PREDICT_NO_SCREEN = 8388608
DBG_PREDICT_SCREEN = 16777216
DBG_SCREEN_FP64 = 33554432
DBG_GROUP_KEY_FP64 = 67108864
DBG_ESTEP_NO_HARD = 134217728
DBG_ESTEP_NO_SMALL = 1024


def estep_last_soft():
    """Measurement hook: frames of this thread's last diagonal E-step that went through the FP64 kernel after the
    hard-assignment screen (csrc/estep_hard.hpp); -1 when the call took the one-kernel path."""
    fn = lib.vcmi_debug_estep_last_soft
    fn.argtypes = [C.POINTER(C.c_int64)]
    fn.restype = _int
    v = C.c_int64(0)
    check(fn(C.byref(v)))
    return int(v.value)


def debug_force(flags):
    lib.vcmi_debug_force.argtypes = [C.c_uint]
    lib.vcmi_debug_force.restype = _int
    check(lib.vcmi_debug_force(int(flags)))       # fails unless the process was started with VCMI_TEST_HOOKS=1
