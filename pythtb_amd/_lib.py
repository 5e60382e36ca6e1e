"""ctypes binding of libtbk.so (the C ABI declared in include/tbk.h).

There is no CPU fallback: importing this module without the built library, or
running a compute call without a visible MI355X, raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TBK_LIBRARY selects another build of the same sources (libtbk_diag.so: ablation branches; libtbk_asan.so:
# host code under AddressSanitizer/UBSan), for profiles/ and the sanitizer run only
LIB_PATH = os.environ.get("TBK_LIBRARY") or os.path.join(_HERE, "libtbk.so")

MAX_DIM = 4
MAX_NSTA = 2048
MAX_NOCC = 16

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "\n\npythtb_amd: %s is missing.  Build it with\n"
        "    python -c 'import __graft_entry__ as g; g.build()'    (or: make -C pythtb_amd/csrc)\n"
        "There is no CPU fallback for the k-mesh kernels." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_p = C.c_void_p
_pp = C.POINTER(C.c_void_p)
_i = C.c_int
_i64 = C.c_int64
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# name -> (restype, argtypes); must list every symbol include/tbk.h declares
SIGNATURES = {
    "tbk_last_error": (C.c_char_p, []),
    "tbk_version": (_i, []),
    "tbk_device_count": (_i, [C.POINTER(C.c_int)]),
    "tbk_knobs_reload": (_i, []),
    "tbk_build_has_diagnostics": (_i, []),
    "tbk_one_phase_cont": (_i, [_dp, _i64, _i64, C.c_double, _dp, _i64]),
    "tbk_array_phases_cont": (_i, [_dp, _i64, _i, _i64, _dp, _dp, _i64]),
    "tbk_model_flatten_host": (_i, [_i, _i, _i, _dp, _dp, _i64, _ip, _ip, _ip, _dp, _i64, C.POINTER(C.c_int64), _ip, _ip, _dp, _ip]),
    "tbk_ctx_create": (_i, [_i, _pp]),
    "tbk_ctx_destroy": (_i, [_p]),
    "tbk_ctx_sync": (_i, [_p]),
    "tbk_ctx_device_info": (_i, [_p, C.c_char_p, _i, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "tbk_dev_alloc": (_i, [_p, _i64, _pp]),
    "tbk_dev_free": (_i, [_p, _p]),
    "tbk_dev_upload": (_i, [_p, _p, _p, _i64]),
    "tbk_dev_download": (_i, [_p, _p, _p, _i64]),
    "tbk_timer_begin": (_i, [_p]),
    "tbk_timer_end": (_i, [_p, _dp]),
    "tbk_prof_enable": (_i, [_p, _i]),
    "tbk_prof_reset": (_i, [_p]),
    "tbk_prof_calibrate": (_i, [_p, _i, _dp]),
    "tbk_prof_count": (_i, [_p, C.POINTER(C.c_int)]),
    "tbk_prof_get": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(C.c_int64), _dp]),
    "tbk_model_upload": (_i, [_p, _i, _i, _i, _dp, _dp, _i64, _ip, _ip, _ip, _dp, _pp]),
    "tbk_model_free": (_i, [_p]),
    "tbk_model_info": (_i, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "tbk_gen_ham": (_i, [_p, _dp, _i64, _dp]),
    "tbk_solve_list": (_i, [_p, _dp, _i64, _dp, _dp]),
    "tbk_solve_list_dev": (_i, [_p, _p, _i64, _p, _p]),
    "tbk_solve_list_dev_checked": (_i, [_p, _p, _i64, _p, _p]),
    "tbk_eigh_batch": (_i, [_p, _i, _dp, _i64, _dp, _dp]),
    "tbk_solver_regime": (C.c_char_p, [_i, _i, _i, _i64, _i64, _i, _i, C.POINTER(C.c_char_p)]),
    "tbk_wfs_create": (_i, [_p, _i, _ip, _i, _i, _pp]),
    "tbk_wfs_free": (_i, [_p]),
    "tbk_wfs_upload": (_i, [_p, _dp]),
    "tbk_wfs_download": (_i, [_p, _dp]),
    "tbk_wfs_copy_bands": (_i, [_p, _p, _ip, _i]),
    "tbk_wfs_device_ptr": (_i, [_p, _pp, C.POINTER(C.c_int64)]),
    "tbk_wfs_download_points": (_i, [_p, C.POINTER(C.c_int64), _i64, _dp]),
    "tbk_wfs_upload_points": (_i, [_p, C.POINTER(C.c_int64), _i64, _dp]),
    "tbk_ctx_transfer_stats": (_i, [_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), _i]),
    "tbk_ctx_solver_stats": (_i, [_p, C.POINTER(C.c_int64), _i]),
    "tbk_wfs_solve_grid": (_i, [_p, _p, _dp, _dp, _i64, _i64, _dp]),
    "tbk_wfs_solve_grid_async": (_i, [_p, _p, _dp, _dp, _i64, _i64]),
    "tbk_wfs_solve_grid_result": (_i, [_p, _dp]),
    "tbk_wfs_solve_grid_flux_async": (_i, [_p, _p, _dp, _dp, _i64, _i64, _ip, _i]),
    "tbk_wfs_solve_window_async": (_i, [_p, _p, _dp, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tbk_wfs_impose": (_i, [_p, _i, _dp]),
    "tbk_berry_flux": (_i, [_p, _ip, _i, _i, _i, _dp, _dp]),
    "tbk_berry_flux_async": (_i, [_p, _ip, _i, _i, _i, _i]),
    "tbk_berry_flux_result": (_i, [_p, _dp, _dp]),
    "tbk_berry_phase": (_i, [_p, _ip, _i, _i, _i, _dp]),
    "tbk_position_hwf": (_i, [_p, _dp, _i64, _i, _i, _dp, _dp, _dp, _dp, _i]),
    "tbk_wfs_position_hwf": (_i, [_p, C.POINTER(C.c_int64), _i64, _ip, _i, _dp, _dp, _dp, _dp, _i]),
    "tbk_k_uniform_mesh_dev": (_i, [_p, _i, _ip, _p]),
    "tbk_k_uniform_mesh_range_dev": (_i, [_p, _i, _ip, _i64, _i64, _p]),
    "tbk_k_path_dev": (_i, [_p, _i, _i, _dp, _ip, _i64, _p]),
    "tbk_solve_mesh": (_i, [_p, _ip, _dp, _dp]),
    "tbk_dos_mesh": (_i, [_p, _ip, _i, _dp, C.POINTER(C.c_int64), _dp, _dp]),
    "tbk_comm_unique_id": (_i, [C.POINTER(C.c_ubyte)]),
    "tbk_comm_init": (_i, [_p, C.POINTER(C.c_ubyte), _i, _i]),
    "tbk_comm_destroy": (_i, [_p]),
    "tbk_comm_allgather_f64": (_i, [_p, _p, _p, _i64]),
    "tbk_comm_allgatherv_f64": (_i, [_p, _p, _i64, _p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tbk_comm_allgatherv_rows_f64": (_i, [_p, _p, _i64, _i64, _p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i64]),
    "tbk_comm_gatherv_rows_f64": (_i, [_p, _p, _i64, _i64, _p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i64, _i]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = library/header mismatch
    _fn.restype = _res
    _fn.argtypes = _args


class TbkError(Exception):
    """Raised for device/runtime failures reported by libtbk."""


def check(rc):
    if rc != 0:
        msg = lib.tbk_last_error()
        raise TbkError("\n\nlibtbk error %d: %s" % (rc, msg.decode() if msg else "?"))


def dptr(a):
    """double* view of a C-contiguous float64/complex128 array (or NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def iptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"] and a.dtype == np.int32
    return a.ctypes.data_as(_ip)


class knob(object):
    """Context manager for A/B runs and tests: set one TBK_* environment knob and make the library re-read it."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        if self.value is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = str(self.value)
        lib.tbk_knobs_reload()
        return self

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old
        lib.tbk_knobs_reload()
        return False


class Context(object):
    """One device + one HIP stream.  Created lazily; one per process by default."""

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("TBK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = C.c_int(0)
        rc = lib.tbk_device_count(C.byref(n))
        if rc != 0 or n.value < 1:
            msg = lib.tbk_last_error()
            raise TbkError(
                "\n\npythtb_amd needs an AMD GPU (MI355X/gfx950): no HIP device is visible (%s).\n"
                "There is no CPU fallback for solve_all / solve_on_grid / berry_*." %
                (msg.decode() if msg else "device count 0"))
        device = device % n.value
        h = C.c_void_p()
        check(lib.tbk_ctx_create(device, C.byref(h)))
        self.handle = h
        self.device = device

    def sync(self):
        check(lib.tbk_ctx_sync(self.handle))

    def info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        hbm = C.c_int64(0)
        check(lib.tbk_ctx_device_info(self.handle, name, 256, C.byref(cus), C.byref(hbm)))
        return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=hbm.value)

    def transfer_stats(self, reset=False):
        """wf_array bytes/calls across PCIe since the last reset: dict(h2d_bytes, d2h_bytes, h2d_calls, d2h_calls)."""
        v = [C.c_int64(0) for _ in range(4)]
        check(lib.tbk_ctx_transfer_stats(self.handle, *[C.byref(x) for x in v], 1 if reset else 0))
        return dict(h2d_bytes=v[0].value, d2h_bytes=v[1].value, h2d_calls=v[2].value, d2h_calls=v[3].value)

    def solver_stats(self, reset=False):
        """dict(listed_matrices=...): matrices of 9..32 states the direct kernels handed to their fallback (the QL replay) since the last reset."""
        v = C.c_int64(0)
        check(lib.tbk_ctx_solver_stats(self.handle, C.byref(v), 1 if reset else 0))
        return dict(listed_matrices=v.value)

    # ---- timing helpers (HIP events on this context's stream)
    def timer_begin(self):
        check(lib.tbk_timer_begin(self.handle))

    def timer_end(self):
        ms = C.c_double(0.0)
        check(lib.tbk_timer_end(self.handle, C.byref(ms)))
        return ms.value

    def prof_enable(self, period=1):
        """0/False: off; 1/True: HIP-event bracket around every kernel launch; N: every N-th."""
        check(lib.tbk_prof_enable(self.handle, int(period)))

    def prof_calibrate(self, reps=50):
        """Median duration (ms) of an empty HIP-event bracket on this stream."""
        ms = C.c_double(0.0)
        check(lib.tbk_prof_calibrate(self.handle, int(reps), C.byref(ms)))
        return ms.value

    def prof_reset(self):
        check(lib.tbk_prof_reset(self.handle))

    def prof_report(self):
        n = C.c_int(0)
        check(lib.tbk_prof_count(self.handle, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(128)
            cnt = C.c_int64(0)
            ms = C.c_double(0.0)
            check(lib.tbk_prof_get(self.handle, i, name, 128, C.byref(cnt), C.byref(ms)))
            out[name.value.decode()] = dict(launches=cnt.value, total_ms=ms.value)
        return out


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx
