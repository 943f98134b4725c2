"""ctypes binding of librcvpose_hip.so (C ABI: include/rcvpose_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible, every product entry
point raises.  PyTorch is used only as plumbing (device buffers, streams).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCV_LIB_PATH") or os.path.join(_HERE, "librcvpose_hip.so")  # (override: A-B builds of tools/)

RCV_OK = 0
RCV_E_INVALID = -1
RCV_E_HIP = -2
RCV_E_WORKSPACE = -3
RCV_E_EMPTY = -4
RCV_E_BADGRID = -5
RCV_E_TOO_LARGE = -6
RCV_E_GRID_BUFFER = -7
RCV_E_UNSUPPORTED = -8
RCV_E_RANGE = -9

RCV_SEAM_RADIAL_F16 = 256
RCV_SEAM_COUNT_NONFINITE = 512  # counts then holds four int32 per item: [points, radii, non-finite pixels, running maximum of those]
ABI_VERSION = 600  # RCV_ABI_VERSION of include/rcvpose_hip.h
RCV_VOTE_AUTO, RCV_VOTE_RASTER = 0, 1
VOTE_ENGINES = {"auto": RCV_VOTE_AUTO, "full": RCV_VOTE_RASTER}

RCV_F32 = 0
RCV_BF16 = 1
RCV_F16 = 2
RCV_F16X3 = 3

OP_STEM, OP_MAXPOOL, OP_CONV, OP_UPSAMPLE2, OP_HEAD = 1, 2, 3, 4, 5
SLOT_NONE, SLOT_INPUT, SLOT_OUTPUT = -1, -2, -3


class Strided(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("stride_frame", ctypes.c_int64), ("stride_kp", ctypes.c_int64)]


class IcpResult(ctypes.Structure):
    """struct rcv_icp_result (include/rcvpose_hip.h)."""
    _fields_ = [("transformation", ctypes.c_double * 16), ("fitness", ctypes.c_double), ("inlier_rmse", ctypes.c_double),
                ("n_correspondences", ctypes.c_int32), ("iterations", ctypes.c_int32), ("status", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class AccResult(ctypes.Structure):
    _fields_ = [
        ("center_mm", ctypes.c_double * 3),
        ("mean_vox", ctypes.c_double * 3),
        ("radius_max", ctypes.c_double),
        ("peak_votes", ctypes.c_int32),
        ("n_ties", ctypes.c_int32),
        ("first_idx", ctypes.c_int32 * 3),
        ("dim", ctypes.c_int32),
        ("zero_boundary", ctypes.c_int32),
        ("n_points", ctypes.c_int32),
        ("n_radii", ctypes.c_int32),
        ("status", ctypes.c_int32),
    ]


class Op(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("src0", ctypes.c_int32),
        ("src1", ctypes.c_int32),
        ("res", ctypes.c_int32),
        ("dst", ctypes.c_int32),
        ("c0", ctypes.c_int32),
        ("c1", ctypes.c_int32),
        ("cout", ctypes.c_int32),
        ("hin", ctypes.c_int32),
        ("win", ctypes.c_int32),
        ("hout", ctypes.c_int32),
        ("wout", ctypes.c_int32),
        ("ksize", ctypes.c_int32),
        ("stride", ctypes.c_int32),
        ("pad", ctypes.c_int32),
        ("relu", ctypes.c_int32),
        ("w_off", ctypes.c_int64),
        ("b_off", ctypes.c_int64),
        ("wq_off", ctypes.c_int64),
    ]


assert ctypes.sizeof(AccResult) == 96

# every symbol include/rcvpose_hip.h declares: (name, restype, argtypes)
_vp, _i, _d, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_size_t
SYMBOLS = {
    "rcv_strerror": (ctypes.c_char_p, [_i]),
    "rcv_version": (_i, []),
    "rcv_abi_check": (_i, [_i, _sz, _sz, _sz, _sz]),
    "rcv_last_hip_error": (ctypes.c_char_p, []),
    "rcv_init": (_i, []),
    "rcv_mfma_calibrate": (_i, [_d, ctypes.POINTER(_d), ctypes.POINTER(_d), _vp]),
    "rcv_acc_workspace_bytes": (_sz, [_i]),
    "rcv_seam_compact": (_i, [_vp, _vp, _vp, _i, _i, ctypes.POINTER(_d), _f, _d, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "rcv_acc_workspace_bytes_min": (_sz, [_i]),
    "rcv_accumulate_3d": (_i, [_vp, _vp, _i, _vp, _i, _i, _d, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "rcv_backproject_no_depth": (_i, [_vp, _i, _i, ctypes.POINTER(_d), _vp, ctypes.c_longlong, ctypes.POINTER(ctypes.c_int32), _vp, _sz, _vp]),
    "rcv_accumulate_3d_no_depth": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "rcv_horn_pose_batch": (_i, [_vp, Strided, _i, _i, _vp, _vp]),
    "rcv_seam_compact_batch": (_i, [Strided, Strided, Strided, _i, _i, ctypes.POINTER(_d), _f, Strided, _i, Strided, Strided,
                                     Strided, _i, Strided, _sz, _i, _i, _vp]),
    "rcv_accumulate_3d_batch": (_i, [Strided, Strided, _i, Strided, _i, _d, _i, Strided, Strided, _sz, _i, _i, _vp]),
    "rcv_collect_ties": (_i, [_vp, _i, ctypes.c_uint32, _vp, _i, _vp, _vp]),
    "rcv_acc_debug_points": (_i, [_vp, _vp, _vp, _i, _vp]),
    "rcv_preprocess_rgb": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "rcv_fcn_create": (_i, [ctypes.POINTER(Op), _i, _i, ctypes.POINTER(ctypes.c_int64), _i, _i, _vp, _vp,
                             ctypes.POINTER(_vp)]),
    "rcv_fcn_destroy": (None, [_vp]),
    "rcv_fcn_workspace_bytes": (_sz, [_vp, _i]),
    "rcv_fcn_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "rcv_fcn_range_status": (_i, [_vp, ctypes.POINTER(ctypes.c_uint), _i, _vp]),
    "rcv_fcn_debug_slot": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "rcv_fcn_debug_q6_pack": (_i, [_vp, ctypes.c_longlong, _vp]),
    "rcv_fcn_set_profiling": (_i, [_vp, _i]),
    "rcv_fcn_set_option": (_i, [_vp, _i, _i]),
    "rcv_fcn_get_profile": (_i, [_vp, ctypes.POINTER(_d), ctypes.POINTER(_d), ctypes.POINTER(_d)]),
    "rcv_fcn_get_op_times": (_i, [_vp, ctypes.POINTER(_d), _i]),
    "rcv_union_depth": (_i, [_vp, _vp, ctypes.c_longlong, _vp, _i, _i, _f, _vp, _i, _i, _vp, _vp]),
    "rcv_nn_distance": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp]),
    "rcv_icp_workspace_bytes": (_sz, [_i]),
    "rcv_icp_point_to_point": (_i, [_vp, _i, _vp, _i, _d, ctypes.POINTER(_d), _d, _d, _i, _vp, _vp, _sz, _vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


class HalfRangeError(HipLibraryError, FloatingPointError):
    """RCV_E_RANGE: an activation left the half-precision range of an f16x3 / fp16 engine (the maps are clipped, not exact)."""


def load_library():
    """dlopen librcvpose_hip.so and bind every symbol of the header (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so (same soname as /opt/rocm's).  It must be in the process before
    # this library is, so that both share ONE HIP runtime; loaded the other way round the process ends up
    # with two runtimes and the first kernel launch fails with hipErrorNoDevice.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "librcvpose_hip.so is not built (%s). Run `python -m rcvpose_amd.build` (needs hipcc); "
            "there is no CPU fallback for the product path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:  # header and library out of sync: refuse to run anything
        raise HipLibraryError("librcvpose_hip.so lacks symbols declared in include/rcvpose_hip.h: " + ", ".join(missing))
    # same header on both sides: version and the struct layouts this binding declares (an ops array with another stride would be
    # read as a garbage program)
    if lib.rcv_abi_check(ABI_VERSION, ctypes.sizeof(Op), ctypes.sizeof(AccResult), ctypes.sizeof(Strided), ctypes.sizeof(IcpResult)) != 0:
        raise HipLibraryError("librcvpose_hip.so (ABI %d) does not match this binding (ABI %d, sizeof rcv_op %d): rebuild with "
                              "`python -m rcvpose_amd.build --force`" % (lib.rcv_version(), ABI_VERSION, ctypes.sizeof(Op)))
    _lib = lib
    return lib


def lib():
    """The bound library, for product calls: additionally requires a visible GPU."""
    import torch

    L = load_library()
    if not torch.cuda.is_available():
        raise HipLibraryError("no HIP device visible: the rcvpose_amd product path runs on MI355X only "
                              "(no CPU fallback)")
    return L


def check(rc, what=""):
    if rc != RCV_OK:
        L = load_library()
        msg = L.rcv_strerror(rc).decode()
        if rc == RCV_E_HIP:
            msg += ": " + L.rcv_last_hip_error().decode()
        if rc == RCV_E_RANGE:
            raise HalfRangeError("%s failed: %s (%d)" % (what or "librcvpose_hip call", msg, rc))
        raise HipLibraryError("%s failed: %s (%d)" % (what or "librcvpose_hip call", msg, rc))


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
