"""CPU checks of the C-ABI boundary: the library builds/loads here (hipcc cross-compiles gfx950) and
exports every symbol include/rcvpose_hip.h declares; the ctypes binding covers exactly that set.
No compute calls are made (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rcvpose_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rcv_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_binding_and_library_agree():
    from rcvpose_amd import _hip, build

    build.build_library()
    declared = _declared_functions()
    assert len(declared) >= 15
    assert sorted(_hip.SYMBOLS) == declared
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    L = _hip.load_library()
    header = open(os.path.join(ROOT, "include", "rcvpose_hip.h")).read()
    assert L.rcv_version() == _hip.ABI_VERSION == int(re.search(r"#define RCV_ABI_VERSION (\d+)", header).group(1))
    assert L.rcv_strerror(0) == b"ok" and b"empty" in L.rcv_strerror(_hip.RCV_E_EMPTY)
    assert L.rcv_acc_workspace_bytes(15000) > 15000 * 28


def test_struct_layouts_match_header():
    from rcvpose_amd import _hip

    assert ctypes.sizeof(_hip.AccResult) == 96      # 7 doubles + 10 int32
    assert ctypes.sizeof(_hip.Op) == 16 * 4 + 3 * 8
    assert ctypes.sizeof(_hip.IcpResult) == 16 * 8 + 2 * 8 + 4 * 4  # rcv_icp_result
    assert _hip.AccResult.status.offset == 92


def test_abi_check_refuses_another_header():
    """A binding compiled against another header (older version, or an rcv_op without wq_off: 80 instead of 88 bytes) would hand
    rcv_fcn_create an ops array with the wrong stride: rcv_abi_check tells, and _hip.load_library calls it."""
    from rcvpose_amd import _hip

    L = _hip.load_library()
    sizes = (ctypes.sizeof(_hip.Op), ctypes.sizeof(_hip.AccResult), ctypes.sizeof(_hip.Strided), ctypes.sizeof(_hip.IcpResult))
    assert L.rcv_abi_check(_hip.ABI_VERSION, *sizes) == 0
    assert L.rcv_abi_check(100, *sizes) == _hip.RCV_E_INVALID
    assert L.rcv_abi_check(_hip.ABI_VERSION, 80, *sizes[1:]) == _hip.RCV_E_INVALID
    assert L.rcv_abi_check(_hip.ABI_VERSION, sizes[0], 88, *sizes[2:]) == _hip.RCV_E_INVALID


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np

    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd import _hip

    with pytest.raises(_hip.HipLibraryError):
        A.Accumulator_3D(np.zeros((4, 3)), np.ones(4, dtype=np.float32))
    from rcvpose_amd.models.fcnresnet import DenseFCNResNet152

    m = DenseFCNResNet152(3, 2).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 96))
    from rcvpose_amd import registration

    with pytest.raises(_hip.HipLibraryError):
        registration.compute_point_cloud_distance(np.zeros((4, 3)), np.ones((5, 3)))
    with pytest.raises(_hip.HipLibraryError):
        registration.registration_icp(np.zeros((4, 3)), np.ones((5, 3)), 1.0)


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "rcvpose_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), (dirpath, f)
