"""Host-side mirror of the reference's AccumulatorSpace.py for the `--mode=test` hot path.

Same names, argument meaning and error behaviour as the reference callables; the arithmetic runs in
the hand-written HIP kernels of librcvpose_hip.so (include/rcvpose_hip.h).  No CPU fallback.

    rgbd_to_point_cloud(K, depth)            reference AccumulatorSpace.py:77-85
    Accumulator_3D(xyz, radial_list)         reference AccumulatorSpace.py:373-419 (+ fast_for :325-341)
    seam_compact(...)                        the inline seam of estimate_6d_pose_lm, :603-610,619
    FCResBackbone(model, path, _)            reference AccumulatorSpace.py:122-162
"""
import ctypes

import numpy as np

from . import _hip

# reference AccumulatorSpace.py:59-61
linemod_K = np.array([[572.4114, 0., 325.2611],
                      [0., 573.57043, 242.04899],
                      [0., 0., 1.]])

ACC_UNIT_MM = 5.0  # reference AccumulatorSpace.py:374 hard-codes 5 mm voxels
SEG_THRESHOLD = 0.8  # reference AccumulatorSpace.py:603


def _torch():
    import torch

    return torch


def _k4(K):
    K = np.asarray(K, dtype=np.float64)
    return (ctypes.c_double * 4)(K[0, 0], K[1, 1], K[0, 2], K[1, 2])


class AccWorkspace:
    """Device buffers for the voting path, reusable across calls (the C ABI never allocates)."""

    def __init__(self, max_points, device="cuda"):
        torch = _torch()
        L = _hip.lib()
        self.max_points = int(max_points)
        self.bytes = int(L.rcv_acc_workspace_bytes(self.max_points))
        self.ws = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self.result = torch.zeros(ctypes.sizeof(_hip.AccResult), dtype=torch.uint8, device=device)
        self.xyz = torch.empty((self.max_points, 3), dtype=torch.float64, device=device)
        self.radial = torch.empty((self.max_points,), dtype=torch.float32, device=device)
        self.counts = torch.zeros(2, dtype=torch.int32, device=device)

    def read_result(self):
        raw = self.result.cpu().numpy().tobytes()
        return _hip.AccResult.from_buffer_copy(raw)


def seam_compact(seg, radial, depth, max_radius_dm, K=linemod_K, mode=0, workspace=None, seg_thr=SEG_THRESHOLD):
    """The mask / back-projection seam (reference AccumulatorSpace.py:603-610,619) on the device.

    seg, radial: float32 [H,W] CUDA tensors (mode 0) -- depth: uint16 [H,W] (int16 storage accepted).
    Returns (workspace, H*W cap); the compacted lists stay on the device in workspace.xyz /
    workspace.radial with their lengths in workspace.counts (no host synchronisation).
    """
    torch = _torch()
    L = _hip.lib()
    H, W = depth.shape
    if workspace is None:
        workspace = AccWorkspace(H * W, device=depth.device)
    assert workspace.max_points >= H * W
    rc = L.rcv_seam_compact(_hip.ptr(seg), _hip.ptr(radial), _hip.ptr(depth), H, W, _k4(K), float(seg_thr),
                            float(max_radius_dm), int(mode), _hip.ptr(workspace.xyz), _hip.ptr(workspace.radial),
                            _hip.ptr(workspace.counts), workspace.max_points, _hip.ptr(workspace.ws),
                            workspace.bytes, _hip.current_stream())
    _hip.check(rc, "rcv_seam_compact")
    return workspace


def _depth_to_u16(depth, device):
    torch = _torch()
    d = np.ascontiguousarray(depth)
    if d.dtype != np.uint16:
        if d.size and (d.min() < 0 or d.max() > 65535):
            raise ValueError("depth must fit uint16 millimetres")
        d = d.astype(np.uint16)
    return torch.from_numpy(d.view(np.int16)).to(device)


def rgbd_to_point_cloud(K, depth):
    """Pin-hole back-projection of the non-zero depth pixels, row-major (reference :77-85).

    depth: integer array [H,W] (mm) -> float64 [N,3] (mm)."""
    torch = _torch()
    L = _hip.lib()
    d = _depth_to_u16(depth, "cuda")
    H, W = d.shape
    ws = AccWorkspace(H * W)
    rc = L.rcv_seam_compact(None, None, _hip.ptr(d), H, W, _k4(K), 0.0, 0.0, 2, _hip.ptr(ws.xyz), None,
                            _hip.ptr(ws.counts), ws.max_points, _hip.ptr(ws.ws), ws.bytes, _hip.current_stream())
    _hip.check(rc, "rcv_seam_compact(mode=2)")
    n = int(ws.counts[0].item())
    return ws.xyz[:n].cpu().numpy()


def accumulate_on_device(xyz_dev, radial_dev, counts_dev, n_cap, m_cap, workspace, acc_unit=ACC_UNIT_MM,
                         vote_grid=None):
    """Enqueue Accumulator_3D for device-resident lists; result lands in workspace.result."""
    L = _hip.lib()
    torch = _torch()
    is_f64 = radial_dev.dtype == torch.float64
    cells = 0 if vote_grid is None else vote_grid.numel()
    rc = L.rcv_accumulate_3d(_hip.ptr(xyz_dev), _hip.ptr(radial_dev), int(is_f64), _hip.ptr(counts_dev), int(n_cap),
                             int(m_cap), float(acc_unit), _hip.ptr(workspace.result), _hip.ptr(vote_grid), cells,
                             _hip.ptr(workspace.ws), workspace.bytes, _hip.current_stream())
    _hip.check(rc, "rcv_accumulate_3d")


def set_vote_mode(mode):
    """'auto' (default): exact branch-and-bound peak search with the LDS rasteriser as fallback;
    'full': LDS rasteriser only.  Identical results; see rcv_set_vote_mode in include/rcvpose_hip.h."""
    _hip.check(_hip.lib().rcv_set_vote_mode({"auto": 0, "full": 1}[mode]), "rcv_set_vote_mode")


def _raise_like_reference(status):
    if status in (_hip.RCV_E_EMPTY, _hip.RCV_E_BADGRID):
        # the reference fails with numpy's ValueError here (AccumulatorSpace.py:390 / :401,406)
        raise ValueError("zero-size array to reduction operation (Accumulator_3D: %s)" %
                         _hip.load_library().rcv_strerror(status).decode())
    if status != _hip.RCV_OK and status != _hip.RCV_E_GRID_BUFFER:
        raise _hip.HipLibraryError("rcv_accumulate_3d: " + _hip.load_library().rcv_strerror(status).decode())


def Accumulator_3D(xyz, radial_list, acc_unit=ACC_UNIT_MM, return_votes=False):
    """Radial voting + arg-max (reference AccumulatorSpace.py:373-419).

    xyz: [N,3] metres; radial_list: [M>=N] decimetres (float32 or float64).  Returns `center`,
    float64 [ties,3]: every voxel attaining the maximum in C order, row 0 converted to mm in the
    camera frame, the other rows left as shifted voxel indices -- exactly as the reference does.
    Inputs are not modified.  Empty input raises ValueError like the reference.
    """
    torch = _torch()
    _hip.lib()
    xyz_h = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64).reshape(-1, 3))
    rad_h = np.ascontiguousarray(np.asarray(radial_list))
    if rad_h.dtype not in (np.float32, np.float64):
        rad_h = rad_h.astype(np.float64)
    n, m = len(xyz_h), len(rad_h)
    if n == 0 or m == 0:
        _raise_like_reference(_hip.RCV_E_EMPTY)
    ws = AccWorkspace(max(n, m, 1))
    xyz_d = torch.from_numpy(xyz_h).cuda()
    rad_d = torch.from_numpy(rad_h).cuda()
    accumulate_on_device(xyz_d, rad_d, None, n, m, ws, acc_unit)
    res = ws.read_result()
    _raise_like_reference(res.status)
    center = np.array([[res.center_mm[0], res.center_mm[1], res.center_mm[2]]], dtype=np.float64)
    votes = None
    if res.n_ties > 1 or return_votes:
        # rare: several voxels share the maximum (or the caller wants the grid) -> materialise VoteMap_3D
        D = res.dim
        grid = torch.empty(D * D * D, dtype=torch.int32, device="cuda")
        accumulate_on_device(xyz_d, rad_d, None, n, m, ws, acc_unit, vote_grid=grid)
        res = ws.read_result()
        _raise_like_reference(res.status)
        if res.n_ties > 1:
            L = _hip.lib()
            idx = torch.empty(res.n_ties, dtype=torch.int64, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            _hip.check(L.rcv_collect_ties(_hip.ptr(grid), D, res.peak_votes, _hip.ptr(idx), res.n_ties, _hip.ptr(cnt),
                                          _hip.current_stream()), "rcv_collect_ties")
            flat = idx.cpu().numpy()
            assert int(cnt.item()) == res.n_ties
            rows = np.stack([flat // (D * D), (flat // D) % D, flat % D], axis=1).astype(np.float64)
            if res.zero_boundary < 0:
                rows = rows + res.zero_boundary
            rows[0] = center[0]
            center = rows
        if return_votes:
            votes = grid.cpu().numpy().view(np.uint32).reshape(D, D, D)
    if return_votes:
        return center, votes, res
    return center


def FCResBackbone(model, input_img_path, normalized_depth=None):
    """Run one RGB image through a DenseFCNResNet152 (reference AccumulatorSpace.py:122-162).

    Returns (sem_out [H,W], radial_out [H,W]) float32 numpy arrays.  `normalized_depth` is accepted and
    ignored, as in the reference.  The image is normalised on the device with the reference's float64
    arithmetic (rcv_preprocess_rgb) and the forward runs in the native engine."""
    from PIL import Image

    from . import fcn_engine

    torch = _torch()
    _hip.lib()
    rgb = np.array(Image.open(input_img_path).convert("RGB"), dtype=np.uint8)
    x = fcn_engine.preprocess_rgb(torch.from_numpy(rgb).cuda())
    model.to("cuda")  # the reference moves the model on every call (:154-156)
    with torch.no_grad():
        sem_out, radial_out = model(x)
    sem_out, radial_out = sem_out.cpu(), radial_out.cpu()
    return np.asarray(sem_out[0][0]), np.asarray(radial_out[0][0])
