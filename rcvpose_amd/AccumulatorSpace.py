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

# reference AccumulatorSpace.py:19-57
lm_cls_names = ['ape', 'benchvise', 'cam', 'can', 'cat', 'duck', 'driller', 'eggbox', 'glue', 'holepuncher', 'iron', 'lamp',
                'phone']
lmo_cls_names = ['ape', 'can', 'cat', 'duck', 'driller', 'eggbox', 'glue', 'holepuncher']
ycb_cls_names = {1: '002_master_chef_can', 2: '003_cracker_box', 3: '004_sugar_box', 4: '005_tomato_soup_can', 5: '006_mustard_bottle',
                 6: '007_tuna_fish_can', 7: '008_pudding_box', 8: '009_gelatin_box', 9: '010_potted_meat_can', 10: '011_banana',
                 11: '019_pitcher_base', 12: '021_bleach_cleanser', 13: '024_bowl', 14: '025_mug', 15: '035_power_drill',
                 16: '036_wood_block', 17: '037_scissors', 18: '040_large_marker', 19: '051_large_clamp', 20: '052_extra_large_clamp',
                 21: '061_foam_brick'}
lm_syms = ['eggbox', 'glue']
ycb_syms = ['024_bowl', '036_wood_block', '051_large_clamp', '052_extra_large_clamp', '061_foam_brick']
add_threshold = {
    'eggbox': 0.019735770122546523, 'ape': 0.01421240983190395, 'cat': 0.018594838977253875, 'cam': 0.02222763033276377,
    'duck': 0.015569664208967385, 'glue': 0.01930723067998101, 'can': 0.028415044264086586, 'driller': 0.031877906042,
    'holepuncher': 0.019606109985, 'benchvise': .033091264970068, 'iron': .03172344425531, 'lamp': .03165980764376,
    'phone': .02543407135792}

DEFAULT_FRAMES_PER_STEP = 53  # frames per device batch of the evaluation drivers (bench.py's optimum for the f16x3 forward)


def default_eval_threads():
    """Host threads for the per-frame Horn / ADD(-S) / ICP evaluation: 16 overlap a batch's chains of small launches on a GPU box
    (128 hardware threads there); never more than the host has."""
    import os

    return max(1, min(16, os.cpu_count() or 1))
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
                         vote_grid=None, engine="auto"):
    """Enqueue Accumulator_3D for device-resident lists; result lands in workspace.result.
    engine: 'auto' = exact branch-and-bound peak search with the LDS rasteriser as fallback, 'full' = LDS rasteriser
    only (identical results; RCV_VOTE_* in include/rcvpose_hip.h)."""
    L = _hip.lib()
    torch = _torch()
    is_f64 = radial_dev.dtype == torch.float64
    cells = 0 if vote_grid is None else vote_grid.numel()
    rc = L.rcv_accumulate_3d(_hip.ptr(xyz_dev), _hip.ptr(radial_dev), int(is_f64), _hip.ptr(counts_dev), int(n_cap),
                             int(m_cap), float(acc_unit), _hip.VOTE_ENGINES[engine], _hip.ptr(workspace.result), _hip.ptr(vote_grid),
                             cells,
                             _hip.ptr(workspace.ws), workspace.bytes, _hip.current_stream())
    _hip.check(rc, "rcv_accumulate_3d")


def _raise_like_reference(status):
    if status in (_hip.RCV_E_EMPTY, _hip.RCV_E_BADGRID):
        # the reference fails with numpy's ValueError here (AccumulatorSpace.py:390 / :401,406)
        raise ValueError("zero-size array to reduction operation (Accumulator_3D: %s)" %
                         _hip.load_library().rcv_strerror(status).decode())
    if status != _hip.RCV_OK and status != _hip.RCV_E_GRID_BUFFER:
        raise _hip.HipLibraryError("rcv_accumulate_3d: " + _hip.load_library().rcv_strerror(status).decode())


def Accumulator_3D(xyz, radial_list, acc_unit=ACC_UNIT_MM, return_votes=False, engine="auto"):
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
    accumulate_on_device(xyz_d, rad_d, None, n, m, ws, acc_unit, engine=engine)
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


def rgbd_to_point_cloud_no_depth(K, depth):
    """Back-projection of every nonzero pixel at each millimetre of the depth range (reference :100-120).
    depth: float array [H,W] in metres -> float64 [Z,N,3].

    The range arithmetic (`int(zs_max*1000)+1-int(zs_min*1000)`, `zs_min+i*0.001`) runs in float64 on the values of the
    input array.  For float64 depth that is what every numpy does.  For float32 depth it is what the numpy of the reference's
    environment does (rcvpose.yml pins numpy 1.20.3, whose scalar promotion turns np.float32 * 1000 and np.float32 + 0.001 into
    float64); numpy >= 2 (NEP 50) would keep those two expressions in float32 and can differ by one hypothesis level."""
    torch = _torch()
    L = _hip.lib()
    d = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float64)).cuda()
    H, W = d.shape
    ws = torch.empty(256 + 4 * H * W, dtype=torch.uint8, device="cuda")
    zn = (ctypes.c_int32 * 2)()
    rc = L.rcv_backproject_no_depth(_hip.ptr(d), H, W, _k4(K), None, 0, zn, _hip.ptr(ws), ws.numel(), _hip.current_stream())
    if rc == _hip.RCV_E_EMPTY:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # zs.min(), reference :103
    _hip.check(rc, "rcv_backproject_no_depth")
    Z, N = int(zn[0]), int(zn[1])
    pts = torch.empty((Z, N, 3), dtype=torch.float64, device="cuda")
    _hip.check(L.rcv_backproject_no_depth(_hip.ptr(d), H, W, _k4(K), _hip.ptr(pts), Z * N, zn, _hip.ptr(ws), ws.numel(),
                                          _hip.current_stream()), "rcv_backproject_no_depth")
    print((Z, N, 3))  # the reference prints pts.shape (:118)
    return pts.cpu().numpy()


def Accumulator_3D_no_depth(xyz, radial_list, pixel_coor=None, engine="auto", return_votes=False):
    """The RGB-only accumulator (reference AccumulatorSpace.py:436-479, fast_for_no_depth :421-434): xyz [Z,N,3] metres (the
    hypotheses of rgbd_to_point_cloud_no_depth), radial_list [N] decimetres.  Returns `center`, int64 [ties,3], computed
    exactly as the reference computes it (see rcv_accumulate_3d_no_depth in include/rcvpose_hip.h).  pixel_coor is
    accepted and ignored, as in the reference.  Fewer than three pixels raise IndexError like the reference."""
    torch = _torch()
    L = _hip.lib()
    xyz_h = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64))
    if xyz_h.ndim != 3 or xyz_h.shape[2] != 3 or xyz_h.shape[1] < 3:
        raise IndexError("index 2 is out of bounds for axis 1 (Accumulator_3D_no_depth needs a [Z, N>=3, 3] array)")
    rad_h = np.ascontiguousarray(np.asarray(radial_list))
    if rad_h.dtype not in (np.float32, np.float64):
        rad_h = rad_h.astype(np.float64)
    Z, N = xyz_h.shape[:2]
    if len(rad_h) < N:
        raise IndexError("radial_list shorter than the pixel list")
    ws = AccWorkspace(Z * N)
    xyz_d, rad_d = torch.from_numpy(xyz_h).cuda(), torch.from_numpy(rad_h).cuda()

    def run(grid):
        cells = 0 if grid is None else grid.numel()
        _hip.check(L.rcv_accumulate_3d_no_depth(_hip.ptr(xyz_d), _hip.ptr(rad_d), int(rad_h.dtype == np.float64), Z, N, len(rad_h),
                                                _hip.VOTE_ENGINES[engine], _hip.ptr(ws.result), _hip.ptr(grid), cells, _hip.ptr(ws.ws),
                                                ws.bytes, _hip.current_stream()), "rcv_accumulate_3d_no_depth")
        res = ws.read_result()
        _raise_like_reference(res.status)
        return res

    res = run(None)
    center = np.array([[res.center_mm[0], res.center_mm[1], res.center_mm[2]]]).astype(np.int64)
    votes = None
    if res.n_ties > 1 or return_votes:
        D = res.dim
        grid = torch.empty(D * D * D, dtype=torch.int32, device="cuda")
        res = run(grid)
        if res.n_ties > 1:
            idx = torch.empty(res.n_ties, dtype=torch.int64, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            _hip.check(L.rcv_collect_ties(_hip.ptr(grid), D, res.peak_votes, _hip.ptr(idx), res.n_ties, _hip.ptr(cnt),
                                          _hip.current_stream()), "rcv_collect_ties")
            flat = idx.cpu().numpy()
            rows = np.stack([flat // (D * D), (flat // D) % D, flat % D], axis=1).astype(np.int64)
            if res.zero_boundary < 0:
                rows = rows + res.zero_boundary
            rows = rows * 5
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


# ---------------------------------------------------------------------------------------------------
# Dataset ingest + the evaluation drivers: reference AccumulatorSpace.py:64-75, :482-490,
# :495-740 (LINEMOD), :742-974 (Occlusion LINEMOD), :976-1197 (YCB-Video), :1202-1227 (command line).  The hot loop runs in
# rcvpose_amd.pipeline.FramePipeline; the Horn fit (f-1), ADD(-S) and the ICP refinement (f-3, rcvpose_amd.registration)
# on the device.
# ---------------------------------------------------------------------------------------------------
def project(xyz, K, RT):
    """Rigid transform followed by the pin-hole projection (the role of reference AccumulatorSpace.py:64-75).

    xyz [N,3], K [3,3], RT [3,4] -> (pixel coordinates [N,2], the points in the camera frame [N,3])."""
    R, t = RT[:, :3], RT[:, 3]
    cam = xyz @ R.T + t[None, :]
    uvw = cam @ K.T
    return uvw[:, :2] / uvw[:, 2:3], cam


def read_depth(path):
    """Depth map of a frame (the role of reference AccumulatorSpace.py:482-490).  A `.dpt` file is a little header of two
    uint32 (rows, columns) followed by rows*columns uint16 millimetres; any other suffix is an image PIL can open."""
    if not path.endswith('dpt'):
        from PIL import Image

        return np.array(Image.open(path))
    raw = np.fromfile(path, dtype=np.uint8)
    rows, cols = (int(v) for v in raw[:8].view(np.uint32))
    return raw[8:8 + 2 * rows * cols].view(np.uint16).reshape(rows, cols).copy()


def read_ply_points(path):
    """Vertex positions of a PLY model (ascii or binary_little_endian), the part of
    open3d.io.read_point_cloud(...).points the reference uses (AccumulatorSpace.py:507,536)."""
    with open(path, 'rb') as f:
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode('ascii', 'replace').strip()
            if line.startswith('format'):
                fmt = line.split()[1]
            elif line.startswith('element'):
                in_vertex = line.split()[1] == 'vertex'
                if in_vertex:
                    n = int(line.split()[2])
            elif line.startswith('property') and in_vertex:
                props.append((line.split()[1], line.split()[2]))
            elif line == 'end_header':
                break
        names = [p[1] for p in props]
        ix = [names.index(a) for a in ('x', 'y', 'z')]
        if fmt == 'ascii':
            rows = np.loadtxt(f, max_rows=n, ndmin=2)
            return rows[:, ix].astype(np.float64)
        types = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'float64': '<f8', 'uchar': 'u1', 'uint8': 'u1', 'char': 'i1',
                 'short': '<i2', 'ushort': '<u2', 'int': '<i4', 'int32': '<i4', 'uint': '<u4'}
        dt = np.dtype([(nm, types[t]) for t, nm in props])
        data = np.fromfile(f, dtype=dt, count=n)
        return np.stack([data['x'], data['y'], data['z']], axis=1).astype(np.float64)


def scene_cloud_of_keypoint_masks(seg_maps, radial_maps, depth_dev, max_radii_dm, K=linemod_K, mode=0, seg_thr=SEG_THRESHOLD):
    """`xyz_mm_icp` of the reference (AccumulatorSpace.py:621-626): the back-projected pixels that belong to the point
    mask of at least one keypoint, as a float64 [n,3] CUDA tensor in mm (row-major pixel order).

    seg_maps / radial_maps: float32 [n_kp,H,W] CUDA tensors (seg_maps None for the pre-computed radial-map branch, mode 1)."""
    torch = _torch()
    L = _hip.lib()
    n_kp, H, W = radial_maps.shape
    radial_maps = radial_maps.contiguous()
    if seg_maps is not None:
        seg_maps = seg_maps.contiguous()
    rmax = torch.from_numpy(np.ascontiguousarray(max_radii_dm, dtype=np.float64)).to(radial_maps.device)
    union = torch.empty((H, W), dtype=torch.int16, device=radial_maps.device)
    _hip.check(L.rcv_union_depth(_hip.ptr(seg_maps), _hip.ptr(radial_maps), H * W, _hip.ptr(depth_dev), H, W, float(seg_thr),
                                 _hip.ptr(rmax), int(mode), n_kp, _hip.ptr(union), _hip.current_stream()), "rcv_union_depth")
    ws = AccWorkspace(H * W, device=radial_maps.device)
    _hip.check(L.rcv_seam_compact(None, None, _hip.ptr(union), H, W, _k4(K), 0.0, 0.0, 2, _hip.ptr(ws.xyz), None, _hip.ptr(ws.counts),
                                  ws.max_points, _hip.ptr(ws.ws), ws.bytes, _hip.current_stream()), "rcv_seam_compact(mode=2)")
    n = int(ws.counts[0].item())
    return ws.xyz[:n].clone()


def _pose_and_add(class_name, xyz_load, keypoints, estimated_kpts, RTGT_mm, seg_maps, rad_maps, depth_dev, max_radii_dm,
                  mask_mode, criteria, seg_thr=SEG_THRESHOLD, K=linemod_K, syms=lm_syms, scene_scale=1.0, reach_is_mean=False, RT=None):
    """Horn pose from the three keypoints, ADD(-S) before ICP, ICP of the CAD model onto the masked scene points and
    ADD(-S) after ICP (reference AccumulatorSpace.py:655-726 / :899-958 / :1115-1190), everything on the GPU.

    scene_scale: millimetres per depth unit (1 for LINEMOD; 1000 / factor_depth for YCB-Video).  reach_is_mean: the
    YCB loop hands the mean distance to ICP even for symmetric classes (:1160), the LINEMOD loops the metric itself."""
    torch = _torch()
    from .registration import compute_point_cloud_distance, registration_icp
    from .util.horn import horn_batch

    kpts = keypoints[1:4, :] * 1000
    if RT is None:  # (the batched LINEMOD loop fits the poses of a whole batch in one rcv_horn_pose_batch call)
        RT = horn_batch(kpts, torch.from_numpy(np.ascontiguousarray(estimated_kpts[None])).cuda())[0].cpu().numpy()  # :659-661
    _, est_model = project(xyz_load * 1000, linemod_K, RT[0:3, :])
    _, gt_model = project(xyz_load * 1000, linemod_K, RTGT_mm)
    gt_dev = torch.from_numpy(np.ascontiguousarray(gt_model)).cuda()
    d = compute_point_cloud_distance(gt_dev, est_model)
    dist = d.min() if class_name in syms else d.mean()  # reference :686-694
    # ICP of the CAD model onto the masked scene points, started from the Horn pose, reach = the distance above
    # (reference :696-712), then the same metric on the refined model (:717-726)
    scene = scene_cloud_of_keypoint_masks(seg_maps, rad_maps, depth_dev, max_radii_dm, K, mask_mode, seg_thr)
    if scene_scale != 1.0:
        scene = scene * float(scene_scale)
    reg = registration_icp(xyz_load * 1000, scene, float(d.mean() if reach_is_mean else dist), RT, criteria)
    T = reg.transformation
    cad_model = (xyz_load * 1000) @ T[:3, :3].T + T[:3, 3]
    d2 = compute_point_cloud_distance(gt_dev, cad_model)
    dist_after = d2.min() if class_name in syms else d2.mean()
    return dict(estimated_kpts=estimated_kpts.copy(), RT=RT, add_distance_mm=float(dist), scene_points=int(scene.shape[0]),
                icp_transformation=T, icp_fitness=reg.fitness, icp_inlier_rmse=reg.inlier_rmse, icp_iterations=reg.iterations,
                add_distance_after_icp_mm=float(dist_after))


def _stage_timers(pipe):
    """The reference's stage timers (AccumulatorSpace.py:510-511, 594-599, 627-630: `net_time` around FCResBackbone,
    `acc_time` around Accumulator_3D, summed over the frames of a class) as device times from HIP events (ms)."""
    if pipe is None:
        return {}
    t = pipe.stage_times()
    return dict(net_time_ms=t.get("net", 0.0), seam_time_ms=t.get("seam", 0.0), acc_time_ms=t.get("acc", 0.0))


def _missing_class_dir(opts, path, missing):
    """A class directory that is not there.  The reference opens the class's files unconditionally and dies with
    FileNotFoundError (AccumulatorSpace.py:503-507); a class the caller named explicitly does the same here.  With the
    default class list a dataset that holds only some classes is evaluated on those -- but never on none (see
    _require_some_class), so a mistyped --root_dataset cannot look like a successful run."""
    if getattr(opts, 'class_names', None):
        raise FileNotFoundError(path)
    missing.append(path)


def _require_some_class(results, missing):
    if not results and missing:
        raise FileNotFoundError("no class directory found and no frame evaluated; looked for e.g. " + missing[0] +
                                " (root_dataset must end with a path separator, as in the reference)")


def estimate_6d_pose_lm(opts, return_details=False):
    """The LINEMOD evaluation loop of the reference (AccumulatorSpace.py:495-740) on the MI355X hot path.

    opts attributes read (as in the reference): root_dataset, model_dir, using_ckpts, demo_mode (ignored), plus the
    optional class_names (default: every LINEMOD class present under root_dataset).  Per frame: 3 keypoints (FCN ->
    seam -> voting, or pre-computed radial maps when using_ckpts is False) -> Horn pose -> ADD(-S) before ICP ->
    point-to-point ICP of the CAD model onto the masked scene points -> ADD(-S) after ICP (all on the GPU).
    """
    torch = _torch()
    from . import utils
    from .models.fcnresnet import DenseFCNResNet152
    from .pipeline import FramePipeline
    from .registration import ICPConvergenceCriteria

    _hip.lib()
    results = {}
    details = {}
    names = getattr(opts, 'class_names', None) or lm_cls_names
    import os

    missing = []
    for class_name in names:
        rootPath = opts.root_dataset + "LINEMOD_ORIG/" + class_name + "/"
        rootpvPath = opts.root_dataset + "LINEMOD/" + class_name + "/"
        if not os.path.isdir(rootpvPath):
            _missing_class_dir(opts, rootpvPath, missing)
            continue
        print("Evaluation on ", class_name)
        test_list = [s.replace('\n', '') for s in open(rootpvPath + "Split/val.txt", "r").readlines()]
        xyz_load = read_ply_points(rootpvPath + class_name + ".ply")
        keypoints = np.load(rootpvPath + "Outside9.npy")
        max_radii_dm = np.zeros(3)
        for i in range(3):  # reference :541-548
            max_radii_dm[i] = (((xyz_load - keypoints[i + 1]) ** 2).sum(axis=1) ** 0.5).max() * 10
        pipe = None
        if opts.using_ckpts:
            sds = []
            for i in range(1, 4):
                model_path = opts.model_dir + class_name + "_pt" + str(i) + ".pth.tar"
                if not os.path.exists(model_path):
                    raise ValueError(model_path + " not found")  # reference :571-573
                model = DenseFCNResNet152(3, 2)
                model, _, _, _ = utils.load_checkpoint(model, None, model_path)
                sds.append(model.state_dict())
        bf_icp, af_icp, general_counter = 0, 0, 0
        per_frame = {}
        dataPath = rootpvPath + 'JPEGImages/'
        todo = []
        for filename in sorted(os.listdir(dataPath)):
            stem = os.path.splitext(filename)[0]
            if filename.endswith(".jpg") and stem in test_list:
                todo.append((filename, stem))

        def account(stem, ev):
            nonlocal bf_icp, af_icp, general_counter
            if ev["add_distance_mm"] <= add_threshold[class_name] * 1000:
                bf_icp += 1
            if ev["add_distance_after_icp_mm"] <= add_threshold[class_name] * 1000:
                af_icp += 1
            general_counter += 1
            per_frame[stem] = ev
            print('Current ADD\\(s\\) of ' + class_name + ' before ICP: ', bf_icp / general_counter)
            print('Currnet ADD\\(s\\) of ' + class_name + ' after ICP: ', af_icp / general_counter)

        import time as _time

        if opts.using_ckpts and todo:  # the engine (BN folding, weight packing, upload) is built once per class, before the frame loop
            H, W = read_depth(rootPath + 'data/depth' + str(int(todo[0][1])) + '.dpt').shape
            n_step = max(1, min(int(getattr(opts, 'frames_per_step', DEFAULT_FRAMES_PER_STEP)), len(todo)))
            # equal batches: the device pipeline runs whole batches, so 128 frames travel as 43 + 43 + 42, not as 53 + 53 + 22 with
            # 31 padded frames of wasted work
            # (the default of 53 is tuned for one MI355X with 288 GB: on a smaller or shared device the workspaces of a batch may not
            # fit -- halve the batch until they do; results do not depend on it)
            # The engine is built ONCE; the batch is then sized from the free device memory and the byte counts of a batch's
            # workspaces (nothing is allocated to find out), and an allocation that fails all the same -- another process took the
            # memory in between -- is retried with half the batch OUTSIDE the except block (inside it the traceback still holds the
            # failed pipeline's tensors, so nothing could be released), whether torch's allocator or a hipMalloc of the library refused.
            from .fcn_engine import FcnEngine
            import gc

            dtype_ = getattr(opts, 'dtype', 'f16x3')
            fcn = FcnEngine(sds, H, W, dtype=dtype_)
            pipe = None
            while pipe is None:
                n_batches = (len(todo) + n_step - 1) // n_step
                n_step = (len(todo) + n_batches - 1) // n_batches
                free_b = _torch().cuda.mem_get_info()[0] + _torch().cuda.memory_reserved() - _torch().cuda.memory_allocated()
                need_b = FramePipeline.device_bytes(fcn, n_step, H, W, len(sds))
                failed = need_b > 0.95 * free_b
                if not failed:
                    try:
                        pipe = FramePipeline(sds, H, W, dtype=dtype_, frames_per_step=n_step, fcn=fcn)
                    except (_torch().cuda.OutOfMemoryError, _hip.HipLibraryError) as e:
                        if isinstance(e, _hip.HipLibraryError) and 'memory' not in str(e).lower():
                            raise
                        failed = True
                if failed:
                    if n_step == 1:
                        raise MemoryError('the workspaces of one frame (%.1f GB) do not fit the device (%.1f GB free)' % (need_b / 1e9, free_b / 1e9))
                    gc.collect()
                    _torch().cuda.empty_cache()
                    n_step = max(1, n_step // 2)
                    print('frames_per_step reduced to %d (device memory)' % n_step)
            pipe.set_stage_timing(True)
        t_loop = _time.perf_counter()
        if opts.using_ckpts and todo:
            # The frames of a class travel through the device pipeline in batches of opts.frames_per_step (default: bench.py's 53):
            # a pool of host threads decodes the next batch's JPEG / .dpt / pose files while the GPU works on the current one, there
            # is one result copy and one Horn launch per batch; ADD(-S) / ICP of the batch's frames then run on opts.eval_threads
            # host threads, each on its own HIP stream (a frame's evaluation is a chain of small launches and a few
            # synchronisations: one thread spends it waiting, sixteen overlap).  A frame's maps and keypoints do not
            # depend on the batch it travels in (tests/test_gpu_fcn_x3.py::test_x3_groups_and_batch), so the results are those
            # of the frame-by-frame loop of the reference (:553-659).
            from concurrent.futures import ThreadPoolExecutor

            from PIL import Image

            from .util.horn import horn_batch

            def load(item):
                filename, stem = item
                rgb = np.array(Image.open(dataPath + filename).convert('RGB'), dtype=np.uint8)
                depth = read_depth(rootPath + 'data/depth' + str(int(stem)) + '.dpt')
                return rgb, _depth_to_u16(depth, "cpu"), np.load(rootpvPath + "pose/pose" + str(int(stem)) + '.npy')

            N = n_step
            kpts_model = keypoints[1:4, :] * 1000
            import threading

            n_eval = max(1, int(getattr(opts, 'eval_threads', default_eval_threads())))  # (main.py's default)
            tls = threading.local()

            def worker_stream():
                if getattr(tls, "stream", None) is None:
                    tls.stream = torch.cuda.Stream()
                return tls.stream

            with ThreadPoolExecutor(max_workers=min(16, N)) as pool, ThreadPoolExecutor(max_workers=n_eval) as eval_pool:
                chunks = [todo[i:i + N] for i in range(0, len(todo), N)]
                pending = [pool.submit(load, it) for it in chunks[0]]
                # Batch ci + 1 is enqueued on the pipeline's own stream (FCN -> seam -> voting -> async copy of the result records
                # into pinned memory) BEFORE the host turns to ADD(-S) / ICP of batch ci on the default stream: the two overlap.  The
                # FCN outputs are double buffered inside the pipeline; enqueue_step makes the pipeline wait for everything enqueued
                # on the default stream so far, i.e. batch ci + 2 cannot overwrite what the ICP of batch ci still reads.
                rec = ctypes.sizeof(_hip.AccResult)
                host_res = [torch.empty((N * 3, rec), dtype=torch.uint8).pin_memory() for _ in range(2)]
                inflight = {}

                def submit(ci):
                    nonlocal pending
                    loaded = [f.result() for f in pending]
                    pending = [pool.submit(load, it) for it in chunks[ci + 1]] if ci + 1 < len(chunks) else []
                    pad = [loaded[-1]] * (N - len(loaded))  # a short last batch is filled with copies of its last frame
                    rgb = torch.from_numpy(np.stack([l[0] for l in loaded + pad])).cuda()
                    depth_dev = torch.stack([l[1] for l in loaded + pad]).cuda()
                    buf = ci & 1
                    done = torch.cuda.Event()

                    def collect(_kp):  # on the pipeline's voting stream, right after the arg-max
                        host_res[buf].copy_(pipe.results, non_blocking=True)
                        done.record()

                    pipe.enqueue_step(rgb, depth_dev, max_radii_dm, collect=collect)
                    inflight[ci] = (loaded, rgb, depth_dev, pipe.seg, pipe.radial, done)

                submit(0)
                for ci, chunk in enumerate(chunks):
                    if ci + 1 < len(chunks):
                        submit(ci + 1)
                    loaded, _rgb, depth_dev, seg_b, rad_b, done = inflight.pop(ci)
                    done.synchronize()                         # one result copy per batch
                    # half-precision engines: a stored activation beyond 65504 makes the maps of that forward invalid -- loud, not clipped
                    pipe.engine.range_status(clear=True, raise_on_flag=True)
                    torch.cuda.current_stream().wait_event(done)
                    raw = host_res[ci & 1].numpy()
                    res = [_hip.AccResult.from_buffer_copy(raw[i].tobytes()) for i in range(raw.shape[0])]
                    kp_all = np.array([[r.center_mm[0], r.center_mm[1], r.center_mm[2]] for r in res]).reshape(N, 3, 3)
                    RT_all = horn_batch(kpts_model, torch.from_numpy(np.ascontiguousarray(kp_all)).cuda()).cpu().numpy()  # :659-661

                    def evaluate(i):  # Horn pose -> ADD(-S) -> ICP -> ADD(-S) of frame i of the batch, on this thread's stream
                        for r in res[3 * i:3 * i + 3]:
                            _raise_like_reference(r.status)
                        RTGT_mm = np.array(loaded[i][2], dtype=np.float64, copy=True)
                        RTGT_mm[:, 3] = RTGT_mm[:, 3] * 1000
                        with torch.cuda.stream(worker_stream()):
                            torch.cuda.current_stream().wait_event(done)  # (maps, depth and keypoints of the batch are complete)
                            ev = _pose_and_add(class_name, xyz_load, keypoints, kp_all[i], RTGT_mm, seg_b[:, i, 0], rad_b[:, i, 0],
                                               depth_dev[i], max_radii_dm, 0, ICPConvergenceCriteria(), RT=RT_all[i])
                            torch.cuda.current_stream().synchronize()
                        return ev

                    # (every frame of the batch is joined here, before the next iteration lets batch ci + 2 overwrite these maps)
                    futures = [eval_pool.submit(evaluate, i) for i in range(len(chunk))]
                    for (filename, stem), fut in zip(chunk, futures):
                        print("Evaluating ", filename)
                        account(stem, fut.result())
                pipe.finish()
            todo = []
        for filename, stem in todo:  # pre-computed radial maps (:611-618): frame by frame
            print("Evaluating ", filename)
            RTGT = np.load(rootpvPath + "pose/pose" + str(int(stem)) + '.npy')
            depth_map1 = read_depth(rootPath + 'data/depth' + str(int(stem)) + '.dpt')
            H, W = depth_map1.shape
            depth_dev = _depth_to_u16(depth_map1, "cuda")
            estimated_kpts, statuses = np.zeros((3, 3)), []
            ws = AccWorkspace(H * W)
            seg_maps, rad_maps, mask_mode = None, torch.empty((3, H, W), dtype=torch.float32, device="cuda"), 1
            for k in range(1, 4):  # reference :611-618
                radial_est = np.load(os.path.join(opts.root_dataset + "LINEMOD_ORIG/", 'estRadialMap', class_name,
                                                  "Out_pt" + str(k) + "_dm", stem + '.npy'))
                rad_dev = torch.from_numpy(np.ascontiguousarray(radial_est, dtype=np.float32)).cuda()
                rad_maps[k - 1].copy_(rad_dev)
                seam_compact(None, rad_dev, depth_dev, max_radii_dm[k - 1], mode=1, workspace=ws)
                accumulate_on_device(ws.xyz, ws.radial, ws.counts, ws.max_points, ws.max_points, ws)
                r = ws.read_result()
                statuses.append(r.status)
                estimated_kpts[k - 1] = r.center_mm[:]
            for st in statuses:
                _raise_like_reference(st)
            RTGT_mm = np.array(RTGT, dtype=np.float64, copy=True)
            RTGT_mm[:, 3] = RTGT_mm[:, 3] * 1000
            ev = _pose_and_add(class_name, xyz_load, keypoints, estimated_kpts, RTGT_mm, seg_maps, rad_maps, depth_dev, max_radii_dm,
                               mask_mode, ICPConvergenceCriteria())
            account(stem, ev)
        if general_counter:
            tag = 'ADDs' if class_name in lm_syms else 'ADD'
            print(tag + ' of ' + class_name + ' before ICP: ', bf_icp / general_counter)
            print(tag + ' of ' + class_name + ' after ICP: ', af_icp / general_counter)
            results[class_name] = dict(frames=general_counter, add_before_icp=bf_icp / general_counter,
                                       add_after_icp=af_icp / general_counter, frame_loop_s=_time.perf_counter() - t_loop,
                                       **_stage_timers(pipe))
            details[class_name] = per_frame
    _require_some_class(results, missing)
    return (results, details) if return_details else results


def estimate_6d_pose_lmo(opts, return_details=False):
    """The Occlusion-LINEMOD evaluation loop of the reference (AccumulatorSpace.py:742-974) on the MI355X hot path
    (SURVEY.md section 8 f-4): same kernels as estimate_6d_pose_lm, the LM-O file layout and its own thresholds.

    Layout read (as in the reference): `root_dataset/OCCLUSION_LINEMOD/RGB-D/{rgb_noseg/color_NNNNN.png,
    depth_noseg/depth_NNNNN.png}`, `.../blender_poses/<class>/pose<N>.npy`, and, when using_ckpts is False,
    `.../estRadialMap/<class>/Out_pt<k>_dm/_NNNNN.npy`; CAD model and `Outside9.npy` from `root_dataset/LINEMOD/<class>/`.
    Differences from the LINEMOD loop that are kept: the point mask is `seg >= 0.5` (:836) resp. `0 < radial <= max_radius`
    (:848-849), the radius list is gathered over the mask and the points over mask-and-depth (:858-860, so lists differ in
    length where the mask covers a depth hole), a frame without pose file (or radial map) is skipped but still counted
    in the denominator (:963), ICP runs with `ICPConvergenceCriteria(add_threshold*1000, add_threshold*1000, 30)` (:935).
    One deliberate deviation: with using_ckpts=True the reference compares the radial map with the whole 3-vector
    `max_radii_dm` (:837), which numpy cannot broadcast (the branch raises); this build applies the keypoint's own
    entry, as the LINEMOD loop does.
    """
    torch = _torch()
    import os

    from PIL import Image

    from . import utils
    from .models.fcnresnet import DenseFCNResNet152
    from .pipeline import FramePipeline
    from .registration import ICPConvergenceCriteria

    _hip.lib()
    results, details = {}, {}
    names = getattr(opts, 'class_names', None) or lmo_cls_names
    rootPath = opts.root_dataset + 'OCCLUSION_LINEMOD/'
    seg_thr = float(np.nextafter(np.float32(0.5), np.float32(0.0)))  # seg > thr  <=>  seg >= 0.5 in float32
    missing = []
    for class_name in names:
        print(class_name)
        modelPath = opts.root_dataset + "LINEMOD/" + class_name + "/"
        if not os.path.isdir(modelPath):
            _missing_class_dir(opts, modelPath, missing)
            continue
        pipe, sds = None, []
        if opts.using_ckpts:
            for i in range(1, 4):
                model = DenseFCNResNet152(3, 2)
                model, _, _, _ = utils.load_checkpoint(model, None, opts.model_dir + class_name + "_pt" + str(i) + ".pth.tar")
                sds.append(model.state_dict())
        xyz_load = read_ply_points(modelPath + class_name + ".ply")
        keypoints = np.load(modelPath + "Outside9.npy")
        max_radii_dm = np.zeros(3)
        for i in range(3):  # reference :776-781
            max_radii_dm[i] = (((xyz_load - keypoints[i + 1]) ** 2).sum(axis=1) ** 0.5).max() * 10
        jpgPath, depthPath = rootPath + "RGB-D/rgb_noseg/", rootPath + "RGB-D/depth_noseg/"
        criteria = ICPConvergenceCriteria(add_threshold[class_name] * 1000, add_threshold[class_name] * 1000, 30)
        bf_icp, af_icp, general_counter = 0, 0, 0
        per_frame = {}
        for filename in sorted(os.listdir(jpgPath)):
            if not filename.endswith(".png"):
                continue
            idx = int(os.path.splitext(filename)[0][6:])
            pose_file = rootPath + "blender_poses/" + class_name + "/pose" + str(idx) + '.npy'
            rad_files = [os.path.join(rootPath, 'estRadialMap', class_name, 'Out_pt' + str(k) + '_dm', '_' + str(idx).zfill(5) + '.npy')
                         for k in range(1, 4)]
            condition = os.path.isfile(pose_file) and (opts.using_ckpts or all(os.path.isfile(f) for f in rad_files))
            if condition:
                RTGT = np.load(pose_file)
                depth_map = np.array(Image.open(depthPath + 'depth_' + str(idx).zfill(5) + '.png'), dtype=np.float64)
                H, W = depth_map.shape
                depth_dev = _depth_to_u16(depth_map, "cuda")
                estimated_kpts, statuses = np.zeros((3, 3)), []
                if opts.using_ckpts:
                    if pipe is None:
                        pipe = FramePipeline(sds, H, W, dtype=getattr(opts, 'dtype', 'f16x3'), frames_per_step=1)
                        pipe.set_stage_timing(True)
                        pipe.seg_threshold = seg_thr
                    rgb = np.array(Image.open(jpgPath + filename).convert('RGB'), dtype=np.uint8)
                    pipe.process(torch.from_numpy(rgb)[None].cuda(), depth_dev[None], max_radii_dm)
                    kp, res = pipe.read_results()
                    seg_maps, rad_maps = pipe.seg[:, 0, 0], pipe.radial[:, 0, 0]
                    for k in range(3):  # `if radial_out.max() != 0` (:857): an empty mask leaves the keypoint at zero
                        if res[k].status == _hip.RCV_E_EMPTY and res[k].n_radii == 0:
                            continue
                        statuses.append(res[k].status)
                        estimated_kpts[k] = kp[0][k]
                else:
                    ws = AccWorkspace(H * W)
                    rad_maps = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
                    seg_maps = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
                    for k in range(3):
                        rad_maps[k].copy_(torch.from_numpy(np.ascontiguousarray(np.load(rad_files[k]), dtype=np.float32)))
                        seg_maps[k].copy_((rad_maps[k] > 0).to(torch.float32))  # sem_out = radial_out > 0 after thresholding (:849)
                        seam_compact(seg_maps[k], rad_maps[k], depth_dev, max_radii_dm[k], mode=0, workspace=ws, seg_thr=0.5)
                        n_radii = int(ws.counts[1].item())
                        if n_radii == 0:  # radial_out.max() == 0 (:857)
                            continue
                        accumulate_on_device(ws.xyz, ws.radial, ws.counts, ws.max_points, ws.max_points, ws)
                        r = ws.read_result()
                        statuses.append(r.status)
                        estimated_kpts[k] = r.center_mm[:]
                for st in statuses:
                    _raise_like_reference(st)
                RTGT_mm = np.array(RTGT, dtype=np.float64, copy=True)
                RTGT_mm[:, 3] = RTGT_mm[:, 3] * 1000
                ev = _pose_and_add(class_name, xyz_load, keypoints, estimated_kpts, RTGT_mm, seg_maps, rad_maps, depth_dev, max_radii_dm,
                                   0, criteria, seg_thr if opts.using_ckpts else 0.5)
                if ev["add_distance_mm"] <= add_threshold[class_name] * 1000:
                    bf_icp += 1
                if ev["add_distance_after_icp_mm"] <= add_threshold[class_name] * 1000:
                    af_icp += 1
                per_frame[os.path.splitext(filename)[0]] = ev
            general_counter += 1
            tag = 'ADDs' if class_name in lm_syms else 'ADD'
            print('Current ' + tag + ' of ' + class_name + ' before ICP: ', bf_icp / general_counter)
            print('Currnet ' + tag + ' of ' + class_name + ' after ICP: ', af_icp / general_counter)
        if general_counter:
            tag = 'ADDs' if class_name in lm_syms else 'ADD'
            print(tag + ' of ' + class_name + ' before ICP: ', bf_icp / general_counter)
            print(tag + ' of ' + class_name + ' after ICP: ', af_icp / general_counter)
            results[class_name] = dict(frames=general_counter, evaluated=len(per_frame), add_before_icp=bf_icp / general_counter,
                                       add_after_icp=af_icp / general_counter, **_stage_timers(pipe))
            details[class_name] = per_frame
    _require_some_class(results, missing)
    return (results, details) if return_details else results


def oriented_bbox_extent(points):
    """Edge lengths of `open3d.geometry.PointCloud.get_oriented_bounding_box().extent` as the YCB loop uses it
    (reference AccumulatorSpace.py:1119-1120): the box aligned with the principal axes of the convex hull's vertices.
    Once per class, on the host (scipy's Qhull).  Parity unpinned: Open3D is not installed here."""
    from scipy.spatial import ConvexHull

    pts = np.asarray(points, dtype=np.float64)
    hull = pts[ConvexHull(pts).vertices]
    c = hull - hull.mean(axis=0)
    _, vec = np.linalg.eigh(c.T @ c / len(hull))
    q = c @ vec
    return (q.max(axis=0) - q.min(axis=0))[::-1]


def _auc(x, y):
    """sklearn.metrics.auc (trapezoidal rule), reference AccumulatorSpace.py:1192-1193."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    return float(((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / 2).sum())


def estimate_6d_pose_ycb(opts, return_details=False):
    """The YCB-Video evaluation loop of the reference (AccumulatorSpace.py:976-1197) on the MI355X hot path (SURVEY.md
    section 8 f-4): the same kernels as the LINEMOD loops behind the YCB-Video file layout.

    Layout read: `root_dataset/Split/<class>/val.txt` (names `<cycle>_<idx>`), `models/<class>/{points.xyz, Outside9.npy}`,
    `data/<cycle>/<idx>{-color.png, -depth.png, .mat}` with `poses`, `cls_indexes`, `factor_depth`, `intrinsic_matrix` in the
    `.mat`, checkpoints `model_dir/<class>_pt<k>.pth.tar` saved from `nn.DataParallel` (`module.` keys, :1005).
    Kept from the reference: the point mask is `seg > 0.8` alone (:1051, no radius bound), depth is divided by `factor_depth`
    (:1053), ADD(-S) passes at 1 % of the CAD model's oriented-bounding-box diagonal (:1119-1131), the ADD(-S) AUC is taken
    over the thresholds 0 ... 0.1 m (:978,1132-1140,1192), ICP runs with reach = mean distance and
    `ICPConvergenceCriteria(max_iteration=2000000)` (:1160-1165).

    The reference's loop cannot run as written; this build follows its evident intent and says where:
      * `keypoint_count` is read before it is assigned (:1003) and then indexes `model_list` and `estimated_kpts` from 1 to 3
        (:1044,1090) -- here network k (1..3) estimates `Outside9[k]` into row k-1, and Horn pairs rows 0..2 with
        `Outside9[1:4]`, as the LINEMOD loop does (the reference pairs `Outside9[0:3]`, :1115);
      * `poses[:, :, np.where(cls_indexes == class_id)]` (:1017) indexes with both coordinate arrays of `np.where`; the pose
        of the class is meant;
      * the frame names come from `<class>.hdf5` through h5py (:1012-1013), which only lists them: here the names of
        `val.txt` in sorted order (what h5py's key order is);
      * the summary is printed once after the class loop with counters that were reset per class (:1192-1195): here per class.
    """
    torch = _torch()
    import os

    import scipy.io
    from PIL import Image

    from . import utils
    from .models.fcnresnet import DenseFCNResNet152
    from .pipeline import FramePipeline
    from .registration import ICPConvergenceCriteria

    _hip.lib()
    auc_threshold = [0, 0.02, 0.04, 0.06, 0.08, 0.1]
    results, details = {}, {}
    wanted = getattr(opts, 'class_names', None)
    missing = []
    for class_id in ycb_cls_names:
        class_name = ycb_cls_names[class_id]
        if wanted and class_name not in wanted:
            continue
        if not os.path.isdir(opts.root_dataset + "models/" + class_name):
            _missing_class_dir(opts, opts.root_dataset + "models/" + class_name, missing)
            continue
        print(class_name)
        test_list = [s.replace('\n', '') for s in open(opts.root_dataset + "Split/" + class_name + "/val.txt", "r").readlines()]
        xyz_load = np.loadtxt(opts.root_dataset + "models/" + class_name + "/points.xyz", ndmin=2)[:, :3]
        keypoints = np.load(opts.root_dataset + "models/" + class_name + "/Outside9.npy")
        sds = []
        for i in range(1, 4):
            model_path = opts.model_dir + class_name + "_pt" + str(i) + ".pth.tar"
            if not os.path.exists(model_path):
                raise ValueError(model_path + " not found")  # reference :1029-1030
            model = DenseFCNResNet152(3, 2)
            model, _, _, _ = utils.load_checkpoint(model, None, model_path)
            sds.append(model.state_dict())
        w, h, d = oriented_bbox_extent(xyz_load)
        bbox_diagonal = (w ** 2 + h ** 2 + d ** 2) ** 0.5
        criteria = ICPConvergenceCriteria(max_iteration=2000000)
        no_bound = np.full(3, np.inf)
        auc_adds_count = np.zeros((2, 6))
        bf_icp, af_icp, general_counter = 0, 0, 0
        per_frame = {}
        pipe, ws = None, None
        for filename in sorted(s for s in test_list if s):
            cycle, idx = filename.split('_')
            print(filename)
            sceneInfo = scipy.io.loadmat(opts.root_dataset + 'data/' + cycle + "/" + idx)
            hit = np.nonzero(np.asarray(sceneInfo['cls_indexes']).reshape(-1) == class_id)[0]
            if len(hit) == 0:
                raise ValueError(class_name + " is not annotated in " + filename)
            RTGT = np.asarray(sceneInfo['poses'], dtype=np.float64)[:, :, hit[0]]
            K = np.asarray(sceneInfo['intrinsic_matrix'], dtype=np.float64)
            factor_depth = float(np.asarray(sceneInfo['factor_depth']).reshape(-1)[0])
            depth_unit_mm = 1000.0 / factor_depth
            rgb = np.array(Image.open(opts.root_dataset + 'data/' + cycle + "/" + idx + '-color.png').convert('RGB'), dtype=np.uint8)
            depth_map1 = np.array(Image.open(opts.root_dataset + 'data/' + cycle + "/" + idx + '-depth.png'))
            H, W = depth_map1.shape
            depth_dev = _depth_to_u16(depth_map1, "cuda")
            if pipe is None:
                pipe = FramePipeline(sds, H, W, dtype=getattr(opts, 'dtype', 'f16x3'), frames_per_step=1)
                pipe.set_stage_timing(True)
                ws = AccWorkspace(H * W)
            pipe._mark("begin_net")
            pipe.forward_fcn(torch.from_numpy(rgb)[None].cuda())
            pipe._mark("net")
            seg_maps, rad_maps = pipe.seg[:, 0, 0], pipe.radial[:, 0, 0]
            estimated_kpts = np.zeros((3, 3))
            for k in range(3):
                # mode 3: mask = seg > 0.8 alone, depth / factor_depth, the frame's intrinsics, no /1000 (:1050-1057)
                seam_compact(seg_maps[k], rad_maps[k], depth_dev, factor_depth, K=K, mode=3, workspace=ws)
                pipe._mark("seam")
                accumulate_on_device(ws.xyz, ws.radial, ws.counts, ws.max_points, ws.max_points, ws)
                pipe._mark("acc")
                r = ws.read_result()
                _raise_like_reference(r.status)
                estimated_kpts[k] = r.center_mm[:]
            RTGT_mm = np.array(RTGT, dtype=np.float64, copy=True)
            RTGT_mm[:, 3] = RTGT_mm[:, 3] * 1000
            ev = _pose_and_add(class_name, xyz_load, keypoints, estimated_kpts, RTGT_mm, seg_maps, rad_maps, depth_dev, no_bound,
                               0, criteria, SEG_THRESHOLD, K=K, syms=ycb_syms, scene_scale=depth_unit_mm, reach_is_mean=True)
            if ev["add_distance_mm"] <= bbox_diagonal * 0.01 * 1000:
                bf_icp += 1
            if ev["add_distance_after_icp_mm"] <= bbox_diagonal * 0.01 * 1000:
                af_icp += 1
            for i, thre in enumerate(auc_threshold):
                auc_adds_count[0, i] += ev["add_distance_mm"] <= thre * 1000
                auc_adds_count[1, i] += ev["add_distance_after_icp_mm"] <= thre * 1000
            general_counter += 1
            per_frame[filename] = ev
        if general_counter:
            auc_before = _auc(auc_threshold, auc_adds_count[0] / general_counter) / 0.1
            auc_after = _auc(auc_threshold, auc_adds_count[1] / general_counter) / 0.1
            print('ADD\\(s\\) AUC of ' + class_name + ' before ICP: ', auc_before)
            print('ADD\\(s\\) AUC of ' + class_name + ' after ICP: ', auc_after)
            print('ADD\\(s\\) of ' + class_name + ' before ICP: ', bf_icp / general_counter)
            print('ADD\\(s\\) of ' + class_name + ' after ICP: ', af_icp / general_counter)
            results[class_name] = dict(frames=general_counter, add_before_icp=bf_icp / general_counter,
                                       add_after_icp=af_icp / general_counter, auc_before_icp=auc_before, auc_after_icp=auc_after,
                                       bbox_diagonal_m=float(bbox_diagonal), **_stage_timers(pipe))
            details[class_name] = per_frame
    _require_some_class(results, missing)
    return (results, details) if return_details else results


def main(argv=None):
    """The module's own command line (reference AccumulatorSpace.py:1202-1227): --dataset lm | lmo | ycb."""
    import argparse

    from .utils import str2bool

    parser = argparse.ArgumentParser()
    parser.add_argument('--root_dataset', type=str, default='Datasets/LINEMOD/')
    parser.add_argument('--model_dir', type=str, default='ckpts/')
    parser.add_argument('--demo_mode', type=str2bool, default=False)
    parser.add_argument('--using_ckpts', type=str2bool, default=False)
    parser.add_argument('--dataset', type=str, default='lm', choices=['lm', 'lmo', 'ycb'])
    opts = parser.parse_args(argv)
    return {'lm': estimate_6d_pose_lm, 'lmo': estimate_6d_pose_lmo, 'ycb': estimate_6d_pose_ycb}[opts.dataset](opts)


if __name__ == "__main__":
    main()
