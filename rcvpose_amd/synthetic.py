"""Synthetic LINEMOD-shaped RGB-D frames for measurement and parity tests.

There is no dataset in the reference tree, so the workload is generated: an ellipsoid is ray-cast
through the LINEMOD intrinsics into a 640x480 uint16 depth map (~15k silhouette pixels), the three
keypoints sit at 2x three corners of the object's bounding box (as rmap_dataset.py:47-49 scales the
Outside9 keypoints), and the "ideal" network outputs are seg = +1 / -1 and radial = distance to the
keypoint in decimetres plus N(0, 0.02 dm) noise (label conventions of data_loader.py:104).
"""
import numpy as np

LINEMOD_K = (572.4114, 573.57043, 325.2611, 242.04899)  # fx, fy, cx, cy -- AccumulatorSpace.py:59-61
YCB_K = (1066.778, 1067.487, 312.9869, 241.3109)        # public YCB-Video camera (cmu) intrinsics

_LM = dict(K=LINEMOD_K, axes=(0.12, 0.10, 0.08), centre=(0.02, -0.03, 0.90))
_YCB = dict(K=YCB_K, axes=(0.0975, 0.079, 0.065), centre=(0.01, -0.02, 0.75))  # ~50k silhouette pixels


def synthetic_frame(frame_idx, H=480, W=640, shape="lm", noise_dm=0.02, outlier_frac=0.0, seg_false_pos=0.0):
    """outlier_frac: fraction of the silhouette pixels whose radius is replaced by a uniform value in (0, max radius] (gross
    regression errors); seg_false_pos: background pixels (a far plane at 1.3 m, in a band around the silhouette) wrongly
    segmented as object, as a fraction of the silhouette size, with uniform random radii.  Both default to 0 and then leave the
    frame exactly as before (the extra random draws come after all others)."""
    cfg = _LM if shape == "lm" else _YCB
    fx, fy, cx, cy = cfg["K"]
    a, b, c = cfg["axes"]
    centre = np.array(cfg["centre"], dtype=np.float64)
    rng = np.random.Generator(np.random.PCG64(1234 + int(frame_idx)))
    rgb = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    # small per-frame jitter of the object position so that frames differ
    centre = centre + rng.normal(0, 0.004, size=3)

    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    d = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)  # ray through the pixel, z = 1
    inv = 1.0 / np.array([a, b, c]) ** 2
    A = (d * d * inv).sum(-1)
    B = -2.0 * (d * centre * inv).sum(-1)
    C = (centre * centre * inv).sum() - 1.0
    disc = B * B - 4 * A * C
    hit = disc > 0
    t = np.where(hit, (-B - np.sqrt(np.where(hit, disc, 0.0))) / (2 * A), 0.0)  # nearest intersection; z = t
    depth = np.where(hit, np.rint(t * 1000.0), 0).astype(np.uint16)

    sign = np.array([[1, 1, 1], [-1, 1, -1], [1, -1, -1]], dtype=np.float64)
    keypoints = centre + 2.0 * sign * np.array([a, b, c])

    z = depth.astype(np.float64)
    xyz = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], axis=-1) / 1000.0
    inside = depth != 0
    seg = np.where(inside, 1.0, -1.0).astype(np.float32)
    radial = np.zeros((3, H, W), dtype=np.float32)
    max_radii = np.zeros(3)
    for k in range(3):
        r = np.linalg.norm(xyz - keypoints[k], axis=-1) * 10.0
        max_radii[k] = r[inside].max()
        noisy = r + rng.normal(0, noise_dm, size=r.shape)
        radial[k] = np.where(inside, np.minimum(noisy, max_radii[k]), 0.0).astype(np.float32)
    if outlier_frac > 0.0:
        for k in range(3):
            bad = inside & (rng.random(size=inside.shape) < outlier_frac)
            radial[k] = np.where(bad, rng.uniform(0.05, max_radii[k], size=inside.shape), radial[k]).astype(np.float32)
    if seg_false_pos > 0.0:
        # band of ~12 pixels around the silhouette (dilation by shifts), background depth = plane at 1.3 m
        band = np.zeros_like(inside)
        for dy in (-12, -6, 0, 6, 12):
            for dx in (-12, -6, 0, 6, 12):
                band |= np.roll(np.roll(inside, dy, axis=0), dx, axis=1)
        band &= ~inside
        p = min(1.0, seg_false_pos * inside.sum() / max(1, band.sum()))
        fp = band & (rng.random(size=inside.shape) < p)
        depth = np.where(fp, 1300, depth).astype(np.uint16)
        seg = np.where(fp, 1.0, seg).astype(np.float32)
        for k in range(3):
            radial[k] = np.where(fp, rng.uniform(0.05, max_radii[k], size=inside.shape), radial[k]).astype(np.float32)
    # float32 storage may round the clipped maximum above the float64 bound; keep the bound inclusive
    max_radii = np.maximum(max_radii, radial.reshape(3, -1).max(axis=1).astype(np.float64))
    return {
        "rgb": rgb, "depth": depth, "seg": seg, "radial": radial, "max_radii_dm": max_radii,
        "keypoints_m": keypoints, "n_silhouette": int(inside.sum()), "K": cfg["K"],
    }
