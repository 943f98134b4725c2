"""Horn pose fit on the device against the golden vectors of the reference and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_horn_golden_bit_exact(golden_dir, oracle):
    from rcvpose_amd.util.horn import HornPoseFitting

    g = np.load(os.path.join(golden_dir, "horn_cases.npz"))
    horn = HornPoseFitting()
    for i in range(len(g["n"])):
        n = int(g["n"][i])
        P1, P2 = g["P1"][i][:n].copy(), g["P2"][i][:n].copy()
        A = np.zeros((4, 4))
        horn.lmshorn(P1, P2, n, A)
        assert np.array_equal(A, g["RT"][i]), (i, np.abs(A - g["RT"][i]).max())
        assert np.array_equal(P1, g["P1"][i][:n]) and np.array_equal(P2, g["P2"][i][:n])


def test_horn_batch_from_result_records(oracle):
    """Poses formed on the device from strided keypoint records equal the oracle's, frame by frame."""
    from rcvpose_amd.util.horn import horn_batch

    rng = np.random.default_rng(4)
    model = rng.normal(size=(3, 3)) * 100
    est = rng.normal(size=(37, 3, 3)) * 100 + np.array([0, 0, 900.0])
    rt = horn_batch(model, torch.from_numpy(est).cuda()).cpu().numpy()
    for f in range(len(est)):
        assert np.array_equal(rt[f], oracle.horn(model, est[f])), f


def test_horn_fuzz_against_oracle(oracle):
    """Seeded fuzz: exact rigid motions, noisy estimates, nearly collinear and coincident keypoints, mirrored sets, tiny and
    huge scales -- RT from the device equals the oracle's (the reference's numpy operations restated), bit for bit, NaNs in
    the same places."""
    from rcvpose_amd.util.horn import horn_batch

    rng = np.random.default_rng(99)
    models, ests = [], []
    for case in range(240):
        scale = float(rng.choice([1e-3, 1.0, 100.0, 1e4]))
        model = rng.normal(size=(3, 3)) * scale
        kind = case % 6
        if kind == 1:
            model[2] = model[0] + (model[1] - model[0]) * 0.5 + rng.normal(size=3) * scale * 1e-9   # nearly collinear
        if kind == 2:
            model[1] = model[0]                                                                   # two coincident keypoints
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        est = model @ R.T + rng.normal(size=3) * scale * 5
        if kind == 3:
            est = est + rng.normal(size=(3, 3)) * scale * 0.05                                   # noisy estimate
        if kind == 4:
            est = est * np.array([1.0, 1.0, -1.0])                                                # mirrored: no proper rotation fits
        if kind == 5:
            est = np.repeat(est[:1], 3, axis=0)                                                   # all estimates coincide
        models.append(model)
        ests.append(est)
    for model, est in zip(models, ests):
        rt = horn_batch(model, torch.from_numpy(est[None]).cuda()).cpu().numpy()[0]
        ref = oracle.horn(model, est)
        assert np.array_equal(rt, ref, equal_nan=True), (model, est, rt, ref)
