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
