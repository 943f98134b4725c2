"""`util.horn` of the reference on the device: HornPoseFitting.lmshorn (reference util/horn.py:75-181).

Same call convention: lmshorn(P1, P2, n, A) fills the 4x4 array A in place with the rigid transform that maps
the model keypoints P1 onto the estimated keypoints P2 (P2 ~ R P1 + T).  Unlike the reference, P1 and P2 are
left bit-for-bit untouched (the reference subtracts and re-adds the centroids in place)."""
import numpy as np

from .. import _hip


def horn_batch(model_kpts_mm, est_kpts_dev, n_points=3):
    """est_kpts_dev: float64 CUDA tensor [F, n_points, 3] (contiguous) -> float64 CUDA tensor [F,4,4]."""
    import torch

    L = _hip.lib()
    model = torch.as_tensor(np.ascontiguousarray(model_kpts_mm, dtype=np.float64)).cuda()
    F = est_kpts_dev.shape[0]
    est = est_kpts_dev.contiguous()
    out = torch.empty((F, 4, 4), dtype=torch.float64, device=est.device)
    _hip.check(L.rcv_horn_pose_batch(_hip.ptr(model), _hip.Strided(est.data_ptr(), n_points * 24, 24), n_points, F,
                                     _hip.ptr(out), _hip.current_stream()), "rcv_horn_pose_batch")
    return out


class HornPoseFitting:
    def lmshorn(self, P1, P2, n, A):
        import torch

        P1 = np.asarray(P1, dtype=np.float64)[:n]
        P2 = np.asarray(P2, dtype=np.float64)[:n]
        rt = horn_batch(P1, torch.from_numpy(np.ascontiguousarray(P2[None])).cuda(), n_points=n)[0].cpu().numpy()
        A[:4, :4] = rt
