"""`train.Trainer` of the reference, inference half (reference train.py:17-26, :190-194).

Only `Test()` is implemented here -- it dispatches to the evaluation driver exactly like the reference.  The
training loop is outside the scope of this build ("OUT OF SCOPE -- not inference").
Unlike the reference, constructing a Trainer in test mode does not already run the evaluation (the reference runs
it once in __init__ and once more from main.py)."""
from .AccumulatorSpace import estimate_6d_pose_lm, estimate_6d_pose_lmo


class Trainer:
    def __init__(self, data_loader, opts, vis=None):
        self.opts = opts
        self.train_loader, self.val_loader = (data_loader + [None, None])[:2] if isinstance(data_loader, list) else (None, None)
        self.vis = vis

    def Train(self):
        raise NotImplementedError("rcvpose_amd covers the --mode=test hot path; training is out of scope")

    def Test(self):
        if getattr(self.opts, 'test_occ', False):
            return estimate_6d_pose_lmo(self.opts)
        return estimate_6d_pose_lm(self.opts)
