"""The per-frame hot loop of `--mode=test` (reference AccumulatorSpace.py:553-659) as one device-side
pipeline: for a batch of frames and the three keypoint networks,

    uint8 RGB -> normalise -> 3 x FCN forward -> per (frame, keypoint): mask/back-project seam ->
    radial voting in LDS -> arg-max -> estimated_kpts[frame, keypoint] (mm, camera frame)

Nothing returns to the host between the image upload and the final copy of the 96-byte results.
"""
import ctypes

import numpy as np

from . import AccumulatorSpace as A
from . import _hip, fcn_engine


class FramePipeline:
    def __init__(self, state_dicts, H=480, W=640, dtype="f16x3", frames_per_step=1, K=A.linemod_K, acc_unit=A.ACC_UNIT_MM,
                 device="cuda", overlap=False, radial_f16=False, engine="auto", q6=False, fcn=None):
        """acc_unit: voxel edge in mm (the reference hard-codes 5, AccumulatorSpace.py:374).  radial_f16: the FCN head
        writes the radial maps as float16 and the seam reads them as such (the stress configuration of BASELINE.json).
        engine: 'auto' / 'full', the arg-max engine of rcv_accumulate_3d (identical results).  q6: FcnEngine's q6 (fp6 cross
        terms in the stride-1 3x3 convolutions of layers 2-4; default off).  fcn: an FcnEngine already built for these networks and this
        input size (BN folding, weight packing and upload take seconds: a caller that sizes its batch first builds the engine once)."""
        import torch

        assert len(state_dicts) >= 1
        self.torch = torch
        self.lib = _hip.lib()
        _hip.check(self.lib.rcv_init(), "rcv_init")
        self.H, self.W, self.B, self.G = H, W, int(frames_per_step), len(state_dicts)
        self.K = np.asarray(K, dtype=np.float64)
        self.acc_unit = float(acc_unit)
        self.radial_f16 = bool(radial_f16)
        self.vote_engine = _hip.VOTE_ENGINES[engine]
        self.seg_threshold = float(A.SEG_THRESHOLD)  # reference :603 (LINEMOD); the LM-O driver uses >= 0.5 (:836)
        self.device = device
        self.engine = fcn if fcn is not None else fcn_engine.FcnEngine(state_dicts, H, W, dtype=dtype, device=device, q6=q6)
        self.engine.workspace(self.B)
        if self.radial_f16:
            self.engine.set_option(8, 1)
        B, G = self.B, self.G
        rad_dtype = torch.float16 if self.radial_f16 else torch.float32
        self.x = torch.empty((B, 3, H, W), dtype=torch.float32, device=device)
        # two sets of FCN outputs: the forward of batch s+1 may run while batch s is being voted on
        self._seg2 = [torch.empty((G, B, 1, H, W), dtype=torch.float32, device=device) for _ in range(2)]
        self._rad2 = [torch.empty((G, B, 1, H, W), dtype=rad_dtype, device=device) for _ in range(2)]
        self.seg, self.radial = self._seg2[0], self._rad2[0]
        n_items = B * G
        cap = H * W
        self.cap = cap
        ws_bytes = int(self.lib.rcv_acc_workspace_bytes_min(cap))  # the items of a batch fill the chip: no replica area
        self.ws_bytes = ws_bytes
        self.ws = torch.empty((n_items, ws_bytes), dtype=torch.uint8, device=device)
        self.xyz = torch.empty((n_items, cap, 3), dtype=torch.float64, device=device)
        self.rlist = torch.empty((n_items, cap), dtype=torch.float32, device=device)
        self.counts = torch.zeros((n_items, 4), dtype=torch.int32, device=device)  # [points, radii, non-finite pixels of the maps, their running maximum]
        self.results = torch.zeros((n_items, ctypes.sizeof(_hip.AccResult)), dtype=torch.uint8, device=device)
        self.k4 = (ctypes.c_double * 4)(self.K[0, 0], self.K[1, 1], self.K[0, 2], self.K[1, 2])
        # overlapped mode (enqueue_step): FCN on one stream, seam + voting + result copy on another
        # overlap=False: one stream for both (kernel timings free of cross-stream time slicing)
        self.s_fcn = torch.cuda.Stream(device=device)
        self.s_vote = torch.cuda.Stream(device=device) if overlap else self.s_fcn
        self._ev_fcn = [None, None]
        self._ev_vote = [None, None]
        self._step = 0
        self._timing = False
        self._marks = []

    @staticmethod
    def device_bytes(fcn, frames_per_step, H, W, G, radial_f16=False):
        """Device bytes a FramePipeline of `frames_per_step` frames allocates around the engine `fcn` (its activation workspace, the
        double-buffered output planes, and per (frame, keypoint) item the voting workspace, point and radius lists)."""
        B, HW = int(frames_per_step), H * W
        per_item = int(_hip.lib().rcv_acc_workspace_bytes_min(HW)) + HW * (24 + 4) + 16 + ctypes.sizeof(_hip.AccResult)
        planes = 2 * G * B * HW * (4 + (2 if radial_f16 else 4))
        return fcn.workspace_bytes(B) + B * 3 * HW * 4 + planes + B * G * per_item

    # -- stages ----------------------------------------------------------------------------
    def forward_fcn(self, rgb_u8):
        """rgb_u8: uint8 [B,H,W,3] on the device -> fills self.seg / self.radial."""
        L = self.lib
        _hip.check(L.rcv_preprocess_rgb(_hip.ptr(rgb_u8), self.B, self.H, self.W, _hip.ptr(self.x), _hip.current_stream()),
                   "rcv_preprocess_rgb")
        self.engine.forward(self.x, out=(self.seg, self.radial))

    def _strided(self, t, stride_frame, stride_kp):
        return _hip.Strided(t.data_ptr(), int(stride_frame), int(stride_kp))

    # -- stage timers (the reference's net_time / acc_time, AccumulatorSpace.py:594-599, 627-630) ------
    def set_stage_timing(self, enable):
        """HIP events on the launch streams around the stages of every enqueued step: 'net' (image preparation + the
        three forwards = the reference's net_time), 'inject' (measurement only), 'seam', 'acc' (Accumulator_3D = the
        reference's acc_time), 'collect' (gather / result copy).  Read with stage_times()."""
        self._timing = bool(enable)
        self._marks = []

    def _mark(self, name):
        if self._timing:
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record()  # on torch's current stream = the stream the kernels are launched on
            self._marks.append((name, ev))

    def stage_times(self):
        """-> {stage: total ms over the steps enqueued since set_stage_timing(True)}; waits for them."""
        out = {}
        self.finish()
        for (n0, e0), (n1, e1) in zip(self._marks[:-1], self._marks[1:]):
            if n1.startswith("begin"):
                continue  # the gap between two steps (or between the two streams of a step) is nobody's stage
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out

    def vote(self, depth_u16, max_radii_dm):
        """depth_u16: [B,H,W] (int16 storage of uint16 mm); max_radii_dm: G floats (one bound per keypoint, as the reference's
        max_radii_dm[k]) or [B,G] (one per frame and keypoint).  Enqueues seam + voting for all B*G (frame, keypoint) items
        as ONE batched launch sequence (item = blockIdx.y/z of every kernel)."""
        L, torch = self.lib, self.torch
        B, G, HW = self.B, self.G, self.H * self.W
        if not torch.is_tensor(max_radii_dm):
            max_radii_dm = torch.tensor(np.asarray(max_radii_dm, dtype=np.float64), device=self.device)
        self._rmax = max_radii_dm.to(torch.float64).contiguous()  # keep alive until the kernels ran
        if tuple(self._rmax.shape) not in ((G,), (B, G)):
            raise ValueError("max_radii_dm must have shape [%d] or [%d, %d]" % (G, B, G))
        st = _hip.current_stream()
        S = self._strided
        item = lambda t, per: S(t, G * per, per)  # buffers indexed [frame * G + keypoint]
        seg = S(self.seg, HW * 4, B * HW * 4)
        rb = self.radial.element_size()
        rad = S(self.radial, HW * rb, B * HW * rb)
        seam_mode = (_hip.RCV_SEAM_RADIAL_F16 if self.radial_f16 else 0) | _hip.RCV_SEAM_COUNT_NONFINITE
        dep = S(depth_u16, HW * 2, 0)
        rmx = S(self._rmax, 8 * G if self._rmax.dim() == 2 else 0, 8)
        xyz = item(self.xyz, self.cap * 24)
        rl = item(self.rlist, self.cap * 4)
        cnt = item(self.counts, 16)
        ws = item(self.ws, self.ws_bytes)
        res = item(self.results, ctypes.sizeof(_hip.AccResult))
        _hip.check(L.rcv_seam_compact_batch(seg, rad, dep, self.H, self.W, self.k4, float(self.seg_threshold), rmx, seam_mode, xyz, rl, cnt,
                                            self.cap, ws, self.ws_bytes, B, G, st), "rcv_seam_compact_batch")
        self._mark("seam")
        _hip.check(L.rcv_accumulate_3d_batch(xyz, rl, 0, cnt, self.cap, self.acc_unit, self.vote_engine, res, ws, self.ws_bytes, B, G, st),
                   "rcv_accumulate_3d_batch")
        self._mark("acc")

    def read_results(self):
        """One D2H copy -> (estimated_kpts float64 [B,G,3] mm, list of AccResult).  Raises _hip.HalfRangeError when a forward
        since the last call stored an activation beyond the half-precision range (f16x3 / fp16 engines: clipped maps)."""
        self.finish()  # (first: the copies below run on the caller's stream, the votes of enqueue_step on the pipeline's own)
        raw = self.results.cpu().numpy()
        nonfinite = int(self.counts[:, 3].max().item())
        self.counts[:, 3].zero_()
        self.engine.range_status(clear=True, raise_on_flag=True)
        # the range guard sees magnitudes, not NaNs (v_max drops them, and so does the next ReLU), and the mask comparisons drop a NaN
        # pixel without a trace: the seam counts the non-finite values of the very maps the votes are cast on (counts[:, 2]) and keeps
        # the largest count of every batch enqueued since the last call (counts[:, 3]) -- whichever output buffer a later forward has taken since
        if nonfinite != 0:
            raise FloatingPointError("non-finite FCN output in %s mode (NaN weights or input?): up to %d pixels of a map"
                                     % (self.engine.dtype_name, nonfinite))
        res = [_hip.AccResult.from_buffer_copy(raw[i].tobytes()) for i in range(raw.shape[0])]
        kpts = np.array([[r.center_mm[0], r.center_mm[1], r.center_mm[2]] for r in res]).reshape(self.B, self.G, 3)
        return kpts, res

    def keypoints_tensor(self):
        """estimated keypoints as a float64 [B,G,3] device tensor (view of the result records)."""
        return self.results[:, :24].contiguous().view(self.torch.float64).view(self.B, self.G, 3)

    # -- overlapped stream of batches ------------------------------------------------------
    def enqueue_step(self, rgb_u8, depth_u16, max_radii_dm, inject=None, collect=None):
        """Enqueue one batch without waiting for it: the FCN of this batch runs on its own stream while the
        previous batch is still being voted on (double-buffered FCN outputs; everything downstream of the
        forward is ordered on the voting stream).  `collect(kpts_dev)` is called on the voting stream after the
        arg-max (e.g. all_gather + async copy to pinned memory) and must only enqueue work.  Call finish() to
        wait.  The caller keeps rgb_u8 / depth_u16 alive and unmodified until then."""
        torch = self.torch
        buf = self._step & 1
        self._step += 1
        cur = torch.cuda.current_stream()
        self.s_fcn.wait_stream(cur)  # inputs prepared on the caller's stream
        # the inputs were allocated on the caller's stream but are consumed on ours: tell the caching allocator
        rgb_u8.record_stream(self.s_fcn)
        depth_u16.record_stream(self.s_vote)
        if torch.is_tensor(max_radii_dm):
            max_radii_dm.record_stream(self.s_vote)
        if inject is not None:
            inject[0].record_stream(self.s_fcn)
            inject[1].record_stream(self.s_fcn)
        with torch.cuda.stream(self.s_fcn):
            if self._ev_vote[buf] is not None:
                self.s_fcn.wait_event(self._ev_vote[buf])  # the votes that read this buffer two batches ago are done
            self.seg, self.radial = self._seg2[buf], self._rad2[buf]
            self._mark("begin_net")
            self.forward_fcn(rgb_u8)
            self._mark("net")
            if inject is not None:
                self.seg.copy_(inject[0])
                self.radial.copy_(inject[1])
                self._mark("inject")
            ev = torch.cuda.Event()
            ev.record(self.s_fcn)
            self._ev_fcn[buf] = ev
        with torch.cuda.stream(self.s_vote):
            self.s_vote.wait_event(self._ev_fcn[buf])
            self._mark("begin_vote")
            self.vote(depth_u16, max_radii_dm)
            ev = torch.cuda.Event()
            ev.record(self.s_vote)
            self._ev_vote[buf] = ev
            if collect is not None:
                collect(self.keypoints_tensor())
                self._mark("collect")

    def finish(self):
        self.s_fcn.synchronize()
        self.s_vote.synchronize()

    # -- whole step ------------------------------------------------------------------------
    def process(self, rgb_u8, depth_u16, max_radii_dm, inject=None):
        """One batch of frames.  `inject`: optional (seg [G,B,1,H,W], radial [G,B,1,H,W]) copied over the FCN
        outputs after the forward (measurement with random weights)."""
        self._mark("begin_net")
        self.forward_fcn(rgb_u8)
        self._mark("net")
        if inject is not None:
            self.seg.copy_(inject[0])
            self.radial.copy_(inject[1])
            self._mark("inject")
        self.vote(depth_u16, max_radii_dm)
