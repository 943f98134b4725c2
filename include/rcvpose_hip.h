/*
 * rcvpose_hip.h -- C ABI of librcvpose_hip.so, the MI355X (gfx950) implementation of the RCVPose
 * `--mode=test` hot path: ResNet-152-FCN forward -> mask / back-project seam -> 3-D radial voting
 * -> arg-max keypoint.
 *
 * The reference (aaronWool/rcvpose) is pure Python and has no FFI layer; its boundary for this
 * path is a set of Python callables.  Each entry point below names the
 * reference callable (file:line) it stands behind.  The Python mirrors of those callables in
 * rcvpose_amd/ (same names, arguments and error behaviour) bind this header with ctypes; the
 * reference-side stub a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer named *_dev (and every `ws`) is a DEVICE pointer owned by the caller;
 *     the callee never allocates on the hot path (one lazily-built constant table per device).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - return value: 0 = enqueued, negative = rcv_status; data-dependent failures that the
 *     reference reports by raising (empty cloud, empty grid) are reported in rcv_acc_result.status.
 *   - re-entrant across distinct workspaces / handles; a workspace or handle must not be used
 *     by two streams at once.
 */
#ifndef RCVPOSE_HIP_H
#define RCVPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rcv_status {
    RCV_OK = 0,
    RCV_E_INVALID = -1,       /* bad argument (null pointer, non-positive size, misaligned) */
    RCV_E_HIP = -2,           /* a HIP runtime call failed; see rcv_last_hip_error() */
    RCV_E_WORKSPACE = -3,     /* workspace too small */
    RCV_E_EMPTY = -4,         /* no points / no radii: reference raises ValueError (AccumulatorSpace.py:390) */
    RCV_E_BADGRID = -5,       /* grid dimension <= 0: reference raises ValueError (AccumulatorSpace.py:401,406) */
    RCV_E_TOO_LARGE = -6,     /* grid dimension, radius or tile count beyond what this build supports */
    RCV_E_GRID_BUFFER = -7,   /* vote_grid_dev given but smaller than dim^3 cells (votes still counted) */
    RCV_E_UNSUPPORTED = -8,   /* layer shape / dtype not supported by the FCN engine */
    RCV_E_RANGE = -9          /* FCN engine, half-precision storage (RCV_F16X3 / RCV_F16): an activation left the half range
                                 (|v| > 65504) in a forward that has completed -- its maps were clipped (RCV_F16X3) or hold
                                 infinities (RCV_F16), not the reference's fp32 values.  Sticky: see rcv_fcn_range_status */
} rcv_status;

const char *rcv_strerror(int status);
/* ABI version of this header: bumped whenever a struct layout, an entry point's signature or a documented behaviour changes
 * (600 = round 6: rcv_op grew wq_off (88 bytes), rcv_fcn_range_status takes a stream, rcv_abi_check, RCV_SEAM_COUNT_NONFINITE).
 * rcv_version() returns the library's value; a binding compiled against another header must refuse to run (an ops array with
 * the wrong stride would be read as a garbage program): call rcv_abi_check once after loading the library. */
#define RCV_ABI_VERSION 600
int rcv_version(void);
/* RCV_OK iff the caller's header matches the library: version == rcv_version() and the struct sizes the caller compiled
 * (sizeof(rcv_op), sizeof(rcv_acc_result), sizeof(rcv_strided), sizeof(rcv_icp_result)) equal the library's; RCV_E_INVALID otherwise. */
int rcv_abi_check(int version, size_t sizeof_op, size_t sizeof_acc_result, size_t sizeof_strided, size_t sizeof_icp_result);
/* Text of the last failing HIP call in this process (for RCV_E_HIP). */
const char *rcv_last_hip_error(void);
/* Measurement aid (bench.py; no reference counterpart): runs a bare v_mfma_f32_16x16x32_f16 loop on every SIMD of the device for
 * about `seconds` and returns the dense fp16 TFLOP/s it reached and (optional) the in-kernel shader clock in MHz
 * (s_memtime / s_memrealtime, median over workgroups) -- the matrix rate THIS device holds under load, which differs by
 * several per cent between MI355X devices.  Synchronises `stream`. */
int rcv_mfma_calibrate(double seconds, double *tflops_out, double *clock_mhz_out, void *stream);
/* Builds the per-device constant tables (shell thresholds).  Called lazily by the entry points;
 * call it once up front to keep the first timed call clean. */
int rcv_init(void);

/* ------------------------------------------------------------------------------------------
 * Voting path
 * ------------------------------------------------------------------------------------------ */

/* Written by the device at the end of rcv_accumulate_3d; copy it back with one 96-byte D2H. */
typedef struct rcv_acc_result {
    double center_mm[3];   /* row 0 of the reference's `center` (AccumulatorSpace.py:413-415) */
    double mean_vox[3];    /* x/y/z_mean_mm, voxel units (AccumulatorSpace.py:381-383) */
    double radius_max;     /* radial_list_mm.max() (AccumulatorSpace.py:392) */
    int32_t peak_votes;    /* VoteMap_3D.max() */
    int32_t n_ties;        /* rows of argwhere(VoteMap_3D == max) (AccumulatorSpace.py:406) */
    int32_t first_idx[3];  /* voxel (i,j,k) of the first maximum in C order */
    int32_t dim;           /* D = length + int(radius_max) (AccumulatorSpace.py:401) */
    int32_t zero_boundary; /* AccumulatorSpace.py:394 */
    int32_t n_points;      /* N */
    int32_t n_radii;       /* M >= N (SURVEY a-4) */
    int32_t status;        /* rcv_status */
} rcv_acc_result;

#define RCV_SEAM_RADIAL_F16 256
/* rcv_seam_compact[_batch] mode bit: `counts` has room for FOUR int32 per item and the seam also writes counts[2] = the number of
 * pixels whose segmentation or radial value is not finite (NaN / Inf -- the mask comparisons silently drop such pixels, here as in
 * the reference, AccumulatorSpace.py:603-610; a caller that wants to refuse such maps reads the count with its results) and
 * counts[3] = max(counts[3], counts[2]): the largest such count of any call since the caller last zeroed it (a stream of batches
 * that reuses the buffer is checked once, at the end). */
#define RCV_SEAM_COUNT_NONFINITE 512

/* Bytes of workspace for rcv_seam_compact / rcv_accumulate_3d for clouds of up to max_points points (and radii).
 * rcv_acc_workspace_bytes_min is what every call needs (~45 MB at 640x480); rcv_acc_workspace_bytes adds the replica area
 * with which the LDS rasteriser splits a tile's point list over up to 8 workgroups (load balance of a SINGLE item;
 * ~250 MB more).  The calls take whatever lies between the two and size the split from it: batched callers, whose
 * items already fill the chip, allocate the minimum per item. */
size_t rcv_acc_workspace_bytes(int max_points);
size_t rcv_acc_workspace_bytes_min(int max_points);

/* How rcv_accumulate_3d finds the arg-max when no vote_grid_dev is requested (per call; both give the identical peak,
 * tie count and first index in C order):
 *   RCV_VOTE_AUTO   exact branch-and-bound peak search; the LDS rasteriser takes over when the search gives up
 *   RCV_VOTE_RASTER LDS rasteriser only (the full VoteMap is accumulated tile by tile in LDS). */
#define RCV_VOTE_AUTO 0
#define RCV_VOTE_RASTER 1

/*
 * The mask / back-projection seam of the per-keypoint loop, reference AccumulatorSpace.py:603-610
 * and :619, with rgbd_to_point_cloud (:77-85) fused in:
 *     mask = (seg > seg_thr) & (radial <= rmax_dm);  depth *= mask
 *     xyz  = backproject(depth.nonzero()) / 1000      (metres, row-major pixel order)
 *     radial_list = radial[mask]                       (decimetres, row-major pixel order)
 * mode 0: the using_ckpts branch above; radial_list may be longer than xyz when the mask covers
 *         a depth hole (the reference's behaviour, kept).
 * mode 1: the using_ckpts=False branch (:612-618): mask = (radial <= rmax_dm ? radial : 0) != 0,
 *         radii gathered at depth.nonzero(); seg_dev is ignored.
 * mode 2: rgbd_to_point_cloud(K, depth) alone (:77-85): points of depth.nonzero() in MILLIMETRES
 *         (no /1000); seg_dev, radial_dev and radial_list_dev are ignored (may be null).
 * mode 3: the YCB-Video seam (estimate_6d_pose_ycb, :1050-1057): mask = seg > seg_thr alone (no radius bound),
 *         depth_map = depth * mask / factor_depth with factor_depth passed in rmax_dm, K = the frame's own intrinsics,
 *         xyz = rgbd_to_point_cloud(K, depth_map) in the depth map's unit (metres; no /1000), radial_list = radial[mask].
 * (The Occlusion-LINEMOD seam with pre-computed maps, :849-851,858-864, is mode 0 with seg := (radial > 0) and
 *  seg_thr = 0.5; both variants are pinned by tests/golden/seam_lmo*.npz / seam_ycb*.npz, generated by the reference.)
 * OR RCV_SEAM_RADIAL_F16 into mode when radial_dev holds float16 (the stress configuration of BASELINE.json); the
 * values then enter the reference arithmetic as the float32 numbers they convert to.
 * seg_dev, radial_dev: float32 [H,W]; depth_dev: uint16 [H,W] (mm); K = {fx, fy, cx, cy}.
 * Outputs: xyz_m_dev float64 [cap,3], radial_list_dev float32 [cap], counts_dev int32[2] =
 * {n_points, n_radii}.  cap must be >= H*W unless the caller knows a tighter bound.
 */
int rcv_seam_compact(const float *seg_dev, const float *radial_dev, const uint16_t *depth_dev, int H, int W,
                     const double K[4], float seg_thr, double rmax_dm, int mode, double *xyz_m_dev,
                     float *radial_list_dev, int32_t *counts_dev, int cap, void *ws, size_t ws_bytes, void *stream);

/*
 * Accumulator_3D(xyz, radial_list) -> center, reference AccumulatorSpace.py:373-419, including
 * fast_for (:325-341) and the arg-max (:406-415).
 *   xyz_m_dev     float64 [n_points,3], metres (not modified)
 *   radial_dm_dev float32 or float64 [n_radii], decimetres (radial_is_f64 selects)
 *   counts_dev    optional int32[2] = {n_points, n_radii} living on the device (output of
 *                 rcv_seam_compact); when non-null it overrides n_points / n_radii, which then only
 *                 give upper bounds (cap) so that nothing has to be copied to the host first.
 *   acc_unit_mm   voxel edge in mm (the reference hard-codes 5, AccumulatorSpace.py:374)
 *   engine        RCV_VOTE_AUTO / RCV_VOTE_RASTER (above)
 *   result_dev    rcv_acc_result on the device
 *   vote_grid_dev optional uint32 [dim^3] copy of VoteMap_3D (tests, tie listing); may be null.
 * The votes are integer counts accumulated in LDS sub-volumes; they are bit-identical to the
 * sequential reference loop.
 */
int rcv_accumulate_3d(const double *xyz_m_dev, const void *radial_dm_dev, int radial_is_f64,
                      const int32_t *counts_dev, int n_points, int n_radii, double acc_unit_mm, int engine,
                      rcv_acc_result *result_dev, uint32_t *vote_grid_dev, size_t vote_grid_cells, void *ws,
                      size_t ws_bytes, void *stream);

/*
 * The RGB-only variant (defined but not called by the reference's drivers).
 *
 * rcv_backproject_no_depth: rgbd_to_point_cloud_no_depth(K, depth), reference AccumulatorSpace.py:100-120.
 *   depth_m_dev  float64 [H,W], METRES (the reference scales by 1000 only to count the hypotheses); K = {fx, fy, cx, cy}
 *   pts_dev      float64 [Z,N,3]: every nonzero pixel (row-major) at the depths zmin + i * 0.001, i < Z =
 *                int(zmax*1000) + 1 - int(zmin*1000); cap_points bounds Z*N.  pts_dev = NULL: size query only.
 *   zn_out       HOST int32[2] = {Z, N}.  The result's shape depends on the data, so this call synchronises the stream
 *                (as the reference's return value does).  ws: 256 + 4*H*W bytes of device scratch.
 *   No nonzero pixel: RCV_E_EMPTY (the reference raises ValueError at :103).
 *
 * rcv_accumulate_3d_no_depth: Accumulator_3D_no_depth(xyz, radial_list, pixel_coor) with fast_for_no_depth, reference
 * AccumulatorSpace.py:421-479, computed exactly as the reference computes it on a [Z,N,3] array:
 *   xyz_mm = xyz*200; the three "means" are taken over xyz_mm[:,0], [:,1], [:,2] -- all hypotheses and coordinates of
 *   pixel 0 / 1 / 2 -- and subtracted from those three pixels only (:440-445); radii = radial_list*20, NOT rounded (:427);
 *   grid edge int(max)+1 without a radius margin (:459); every hypothesis of pixel p votes with radius p; `center` is an
 *   int64 array, so row 0 = int64(idx + zero_boundary + mean) * 5 and the other rows (idx + zero_boundary) * 5 (:468-477).
 *   xyz_m_dev float64 [n_hypotheses, n_pixels, 3] metres; radial_dm_dev float32 / float64 [n_radii >= n_pixels].
 *   result_dev->center_mm holds row 0 (integer-valued), mean_vox the three means, dim the grid edge.
 *   n_pixels < 3: RCV_E_INVALID (the reference raises IndexError).  Workspace: rcv_acc_workspace_bytes(Z*N).
 */
int rcv_backproject_no_depth(const double *depth_m_dev, int H, int W, const double K[4], double *pts_dev, long long cap_points,
                             int32_t *zn_out, void *ws, size_t ws_bytes, void *stream);
int rcv_accumulate_3d_no_depth(const double *xyz_m_dev, const void *radial_dm_dev, int radial_is_f64, int n_hypotheses, int n_pixels,
                               int n_radii, int engine, rcv_acc_result *result_dev, uint32_t *vote_grid_dev, size_t vote_grid_cells,
                               void *ws, size_t ws_bytes, void *stream);

/*
 * Batched forms: one launch sequence for n_frames x n_keypoints independent (frame, keypoint) items -- the
 * per-frame loop body of estimate_6d_pose_lm (reference AccumulatorSpace.py:566-659) for a whole batch.
 * Every buffer is given as base + frame * stride_frame + keypoint * stride_kp (bytes; stride 0 = shared,
 * e.g. one depth map per frame, one max_radii_dm entry per keypoint).  Same results as the single-item calls.
 */
typedef struct rcv_strided {
    void *ptr;
    int64_t stride_frame;
    int64_t stride_kp;
} rcv_strided;

/* rmax_dm: device float64 per item (max_radii_dm[k], AccumulatorSpace.py:540-549). */
int rcv_seam_compact_batch(rcv_strided seg, rcv_strided radial, rcv_strided depth, int H, int W, const double K[4], float seg_thr,
                           rcv_strided rmax_dm, int mode, rcv_strided xyz_m, rcv_strided radial_list, rcv_strided counts, int cap,
                           rcv_strided ws, size_t ws_bytes, int n_frames, int n_keypoints, void *stream);
/* counts (int32[2] per item, device) give the list lengths; cap bounds them.  radial lists are float32 unless
 * radial_is_f64. */
int rcv_accumulate_3d_batch(rcv_strided xyz_m, rcv_strided radial_dm, int radial_is_f64, rcv_strided counts, int cap,
                            double acc_unit_mm, int engine, rcv_strided result, rcv_strided ws, size_t ws_bytes, int n_frames,
                            int n_keypoints, void *stream);

/*
 * Horn absolute orientation, reference util/horn.py:75-181 (HornPoseFitting.lmshorn), as called on the three
 * keypoints of a frame at AccumulatorSpace.py:659-661: RT (4x4 row-major float64) with est ~ R * model + T.
 *   model_kpts_dev  float64 [n_points,3] (mm), shared by all frames
 *   est_kpts        float64 [3] per (frame, point): ptr + frame * stride_frame + point * stride_kp -- e.g. the
 *                   center_mm fields of the rcv_acc_result records (stride_kp = sizeof(rcv_acc_result))
 *   rt_out_dev      float64 [n_frames,4,4]
 */
int rcv_horn_pose_batch(const double *model_kpts_dev, rcv_strided est_kpts, int n_points, int n_frames, double *rt_out_dev,
                        void *stream);

/*
 * Depth map restricted to the union of the point masks of the keypoints of one frame -- the pixels behind the scene cloud
 * `xyz_mm_icp` that the reference hands to ICP (AccumulatorSpace.py:621-626).  seg / radial: n_keypoints float32 [H,W]
 * maps, kp_stride elements apart (seg may be NULL for mode 1); masks as rcv_seam_compact modes 0 / 1;
 * rmax_dm_dev: float64 [n_keypoints] on the device.  rcv_seam_compact(mode 2) on depth_out_dev yields the cloud (mm).
 */
int rcv_union_depth(const float *seg_dev, const float *radial_dev, long long kp_stride, const uint16_t *depth_dev, int H, int W,
                    float seg_thr, const double *rmax_dm_dev, int mode, int n_keypoints, uint16_t *depth_out_dev, void *stream);

/*
 * Nearest-neighbour distances and point-to-point ICP, the two Open3D 0.14.1 calls of the evaluation that follows the
 * keypoints.  Open3D is not part of the reference tree (rcvpose.yml:176 pins it); these entry
 * points restate its published algorithm (see rcvpose_amd/csrc/icp.hip).  Points are float64 [n,3], row-major.
 *
 * rcv_nn_distance: open3d.geometry.PointCloud.compute_point_cloud_distance, AccumulatorSpace.py:689,693,719,723:
 *   dist_dev[i] = min_j |a_i - b_j|; idx_dev (optional) = arg-min (lowest index among equals).  Empty b: RCV_E_EMPTY.
 */
int rcv_nn_distance(const double *a_dev, int n_a, const double *b_dev, int n_b, double *dist_dev, int *idx_dev, void *stream);

typedef struct rcv_icp_result {
    double transformation[16]; /* row-major 4x4, source -> target (open3d RegistrationResult.transformation) */
    double fitness;            /* correspondences / source points */
    double inlier_rmse;        /* sqrt(mean squared correspondence distance) */
    int32_t n_correspondences;
    int32_t iterations;        /* updates applied */
    int32_t status;
    int32_t reserved;
} rcv_icp_result;

/*
 * open3d.pipelines.registration.registration_icp(source, target, max_correspondence_distance, init,
 * TransformationEstimationPointToPoint(), ICPConvergenceCriteria(relative_fitness, relative_rmse, max_iteration)),
 * AccumulatorSpace.py:706-712 (defaults 1e-6, 1e-6, 30) and :935-943.  Synchronous (`out` is host memory): the convergence
 * test runs on the device, the host reads the loop's control words once per 32 enqueued iterations.  The neighbour search is a
 * hashed grid at max_corr_dist when that bins the target finely (its buckets come from the stream's memory pool,
 * hipMallocAsync / hipFreeAsync) and an exhaustive scan otherwise -- same correspondences either way; the environment variable
 * RCV_ICP_NO_GRID=1 forces the exhaustive scan (used by the tests to compare the two).
 * Workspace: rcv_icp_workspace_bytes(n_src) device bytes.
 */
size_t rcv_icp_workspace_bytes(int n_src);
int rcv_icp_point_to_point(const double *src_dev, int n_src, const double *tgt_dev, int n_tgt, double max_corr_dist,
                           const double T0[16], double relative_fitness, double relative_rmse, int max_iteration,
                           rcv_icp_result *out, void *ws, size_t ws_bytes, void *stream);

/* argwhere(VoteMap_3D == peak) in C order over a grid produced by rcv_accumulate_3d
 * (AccumulatorSpace.py:406): writes up to cap flat indices (ascending) and the total count. */
int rcv_collect_ties(const uint32_t *vote_grid_dev, int dim, uint32_t peak, int64_t *flat_idx_dev, int cap,
                     int32_t *count_dev, void *stream);

/* Test hook: the recentred + shifted voxel-space points fast_for receives (AccumulatorSpace.py:403),
 * recomputed from the workspace of the last rcv_accumulate_3d call.  points_vox_dev float64 [n,3]. */
int rcv_acc_debug_points(const void *ws, double *points_vox_dev, int32_t *radius_int_dev, int cap, void *stream);

/* ------------------------------------------------------------------------------------------
 * FCN path
 * ------------------------------------------------------------------------------------------ */

/* FCResBackbone's image preparation, reference AccumulatorSpace.py:140-151:
 * uint8 HWC -> float64 /255, -mean, /std -> float32 CHW.  rgb_dev uint8 [B,H,W,3] -> x_dev float32 [B,3,H,W]. */
int rcv_preprocess_rgb(const uint8_t *rgb_dev, int B, int H, int W, float *x_dev, void *stream);

/* Storage type of weights and activations (accumulation is always fp32):
 *   RCV_F32   the parity mode: radial maps within 1e-4 of the reference's fp32 forward
 *   RCV_BF16  the throughput mode BASELINE.json names (8 significand bits: map error 1.5-5.5e-2 of range)
 *   RCV_F16   IEEE half at the same speed (11 significand bits: map error ~3e-3; magnitudes beyond 65504 saturate)
 *   RCV_F16X3 the fast parity mode: every value is a pair of IEEE halves hi + lo (22 significand bits), a product is
 *             three half-precision MFMAs (hi*hi + hi*lo + lo*hi) accumulated in fp32: maps within 1e-4 of the reference's
 *             fp32 forward (measured ~5e-6, like RCV_F32) at a third of the fp16 matrix rate instead of a sixteenth.
 *             Storage: 4 bytes per element, NHWC with channels in groups of 32 as [32 hi | 32 lo]; weights per op
 *             [G][cout][kh][kw][cin/32][hi 32 | lo 32], each output channel scaled by a power of two 2^k (so that the lo
 *             halves are normal numbers) and biases per op [G][cout bias | cout scale 2^-k]; magnitudes beyond 65504 are
 *             clamped and reported (RCV_E_RANGE, rcv_fcn_range_status).  Packing: rcvpose_amd/fcn_engine.py::pack_weights. */
typedef enum rcv_dtype { RCV_F32 = 0, RCV_BF16 = 1, RCV_F16 = 2, RCV_F16X3 = 3 } rcv_dtype;

typedef enum rcv_op_kind {
    RCV_OP_STEM = 1,     /* conv1 7x7 s2 p3 (no bias) + BN + ReLU, NCHW fp32 in -> NHWC out; fcnresnet.py:54-57,123-125 */
    RCV_OP_MAXPOOL = 2,  /* 3x3 s2 p1; fcnresnet.py:58,126 */
    RCV_OP_CONV = 3,     /* KxK conv (+BN folded) [+ residual] [+ ReLU], one or two channel-concatenated sources */
    RCV_OP_UPSAMPLE2 = 4,/* bilinear x2, align_corners=False; fcnresnet.py:93 */
    RCV_OP_HEAD = 5      /* conv8 1x1 -> 2 channels, split into seg / radial planes (fp32 NCHW); fcnresnet.py:186-191 */
} rcv_op_kind;

/* One step of the layer program.  Buffers are numbered slots inside the activation workspace
 * (slot -1 = none, slot -2 = the network input x, slot -3 = the (seg, radial) output planes). */
typedef struct rcv_op {
    int32_t kind;
    int32_t src0, src1;      /* input slots (src1 = second concat source or -1) */
    int32_t res;             /* residual slot added before the ReLU, or -1 */
    int32_t dst;             /* output slot */
    int32_t c0, c1;          /* channels of src0 / src1 */
    int32_t cout;
    int32_t hin, win;        /* input spatial size */
    int32_t hout, wout;      /* output spatial size */
    int32_t ksize, stride, pad;
    int32_t relu;
    int64_t w_off;           /* element offset of this op's weights [G][cout][k][k][c0+c1] in the weight blob */
    int64_t b_off;           /* element offset of this op's fp32 bias [G][cout] in the bias blob */
    int64_t wq_off;          /* RCV_F16X3: element offset of a SECOND copy of this op's weights, or 0 = none.  For a 3x3 / stride-1 op:
                              * the q6 row layout (engine option 11, below).  For the conv7 op of the tail (the 3x3 behind the last
                              * upsampling): the parity-composed weights [G][4 * cout][3][3][cin] of engine option 13, whose
                              * [bias x 4 | scale] records sit behind the first copy's in the bias blob */
} rcv_op;

typedef struct rcv_fcn rcv_fcn; /* opaque engine handle */

/*
 * Build an engine for `groups` networks that share one topology and one input (the reference runs
 * one DenseFCNResNet152 per keypoint on the same image, AccumulatorSpace.py:520-530,596).
 *   ops / n_ops      the layer program (host memory, copied)
 *   n_slots, slot_elems[n_slots]: activation slots, sizes in elements PER GROUP AND PER IMAGE
 *   weights_dev      packed weights, dtype `dtype`, BN folded, layout per op [G][cout][kh][kw][cin]
 *   bias_dev         fp32 biases (BN folded), per op [G][cout]
 * The handle stores pointers and the program; the only device memory it owns is the 256-byte range-flag word of the
 * half-precision engines (below), allocated here, not on the hot path.
 * RCV_F16X3 engines need every tensor of the program to have a multiple of 32 channels (the network input and the head's two
 * planes excepted): RCV_E_UNSUPPORTED otherwise.
 */
int rcv_fcn_create(const rcv_op *ops, int n_ops, int n_slots, const int64_t *slot_elems, int groups, int dtype,
                   const void *weights_dev, const float *bias_dev, rcv_fcn **out);
void rcv_fcn_destroy(rcv_fcn *h);
/* Activation workspace bytes for a batch of B images. */
size_t rcv_fcn_workspace_bytes(const rcv_fcn *h, int B);
/*
 * DenseFCNResNet152.forward, reference models/fcnresnet.py:121-191, for all groups:
 *   x_dev    float32 [B,3,H,W] (normalised image, shared by the groups)
 *   seg_dev, radial_dev  float32 [G,B,1,H,W] (radial_dev float16 under engine option 8)
 */
int rcv_fcn_forward(rcv_fcn *h, const float *x_dev, int B, float *seg_dev, float *radial_dev, void *ws,
                    size_t ws_bytes, void *stream);
/*
 * Range guard of the half-precision engines.  The reference's forward is fp32 (models/fcnresnet.py:121-191) and has fp32's
 * range; RCV_F16X3 stores every activation as a pair of halves and CLAMPS magnitudes beyond 65504 (the stored pair is finite
 * and wrong), RCV_F16 turns them into infinities.  Every kernel that rounds to halves therefore ORs a bit into a flag word of
 * the engine when it stored such a value (bit 0: an RCV_F16X3 pair was clamped; bit 1: an RCV_F16 store overflowed):
 *   - rcv_fcn_forward copies the word to the host at the end of each forward (asynchronously, on `stream`) and returns
 *     RCV_E_RANGE without enqueuing anything once a COMPLETED forward has set it -- sticky until cleared;
 *   - rcv_fcn_range_status waits for `stream`, writes the word to *flags_out (may be null), clears it when `clear` != 0 and
 *     returns RCV_E_RANGE if it was set, RCV_OK otherwise.  Call it after a forward whose maps must be trusted (the Python
 *     mirrors do: models.fcnresnet forward, FCResBackbone, FramePipeline.read_results); use RCV_F32 for such a checkpoint.
 * RCV_F32 / RCV_BF16 engines have fp32's exponent range: always RCV_OK, flags 0.
 */
int rcv_fcn_range_status(rcv_fcn *h, unsigned *flags_out, int clear, void *stream);
/* Test hook: copy activation slot `slot` of group g (as stored: NHWC, engine dtype) to out_dev as fp32. */
int rcv_fcn_debug_slot(rcv_fcn *h, const void *ws, int B, int g, int slot, float *out_dev, void *stream);

/* Test hook (RCV_F16X3, engine option 11): rewrite n_rows 128-byte rows [32 hi halves | 32 lo halves] in place into q6 rows
 * [32 hi halves | 4 x (16 e2m3 codes, scale in byte 12)] -- the kernel the executor runs on the source of a q6 op. */
int rcv_fcn_debug_q6_pack(void *rows_dev, long long n_rows, void *stream);

/* Name and average device time (ms, HIP events on `stream`) of the slowest op kind in the last
 * profiled forward; used by bench.py for the roofline line.  Enable with rcv_fcn_set_profiling. */
int rcv_fcn_set_profiling(rcv_fcn *h, int enable);
/* Engine options (A-B tests; all default 1; 16-bit and split-precision engines -- in an RCV_F16X3 engine key 3 selects
 * k_tail_x3, key 4 the fused 1x1 expand + next 1x1 reduce kernel k_chain_x3 (keys 5 and 6 have no effect there), key 2 the
 * split-precision MFMA stem; every fused / fast kernel of that engine is bit-identical to the plain kernels as well, except the
 * MFMA stem against the direct fp32-input stem (5e-7)): key 1 = LDS-DMA convolution kernel where it applies
 * (k_conv_dma), key 2 = MFMA stem (k_stem_mfma; 0: direct kernel with fp32 input), key 3 = fused last upsampling + conv7
 * + head (k_tail; 0: three kernels, bit-identical output), key 4 = fused 3x3 + expand 1x1 of the stride-1 bottleneck blocks (layers 1-3)
 * (k_conv_block; 0: two kernels, bit-identical output), key 5 = chain the next block's 1x1 reduce into k_conv_block where it
 * follows directly (0: separate kernel, bit-identical output), key 6 = staggered schedule of the 1x1 phases of the chained
 * 256-wide block (0: all waves in lockstep, bit-identical output), key 7 = halo-patch kernel for the 3x3 / stride-1 layers whose
 * image tiles well (k_conv_halo; 0: k_conv_dma, bit-identical output; 2: also where the pixel rectangles fit the image badly -- tests).
 * key 8 (default 0, both dtypes): 1 = the radial plane is written as float16 [G,B,1,H,W] (radial_dev then points at 2-byte
 * elements; the stress configuration of BASELINE.json; feed it to rcv_seam_compact with RCV_SEAM_RADIAL_F16).  The value is
 * the float32 head output rounded to nearest-even float16.
 * key 9 (default 1, bf16): latency kernel (k_conv_lat: 64 x 64 tiles, eight LDS-DMA stages in flight) for launches with few output
 * tiles -- one or a few frames; 0: k_conv, bit-identical output.
 * key 10 (default 1, RCV_F16X3): the fused 1x1 expand + next 1x1 reduce kernel in its round-4 form (k_chain2_x3: no LDS
 * transposition, half chunks, epilogue between the MFMAs); 0: k_chain_x3; 2: k_chain3_x3 for the 256-wide block (four waves of 32
 * pixels, one per SIMD: half the LDS fragment traffic, measured 4 % slower); bit-identical output in every form.
 * key 11 (default 1, RCV_F16X3): fp6 cross terms.  For a 3x3 / stride-1 op whose rcv_op::wq_off names a q6 copy of its weights
 * (and whose source tensor feeds no other op) the two cross terms hi*lo + lo*hi of the split product run as ONE block-scaled
 * v_mfma_scale_f32_16x16x128_f8f6f4 on e2m3 operands instead of two half-precision MFMAs (hi*hi stays fp16): the source is
 * rewritten in place (k_q6_pack) into rows [32 hi halves | 4 x (16 e2m3 codes in 12 bytes, e8m0 block scale in byte 12, 3 zero
 * bytes)] for the blocks hi[0:16], hi[16:32], lo[0:16], lo[16:32] of each 32-channel group; the q6 weight rows are
 * [32 hi halves | lo[0:16], lo[16:32], hi[0:16], hi[16:32]] with the scale in byte 14 of each 16-byte chunk; block scale
 * 2^(floor(log2 max|block|) - 2), codes rounded to nearest even, saturating at 7.5.  The key only ENABLES the form: an op takes it
 * when its wq_off names a q6 copy, so the feature is opt-in per op through the program.  rcvpose_amd/fcn_engine.py packs q6 copies
 * only on request (FcnEngine(q6=True) / a regular expression over op names; its Q6_DEFAULT = the 44 stride-1 3x3 convolutions of
 * layers 2-4, `block[234].N.conv2`): measured map error against the fp32 engine 2-8e-5 on the bench networks (3.4e-5 -> 7.8e-5 on the
 * worst one; layers 2 + 4 alone: unchanged; every 3x3 layer incl. the decoder: 6e-5 and more per layer group, tools/q6_precision.py,
 * profiles/r05_q6_precision.txt) -- most of the 1e-4 tolerance, which is why the Python default is off.  0: three half-precision
 * MFMAs everywhere (the bit-identical family of kernels).
 * key 12 (default 0, 16-bit and split engines): latency option for one or a few frames in flight (the reference calls the model
 * frame by frame, AccumulatorSpace.py:596).  Launches of k_conv_lat with far fewer tiles than the chip has CUs deal their K steps
 * over S workgroups per tile (S depends on the layer's shape only) and a second kernel adds the S partial sums in a fixed order
 * and applies the epilogue: run-to-run identical results, within the engine's tolerance of -- but not bit-identical to -- the
 * unsplit kernels, so a frame's maps then depend on the batch size it travels in.  Switching it on allocates a 48 MB buffer
 * owned by the handle.
 * key 13 (default 1, RCV_F16X3): the tail (x2 bilinear upsampling -> conv7 3x3 -> head; reference models/fcnresnet.py:177-191) as ONE 3x3
 * convolution on the low-resolution tensor: both maps are linear, so conv7 on the upsampled image equals a convolution with 4 x 32
 * parity-stacked output channels whose weights the host composes (rcv_op::wq_off of the conv7 op: [G][128][3][3][cin] behind the first
 * copy, its [bias x 4 | scale] records behind the first copy's; rcvpose_amd/fcn_engine.py::compose_upsample_conv3).  k_conv_halo reads
 * the low-resolution tensor with replicate padding and applies ReLU + the head in a register-only epilogue (no upsampled tensor, no
 * conv7 activation ever exists in memory); the tiles that touch the image border -- the outermost ring of output pixels is where the
 * upsampling's clamp and the convolution's zero padding meet, which the composed form cannot express -- are recomputed by k_tail_x3
 * (ring_only).  Same function, another summation: within ~3e-6 of k_tail_x3, not its bits, and 4.9 ms against 6.2 per forward of
 * 53 x 3 (r06).  1: on images with at least 64 rectangles of 320 low-resolution pixels (a rule of the image size alone: a frame's maps
 * do not depend on the batch it travels in -- a batch beyond the 32-bit DMA range, ~108 frames at 480x640, runs in slices of images);
 * 2: wherever the halo tiling applies (tests); 0: k_tail_x3 everywhere (member of the bit-identical family).
 * key 14 (default 0 -- a measured experiment, r06; RCV_F16X3): the 4*CMID-channel trunk tensor between two chain kernels of layer 2 (written once, read
 * once, as a residual) in a 3-byte form (sign / 5-bit exponent / 18-bit mantissa, 96 instead of 128 bytes per 32-channel group): maps within
 * 6e-6 of the [hi | lo] form, 20 % fewer bytes, and SLOWER (9.5 against 7.2 ms per forward of 53 x 3: the encode / decode instructions).
 * key 15 (default 0 -- a measured experiment, r06; RCV_F16X3): the 128-wide chain kernel as four-wave workgroups (three per CU): bit-identical,
 * 7.9 against 7.3 ms. */
int rcv_fcn_set_option(rcv_fcn *h, int key, int value);
int rcv_fcn_get_profile(rcv_fcn *h, double *conv_ms, double *other_ms, double *conv_flop);
/* Per-op device times (ms) of the last profiled forward; ms_out has room for n_ops doubles. */
int rcv_fcn_get_op_times(rcv_fcn *h, double *ms_out, int n_ops);

#ifdef __cplusplus
}
#endif
#endif /* RCVPOSE_HIP_H */
