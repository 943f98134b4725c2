// accumulator.hip -- the voting half of the RCVPose hot path on MI355X (gfx950 / CDNA4).
//
//   rcv_seam_compact   : mask + back-projection seam      (reference AccumulatorSpace.py:603-610,619; :77-85)
//   rcv_accumulate_3d  : Accumulator_3D + fast_for + arg-max (reference AccumulatorSpace.py:373-419, :325-341)
//
// Design (MI355X-first, not a translation of the reference's N x D^3 loop):
//   * the hot path only needs argwhere(votes == votes.max()), so the default engine is an exact branch-and-bound
//     peak search (k_bb_*): box upper bounds in float32 over a point list streamed through LDS by LDS-DMA, exact
//     float64 counts for the probe / hill-climb lower bound and for the few surviving voxels.
//   * the general engine (and the one that materialises the grid) never needs the vote grid in HBM either: the
//     grid is cut into column tiles (TI x TJ x TK voxels) that fit the 160 KB LDS of a CU as 32-bit counters
//     (120 KB per tile); one 1024-thread workgroup owns a tile, culls the point list against the tile footprint, and
//     rasterises, per (point, column), only the few voxels of the spherical shell that cross that column.
//     Votes are LDS atomics; no global atomics exist anywhere on the path.
//   * the arg-max is taken per tile straight out of LDS with a wavefront reduction on the packed key
//     (votes << 32 | ~flat_index), so ties resolve to the first voxel in C order like np.argwhere.
//   * exactness: the reference decides `0 < round(r) - sqrt(d2) < sqrt(3)/4` in float64.  sqrt and the
//     subtraction are monotone, so for every integer radius the test is equivalent to
//     T_lo[r] <= d2 < T_hi[r] with two float64 thresholds that the host finds once by bisection using
//     the very same IEEE operations.  d2 is evaluated in float64 in the reference's operation order
//     with FMA contraction off, hence the vote counts are bit-identical to the sequential reference.
//   * the recentring mean reproduces numpy's chunked pairwise summation bit for bit (see k_setup).
#pragma clang fp contract(off)

#include <math.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#include "common.hpp"

namespace rcv {
namespace {

constexpr int TABLE_N = 8192;          // largest integer radius (voxels) with a threshold pair
constexpr int MAX_TILES = 1 << 17;
constexpr int MAX_REPLICAS = 2048;      // workgroups that share a tile park their counters here (CELL_BYTES each)
constexpr int BB_CAP = 1 << 19;         // candidate blocks per level of the branch-and-bound peak search
constexpr int BB_HALF = BB_CAP / 2;     // a level's list is two regions: [0, BB_HALF) x-sibling pairs, [BB_HALF, BB_CAP) single blocks
constexpr int BB_ROOT = 64;            // edge (voxels) of the root blocks; 6 halvings down to single voxels
constexpr int BB_LEVELS = 6;
// cost model of the two arg-max engines (measured on the bench cloud, N = 15.5k, radii 50 ... 94 voxels: tools/vote_sweep.py,
// profiles/r04_rasteriser_pmc.json): what the hand-over rules of k_bb_gate (r05) and d_bb_restart (r04) are derived from
constexpr double BB_PS_PER_BLOCK_POINT = 1.1, RAST_PS_PER_POINT_COLUMN = 4.8, BB_BLOCKS_TIMES_F2 = 500.0, RAST_DISC_FILL = 0.6;
constexpr double BB_FLAT_F_RMAX = 7.8;  // sqrt(BB_BLOCKS_TIMES_F2 * BB_PS_PER_BLOCK_POINT / (RAST_PS_PER_POINT_COLUMN * pi * RAST_DISC_FILL))
static_assert(BB_FLAT_F_RMAX * BB_FLAT_F_RMAX * RAST_PS_PER_POINT_COLUMN * 3.14159265 * RAST_DISC_FILL > 0.97 * BB_BLOCKS_TIMES_F2 * BB_PS_PER_BLOCK_POINT &&
              BB_FLAT_F_RMAX * BB_FLAT_F_RMAX * RAST_PS_PER_POINT_COLUMN * 3.14159265 * RAST_DISC_FILL < 1.03 * BB_BLOCKS_TIMES_F2 * BB_PS_PER_BLOCK_POINT,
              "BB_FLAT_F_RMAX follows from the four measured constants");
constexpr double BB_HANDOVER_R2 = 2.5;  // d_bb_restart gives up above BB_HANDOVER_R2 * r_max^2 four-voxel candidates (derivation there)
// Probe width and hill-climb length only steer how close the lower bound gets to the peak before the wide levels run
// (any exact count is a valid bound).  Measured on the bench workload (72 frames x 3 keypoints, seam + voting, ms):
// beam/rounds 8/8 26.1, 4/8 27.3, 2/8 83.6, 8/4 32.1, 8/12 25.7, 8/16 26.3, 4/12 24.8, 4/16 25.3, 3/16 25.4, 4/24 26.5.
// (Since the least-squares seed -- k_bb_seed -- two rounds are active on such frames and the probe is gated off; beam and
// rounds are what an item without a usable seed falls back on.)
#ifndef RCV_BB_BEAM
#define RCV_BB_BEAM 4
#endif
#ifndef RCV_BB_CLIMB_ROUNDS
#define RCV_BB_CLIMB_ROUNDS 12
#endif
constexpr int BB_SEED_THREADS = 1024;  // k_bb_seed: one workgroup per item
constexpr int BB_BEAM = RCV_BB_BEAM;   // blocks per level followed by the greedy probe that seeds the lower bound
constexpr int PREP_BLOCKS = 128;
constexpr int PREP_THREADS = 256;
constexpr int SEAM_PIX_PER_BLOCK = 1024;
constexpr int SEAM_MAX_BLOCKS = 4096;
constexpr int VOTE_THREADS = 1024;
constexpr int VOTE_WAVES = VOTE_THREADS / 64;
constexpr int CELL_BYTES = 120 * 1024;  // LDS bytes of vote counters per tile (32-bit; the 1024-thread workgroup owns its CU)
constexpr int MAX_DIM = 1600;          // flat index must fit 32 bits
constexpr int NPY_BUFSIZE = 8192;      // numpy ufunc buffer (elements): chunking of add.reduce
constexpr int PW_BLOCK = 128;          // numpy pairwise-sum leaf size

struct alignas(16) AccHeader {
    double mean[3];
    double rmax;
    double shift;     // (double)zero_boundary when negative, else 0
    double unit;
    int32_t n, m;
    int32_t zb, dim;
    int32_t TI, TJ, TK, TKp;  // tile footprint, k-extent, padded (even) cells per column (32-bit counters)
    int32_t nti, ntj, ntk, ntiles;
    int32_t wide_;    // (r01-r03: packed 16-bit counters below 65536 points; all counters are 32-bit since r04)
    int32_t nparts;   // every tile's point list is split over this many workgroups (load balance)
    int32_t status;
    uint32_t tile_counter;
    // exact branch-and-bound peak search (k_bb_*): state shared by its level kernels
    int32_t bb_enabled;      // 0: skip the search, the LDS rasteriser decides
    int32_t bb_done;         // set by k_bb_final when the search produced the result (rasteriser is skipped)
    int32_t bb_best;         // best exact vote count seen so far (lower bound of the peak)
    int32_t bb_overflow;     // a candidate list ran out of room -> fall back
    uint32_t bb_count[8];    // entries of the PAIR region of the candidate list consumed by level l (two per pair)
    uint32_t bs_count[8];    // entries of its SINGLES region (bs_count[0] = root blocks)
    uint32_t pb_count[8];    // same for the greedy probe (beam of BB_BEAM blocks per level)
    uint32_t sel_count[8];
    unsigned long long bb_best_key;  // bb_best << 32 | flat index of a voxel attaining it (hill-climb start)
    // RGB-only variant (rcv_accumulate_3d_no_depth): the points in P are final voxel coordinates (mean = shift = 0 above)
    // and the result is formed the reference's other way (AccumulatorSpace.py:468-477)
    int32_t variant;         // 0: Accumulator_3D, 1: Accumulator_3D_no_depth
    int32_t gate_best;       // diagnostics (tools/vote_sweep.py): bb_best when k_bb_gate(2) took the flat-landscape decision
    double nd_mean[3];       // x/y/z_mean_mm of :440-442
    unsigned long long climb_center[2];  // 1 + flat index of the voxel the hill-climb round (parity) was centred on; 0: none
    unsigned long long seed;             // 1 + flat index of the least-squares estimate of the keypoint's voxel (k_bb_seed); 0: none
    int32_t probe_skip;                  // the round of exact counts around the seed already found a strong peak: no greedy probe
    int32_t bb_handover;                 // the search gave up on cost: 1 = d_bb_restart (survivor count), 2 = k_bb_gate(2) (flat landscape predicted); the rasteriser decides
    // best key (votes << 32 | flat index) as of the END of hill-climb round r, in slot r & 1: round r + 1 centres on it.  Round r
    // only writes slot r & 1 and only reads the other one, so every workgroup of a round sees the same, frozen centre (r03;
    // before, each workgroup took the live bb_best_key at its own start and a late one could centre on a voxel found by its
    // own round -- harmless for exactness, but the lower bound and the list sizes became timing dependent)
    unsigned long long climb_key[2];
    unsigned long long restart;  // 1 + flat index of the centre voxel of the candidate block with the largest upper bound (k_bb_restart); 0: none
    int32_t box[2][6];           // diagnostics (tools/bb_stats.py): bounding box (i0, i1, j0, j1, k0, k1) of the 8- and 4-voxel candidates
};
static_assert(sizeof(AccHeader) <= 512, "AccHeader must fit the 512 bytes reserved in the workspace");

struct PrepPartial {
    double mn[3], mx[3], rmax;
    double radius_overflow;  // != 0: some integer radius has no threshold pair (>= TABLE_N)
};

struct Layout {
    size_t header, partials, seam_counts, seam_offsets, tile_best, tile_ties, bb_lists, bb_probe, Q, P, ri, F, capF, replicas, min_total, total;
};

Layout make_layout(int cap)
{
    Layout L;
    size_t o = 0;
    L.header = o;        o += 512;
    L.partials = o;      o += align_up(sizeof(PrepPartial) * PREP_BLOCKS, 256);
    L.seam_counts = o;   o += align_up(sizeof(int2) * SEAM_MAX_BLOCKS, 256);
    L.seam_offsets = o;  o += align_up(sizeof(int2) * SEAM_MAX_BLOCKS, 256);
    L.tile_best = o;     o += align_up(sizeof(uint64_t) * MAX_TILES, 256);
    L.tile_ties = o;     o += align_up(sizeof(uint32_t) * MAX_TILES, 256);
    L.bb_lists = o;      o += (size_t)2 * BB_CAP * 16;
    L.bb_probe = o;      o += 4096;  // probe lists: BB_BEAM selected blocks + their 8*BB_BEAM children, twice
    L.Q = o;             o += align_up(sizeof(double) * 3 * (size_t)cap, 256);  // final voxel-space coordinates (SoA)
    L.P = o;             o += align_up(sizeof(double) * 5 * (size_t)cap, 256);  // x, y, z, T_lo, T_hi per point (SoA)
    L.ri = o;            o += align_up(sizeof(int32_t) * (size_t)cap, 256);
    L.capF = align_up((size_t)cap, 1024);  // float32 copy for the wide search levels, padded to whole LDS chunks
    L.F = o;             o += align_up(sizeof(float) * 5 * L.capF, 256);  // x, y, z, T_lo - slack, T_hi + slack (SoA)
    // optional tail: LDS images of tiles whose point list is split over several workgroups of the rasteriser.  A
    // workspace may stop at min_total (batched callers: the items of a batch already fill the chip); whatever lies
    // beyond it is used as replica slots (k_setup sizes nparts from the bytes it is given).
    L.replicas = o;      L.min_total = o;
    o += (size_t)MAX_REPLICAS * CELL_BYTES;
    L.total = o;
    return L;
}

// ---------------------------------------------------------------------------------------------
// shell thresholds (host)
// ---------------------------------------------------------------------------------------------
double *g_thr_dev = nullptr;  // [TABLE_N][2] = {T_lo, T_hi}
std::mutex g_init_mutex;

inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

void build_thresholds(double *tab)
{
    // f = (3**0.5)/4, AccumulatorSpace.py:328; volatile keeps the host compiler from folding/fusing
    volatile double three = 3.0;
    const double f = sqrt(three) / 4.0;
    tab[0] = 0.0;
    tab[1] = 0.0;  // radius 0: `radius - distance > 0` is never true
    for (int r = 1; r < TABLE_N; r++) {
        const double R = (double)r;
        // T_hi = min{ d : sqrt(d) >= R }  (then `R - sqrt(d) > 0`  <=>  d < T_hi)
        double hi = R * R;
        for (;;) {
            double below = u2d(d2u(hi) - 1);
            volatile double s = sqrt(below);
            if (s >= R) hi = below; else break;
        }
        // T_lo = min{ d : (R - sqrt(d)) < f }  -- bisection on the (monotone) bit pattern
        uint64_t lo_u = d2u(0.0), hi_u = d2u(R * R);  // predicate false at lo_u, true at hi_u
        while (hi_u - lo_u > 1) {
            uint64_t mid = lo_u + (hi_u - lo_u) / 2;
            volatile double s = sqrt(u2d(mid));
            volatile double t = R - s;
            if (t < f) hi_u = mid; else lo_u = mid;
        }
        tab[2 * r] = u2d(hi_u);
        tab[2 * r + 1] = hi;
    }
}

int ensure_init()
{
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (g_thr_dev) return RCV_OK;
    double *host = (double *)malloc(sizeof(double) * 2 * TABLE_N);
    if (!host) return RCV_E_INVALID;
    build_thresholds(host);
    double *dev = nullptr;
    hipError_t e = hipMalloc((void **)&dev, sizeof(double) * 2 * TABLE_N);
    if (e == hipSuccess) e = hipMemcpy(dev, host, sizeof(double) * 2 * TABLE_N, hipMemcpyHostToDevice);
    free(host);
    if (e != hipSuccess) {
        set_hip_error(e, "threshold table upload", __FILE__, __LINE__);
        return RCV_E_HIP;
    }
    g_thr_dev = dev;
    return RCV_OK;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    return __longlong_as_double((long long)shfl_xor_u64((uint64_t)__double_as_longlong(v), m));
}
__device__ __forceinline__ double wave_min(double v) { for (int m = 32; m; m >>= 1) v = fmin(v, shfl_xor_f64(v, m)); return v; }
__device__ __forceinline__ double wave_max(double v) { for (int m = 32; m; m >>= 1) v = fmax(v, shfl_xor_f64(v, m)); return v; }
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{
    for (int m = 32; m; m >>= 1) { uint64_t o = shfl_xor_u64(v, m); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) { for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m); return v; }

__device__ __forceinline__ int clamp_to_int(double v)
{
    // int() of the reference truncates toward zero; clamp so the cast is defined
    if (!(v > -2.0e9)) return -2000000000;
    if (!(v < 2.0e9)) return 2000000000;
    return (int)v;
}

// ---------------------------------------------------------------------------------------------
// seam: ordered stream compaction of the masked pixels
// ---------------------------------------------------------------------------------------------
// radial maps may be stored as float32 (the FCN head's output) or float16 (BASELINE.json stress config); a
// float16 value takes part in the reference arithmetic as the float32 it converts to exactly
template <typename RT>
__device__ __forceinline__ void seam_flags(const float *seg, const RT *rad, const uint16_t *depth, int p, float thr,
                                           double rmax, int mode, bool &m_rad, bool &m_pt)
{
    if (mode == 2) {  // plain rgbd_to_point_cloud: depth.nonzero()
        m_rad = m_pt = depth[p] != 0;
        return;
    }
    if (mode == 3) {  // YCB-Video (:1050-1054): the segmentation alone; rmax carries factor_depth
        m_rad = seg[p] > thr;
        m_pt = m_rad && depth[p] != 0;
        return;
    }
    const float r = (float)rad[p];
    const bool rok = (double)r <= rmax;  // float32 array vs float64 scalar: compared in float64
    if (mode == 0) {
        m_rad = (seg[p] > thr) && rok;   // python float vs float32 array: compared in float32
        m_pt = m_rad && depth[p] != 0;
    } else {
        m_rad = rok && (r != 0.f) && depth[p] != 0;
        m_pt = m_rad;
    }
}

// Depth map restricted to the union of the keypoints' point masks of one frame: the pixels whose back-projection the
// reference collects into `xyz_mm_icp` (AccumulatorSpace.py:621-626: points of keypoint 1, then the not yet present
// points of keypoints 2 and 3).  Two pixels back-project to the same point only if they are the same pixel, so the
// union of points is the union of pixels; rcv_seam_compact(mode 2) on the result gives the scene cloud (row-major).
__global__ __launch_bounds__(256) void k_union_depth(const float *seg, const float *rad, long long kp_stride, const uint16_t *depth,
                                                     int HW, float thr, const double *rmax, int mode, int n_kp, uint16_t *out)
{
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        bool any = false;
        for (int k = 0; k < n_kp; k++) {
            bool mr, mp;
            seam_flags<float>(seg ? seg + (size_t)k * kp_stride : nullptr, rad + (size_t)k * kp_stride, depth, p, thr, rmax[k], mode, mr, mp);
            any = any || mp;
        }
        out[p] = any ? depth[p] : (uint16_t)0;
    }
}

template <typename RT>
__device__ __forceinline__ void d_seam_count(const float *seg, const RT *rad, const uint16_t *depth, int HW,
                                                    float thr, double rmax, int mode, int2 *blk_counts)
{
    __shared__ int sp[4], sr[4], sn[4];
    const int base = blockIdx.x * SEAM_PIX_PER_BLOCK;
    int cp = 0, cr = 0, cn = 0;
    if (mode == 0) {  // the hot path (LINEMOD seam), straight-line: both maps of all four pixels requested up front
        float sv[4], rv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = min(base + j * 256 + (int)threadIdx.x, HW - 1);
            sv[j] = seg[p];
            rv[j] = (float)rad[p];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = base + j * 256 + threadIdx.x;
            if (p < HW) {
                const bool mr = (sv[j] > thr) && ((double)rv[j] <= rmax);  // (seam_flags, mode 0)
                cr += mr;
                if (mr) cp += depth[p] != 0;
                // RCV_SEAM_COUNT_NONFINITE: NaN / Inf in the maps (the comparisons above drop such pixels without a trace)
                cn += !(fabsf(sv[j]) <= 3.402823466e38f) || !(fabsf(rv[j]) <= 3.402823466e38f);
            }
        }
    } else
    for (int j = 0; j < 4; j++) {
        int p = base + j * 256 + threadIdx.x;
        if (p < HW) {
            bool mr, mp;
            seam_flags(seg, rad, depth, p, thr, rmax, mode, mr, mp);
            cp += mp;
            cr += mr;
            cn += (mode == 3 && !isfinite(seg[p])) || (mode == 1 && !isfinite((float)rad[p]));
        }
    }
    cp = wave_sum_i32(cp);
    cr = wave_sum_i32(cr);
    cn = wave_sum_i32(cn);
    if (lane_id() == 0) { sp[wave_id()] = cp; sr[wave_id()] = cr; sn[wave_id()] = cn; }
    __syncthreads();
    // a block holds SEAM_PIX_PER_BLOCK = 1024 pixels: the radii count in bits 0..15 of .y, the non-finite count in bits 16..31
    if (threadIdx.x == 0)
        blk_counts[blockIdx.x] = make_int2(sp[0] + sp[1] + sp[2] + sp[3], (sr[0] + sr[1] + sr[2] + sr[3]) | ((sn[0] + sn[1] + sn[2] + sn[3]) << 16));
}

__device__ __forceinline__ void d_seam_scan(const int2 *blk_counts, int nb, int2 *blk_offsets, int32_t *counts_out, int want_nonfinite)
{
    __shared__ int2 s[1024];
    __shared__ int s_nf;
    const int t = threadIdx.x;
    if (t == 0) s_nf = 0;
    __syncthreads();
    int2 loc[4];
    int2 sum = make_int2(0, 0);
    int nf = 0;
    for (int j = 0; j < 4; j++) {
        int b = t * 4 + j;
        loc[j] = b < nb ? blk_counts[b] : make_int2(0, 0);
        nf += (int)((unsigned)loc[j].y >> 16);
        loc[j].y &= 0xFFFF;
        sum.x += loc[j].x;
        sum.y += loc[j].y;
    }
    if (nf) atomicAdd(&s_nf, nf);
    s[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int2 v = make_int2(0, 0);
        if (t >= off) v = s[t - off];
        __syncthreads();
        s[t].x += v.x;
        s[t].y += v.y;
        __syncthreads();
    }
    int2 run = make_int2(s[t].x - sum.x, s[t].y - sum.y);  // exclusive prefix of this thread's 4 blocks
    for (int j = 0; j < 4; j++) {
        int b = t * 4 + j;
        if (b < nb) blk_offsets[b] = run;
        run.x += loc[j].x;
        run.y += loc[j].y;
    }
    if (t == 1023) {
        counts_out[0] = s[t].x;
        counts_out[1] = s[t].y;
        if (want_nonfinite) {  // (s_nf is complete: every add precedes the scan's barriers)
            counts_out[2] = s_nf;
            counts_out[3] = max(counts_out[3], s_nf);  // sticky: the largest count since the caller last zeroed it
        }
    }
}

template <typename RT>
__device__ __forceinline__ void d_seam_scatter(const float *seg, const RT *rad, const uint16_t *depth, int HW,
                                                      int W, double fx, double fy, double cx, double cy, float thr,
                                                      double rmax, int mode, const int2 *blk_offsets, double *xyz_m,
                                                      float *radial_list, int cap)
{
    __shared__ int segp[16], segr[16];
    const int base = blockIdx.x * SEAM_PIX_PER_BLOCK;
    const int lane = lane_id(), wv = wave_id();
    bool mr[4], mp[4];
    int pre_p[4], pre_r[4];
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int j = 0; j < 4; j++) {
        int p = base + j * 256 + threadIdx.x;
        mr[j] = mp[j] = false;
        if (p < HW) seam_flags(seg, rad, depth, p, thr, rmax, mode, mr[j], mp[j]);
        uint64_t bp = __ballot(mp[j]), br = __ballot(mr[j]);
        pre_p[j] = __popcll(bp & lt);
        pre_r[j] = __popcll(br & lt);
        if (lane == 0) { segp[j * 4 + wv] = __popcll(bp); segr[j * 4 + wv] = __popcll(br); }
    }
    __syncthreads();
    const int2 off = blk_offsets[blockIdx.x];
    for (int j = 0; j < 4; j++) {
        int p = base + j * 256 + threadIdx.x;
        int before_p = 0, before_r = 0;
        for (int q = 0; q < j * 4 + wv; q++) { before_p += segp[q]; before_r += segr[q]; }
        if (mr[j] && mode != 2) {
            int dst = off.y + before_r + pre_r[j];
            if (dst < cap) radial_list[dst] = (float)rad[p];
        }
        if (mp[j]) {
            int dst = off.x + before_p + pre_p[j];
            if (dst < cap) {
                // rgbd_to_point_cloud (AccumulatorSpace.py:81-84) then xyz = xyz_mm/1000 (:619)
                // mode 3 (:1052,1057): depth_map = depth * mask / factor_depth first, and no division by 1000 afterwards
                const int v = p / W, u = p - v * W;
                const double zs = mode == 3 ? (double)depth[p] / rmax : (double)depth[p];
                const double xs = (((double)u - cx) * zs) / fx;
                const double ys = (((double)v - cy) * zs) / fy;
                const bool raw = mode == 2 || mode == 3;
                xyz_m[3 * (size_t)dst + 0] = raw ? xs : xs / 1000.0;
                xyz_m[3 * (size_t)dst + 1] = raw ? ys : ys / 1000.0;
                xyz_m[3 * (size_t)dst + 2] = raw ? zs : zs / 1000.0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// prepare: voxel-space points, integer radii, per-block extrema  (AccumulatorSpace.py:376,388,332)
// ---------------------------------------------------------------------------------------------
template <typename RT>
__device__ __forceinline__ void d_prep(const double *xyz_m, const RT *radial, const int32_t *counts,
                                                       int n_arg, int m_arg, int cap, double unit, const double *thr,
                                                       double *P, int32_t *ri, PrepPartial *partials)
{
    __shared__ int s_overflow;
    if (threadIdx.x == 0) s_overflow = 0;
    __syncthreads();
    __shared__ double red[7][PREP_THREADS / 64];
    int n = n_arg, m = m_arg;
    if (counts) { n = min(counts[0], cap); m = min(counts[1], cap); }
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF}, rmax = -INF;
    const RT unit_rt = (RT)unit;  // python int 5 is a weak scalar: float32 lists stay float32 (:388)
    const int top = n > m ? n : m;
    for (int i = blockIdx.x * PREP_THREADS + threadIdx.x; i < top; i += PREP_BLOCKS * PREP_THREADS) {
        if (i < n) {
            for (int c = 0; c < 3; c++) {
                double v = (xyz_m[3 * (size_t)i + c] * 1000.0) / unit;  // xyz*1000/acc_unit, :376
                P[(size_t)c * cap + i] = v;
                mn[c] = fmin(mn[c], v);
                mx[c] = fmax(mx[c], v);
            }
        }
        if (i < m) {
            RT rv = (radial[i] * (RT)100) / unit_rt;  // radial_list*100/acc_unit, :388
            double rd = (double)rv;
            if (rd > rmax) rmax = rd;
            if (i < n) {
                double rr = rint(rd);  // np.around: half to even (:332); exact for float32 values too
                const int r = (rd == rd) ? clamp_to_int(rr) : 0;
                ri[i] = r;
                // shell thresholds of this point's integer radius: vote <=> T_lo <= d2 < T_hi
                double tlo = 0.0, thi = 0.0;  // radius <= 0 never votes (`radius - distance > 0` fails)
                if (r >= TABLE_N) s_overflow = 1;
                else if (r > 0) { tlo = thr[2 * r]; thi = thr[2 * r + 1]; }
                P[3 * (size_t)cap + i] = tlo;
                P[4 * (size_t)cap + i] = thi;
            }
        }
    }
    double vals[7] = {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], rmax};
    for (int q = 0; q < 7; q++) {
        double v = q < 3 ? wave_min(vals[q]) : wave_max(vals[q]);
        if (lane_id() == 0) red[q][wave_id()] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        PrepPartial pp;
        for (int q = 0; q < 7; q++) {
            double v = red[q][0];
            for (int w = 1; w < PREP_THREADS / 64; w++) v = q < 3 ? fmin(v, red[q][w]) : fmax(v, red[q][w]);
            if (q < 3) pp.mn[q] = v; else if (q < 6) pp.mx[q - 3] = v; else pp.rmax = v;
        }
        pp.radius_overflow = s_overflow ? 1.0 : 0.0;
        partials[blockIdx.x] = pp;
    }
}

// numpy pairwise_sum leaf: n <= 128 elements (loops_utils.h.src), exact operation order
__device__ double pw_leaf(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

// setup: numpy-exact means, grid sizing, tile geometry.  One workgroup.
//   np.mean(xyz_mm[:,c]) (:381-383) = add.reduce over a strided float64 view: the ufunc machinery
//   feeds the inner loop chunks of 8192 elements, each chunk is summed with the pairwise recursion
//   (halves split at a multiple of 8, leaves <= 128 with 8 interleaved partial sums) and added to the
//   running total in order.  Here a chunk's recursion tree is laid out as a binary heap (<= 255 nodes):
//   one thread per heap node finds its (offset, length) by walking down from the root, leaves are
//   summed in parallel, inner nodes level by level; the chunk totals are then added sequentially.
// Sum of each of three arrays A + a * stride (a = 0..2, n elements each) exactly as numpy's add.reduce forms it (see above);
// all 1024 threads of the workgroup call it; the totals land in shared `total[3]`.
__device__ __forceinline__ void np_sums3(const double *A, size_t stride, int n, double *total)
{
    __shared__ double val[4][3][256];
    __shared__ unsigned char kind[4][256];  // 0 = absent, 1 = leaf, 2 = inner
    const int t = threadIdx.x, grp = t >> 8, h = t & 255;
    if (t < 3) total[t] = 0.0;
    __syncthreads();
    const int nchunks = n > 0 ? (n + NPY_BUFSIZE - 1) / NPY_BUFSIZE : 0;
    for (int c0 = 0; c0 < nchunks; c0 += 4) {
        const int c = c0 + grp;
        unsigned char kd = 0;
        if (c < nchunks && h >= 1) {
            const int off0 = c * NPY_BUFSIZE;
            int len = min(NPY_BUFSIZE, n - off0), o = 0;
            const int depth = 31 - __clz(h);
            bool exists = true;
            for (int b = depth - 1; b >= 0; b--) {
                if (len <= PW_BLOCK) { exists = false; break; }
                int n2 = len / 2;
                n2 -= n2 % 8;
                if ((h >> b) & 1) { o += n2; len -= n2; } else { len = n2; }
            }
            if (exists) {
                if (len <= PW_BLOCK) {
                    kd = 1;
                    for (int a = 0; a < 3; a++) val[grp][a][h] = pw_leaf(A + (size_t)a * stride + off0 + o, len);
                } else {
                    kd = 2;
                }
            }
        }
        kind[grp][h] = kd;
        __syncthreads();
        for (int d = 6; d >= 0; d--) {
            if (h >= (1 << d) && h < (2 << d) && kind[grp][h] == 2)
                for (int a = 0; a < 3; a++) val[grp][a][h] = val[grp][a][2 * h] + val[grp][a][2 * h + 1];
            __syncthreads();
        }
        if (t < 3) {
            double s = total[t];
            for (int g = 0; g < 4 && c0 + g < nchunks; g++) s = s + val[g][t][1];
            total[t] = s;
        }
        __syncthreads();
    }
}

// grid-dependent part of the header: tile geometry of the LDS rasteriser, root blocks of the peak search
__device__ __forceinline__ bool finish_header(AccHeader &H, long long dim, int n, bool overflow, int bb_enable, long long replica_slots)
{
    if (dim <= 0) { H.status = RCV_E_BADGRID; return false; }
    if (overflow) { H.status = RCV_E_TOO_LARGE; return false; }
    if (dim > MAX_DIM) { H.status = RCV_E_TOO_LARGE; H.dim = (int)min(dim, 2000000000ll); return false; }
    const int D = (int)dim;
    H.dim = D;
    H.wide_ = 0;
    const int cells = CELL_BYTES / 4;
    // tile: TJ = 16 (or D if smaller) columns across j; TI rows; full k extent when it fits
    int TJ = D < 16 ? D : 16, TI, TK = D, TKp = D | 1;  // odd stride: the lanes' columns fall into different LDS banks
    if ((long long)TJ * TKp > cells) {  // a single row of columns does not fit: split k
        TI = 1;
        TJ = D < 64 ? D : 64;
        TKp = (cells / TJ - 1) | 1;  // odd, <= cells / TJ
        TK = TKp;
    } else {
        TI = cells / (TJ * TKp);
        if (TI * TJ > 128) TI = 128 / TJ;  // two 64-column passes per point
        if (TI > D) TI = D;
        // prefer at least ~2 workgroups per CU worth of tiles as long as a wave still gets 64 columns
        while (TI > 4 && (long long)((D + TI - 1) / TI) * ((D + TJ - 1) / TJ) < 512) TI >>= 1;
    }
    H.TI = TI; H.TJ = TJ; H.TK = TK; H.TKp = TKp;
    H.nti = (D + TI - 1) / TI; H.ntj = (D + TJ - 1) / TJ; H.ntk = (D + TK - 1) / TK;
    const long long nt = (long long)H.nti * H.ntj * H.ntk;
    if (nt > MAX_TILES) { H.status = RCV_E_TOO_LARGE; return false; }
    H.ntiles = (int)nt;
    // the central tiles see every point, the outer ones few: split each tile's point list over nparts
    // workgroups (their LDS counters are summed by k_merge) so that no work item dominates the makespan
    H.nparts = (int)max(1ll, min(8ll, min((long long)MAX_REPLICAS, replica_slots) / nt));
    H.status = RCV_OK;
    H.tile_counter = 0;
    {
        const long long nb = (D + BB_ROOT - 1) / BB_ROOT;
        H.bb_enabled = bb_enable && nb * nb * nb <= BB_HALF;
        H.bs_count[0] = (uint32_t)(nb * nb * nb);
    }
    return true;
}

__device__ __forceinline__ void d_setup(const PrepPartial *partials, const double *P, const int32_t *counts,
                                                int n_arg, int m_arg, int cap, double unit, int bb_enable,
                                                long long replica_slots, AccHeader *hdr)
{
    __shared__ double total[3];
    int n = n_arg, m = m_arg;
    if (counts) { n = min(counts[0], cap); m = min(counts[1], cap); }
    const int t = threadIdx.x;
    np_sums3(P, (size_t)cap, n, total);
    // extrema of the per-block partials (k_prep), reduced by the first wave
    __shared__ double s_ext[8];
    if (t < 64) {
        const double INF_ = __longlong_as_double(0x7FF0000000000000ll);
        double v[8] = {INF_, INF_, INF_, -INF_, -INF_, -INF_, -INF_, 0.0};
        for (int b = t; b < PREP_BLOCKS; b += 64) {
            const PrepPartial pp = partials[b];
            for (int a = 0; a < 3; a++) { v[a] = fmin(v[a], pp.mn[a]); v[3 + a] = fmax(v[3 + a], pp.mx[a]); }
            v[6] = fmax(v[6], pp.rmax);
            v[7] = fmax(v[7], pp.radius_overflow);
        }
        for (int q = 0; q < 8; q++) {
            const double r = q < 3 ? wave_min(v[q]) : wave_max(v[q]);
            if (t == 0) s_ext[q] = r;
        }
    }
    __syncthreads();
    if (t != 0) return;

    __shared__ AccHeader sH;  // (thread 0's working copy: in LDS -- as a local it lived in scratch memory, 416 bytes per thread until r05)
    AccHeader &H = sH;
    memset(&H, 0, sizeof(H));
    H.n = n; H.m = m; H.unit = unit;
    if (n <= 0 || m <= 0) {
        H.status = RCV_E_EMPTY;  // reference: ValueError from .min() of an empty array (:390)
        *hdr = H;
        return;
    }
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF}, rmax = -INF;
    for (int a = 0; a < 3; a++) { mn[a] = s_ext[a]; mx[a] = s_ext[3 + a]; }
    rmax = s_ext[6];
    const bool overflow = s_ext[7] != 0.0;
    double qmin = INF, qmax = -INF;
    for (int a = 0; a < 3; a++) {
        H.mean[a] = total[a] / (double)n;                 // np.mean: sum / count
        qmin = fmin(qmin, mn[a] - H.mean[a]);             // min of (x - mean): rounding is monotone
        qmax = fmax(qmax, mx[a] - H.mean[a]);
    }
    H.rmax = rmax;
    const int zb = clamp_to_int(qmin - rmax) + 1;          // :394
    H.zb = zb;
    H.shift = zb < 0 ? (double)zb : 0.0;                    // :396-397
    const double pmax = zb < 0 ? qmax - (double)zb : qmax;
    const int length = clamp_to_int(pmax);                  // :399
    const long long dim = (long long)length + (long long)clamp_to_int(rmax);  // :401
    finish_header(H, dim, n, overflow, bb_enable, replica_slots);
    *hdr = H;
}

// Tiles are handed out heaviest first: a tile's load is the number of points whose disc covers it, which
// peaks at the grid centre.  Order index -> tile: rows from the central row outwards, inside a row from the
// central column outwards (zig-zag around the middle is a bijection of [0, n)).
__device__ __forceinline__ int zigzag(int o, int n)
{
    const int c = n / 2;
    return (o & 1) ? c - (o + 1) / 2 : c + o / 2;
}
__device__ __forceinline__ int tile_from_order(int order, const AccHeader &H)
{
    const int o2 = order / H.ntk, tk = order - o2 * H.ntk;
    const int oi = o2 / H.ntj, oj = o2 - oi * H.ntj;
    return (zigzag(oi, H.nti) * H.ntj + zigzag(oj, H.ntj)) * H.ntk + tk;
}

// ---------------------------------------------------------------------------------------------
// vote: one workgroup per tile, 32-bit counters in LDS, arg-max straight from LDS
//
// Per (point, column) the shell crosses the column in at most two k-intervals whose end points are
// z -/+ sqrt(T_hi - sxy) and z -/+ sqrt(T_lo - sxy).  The end points are evaluated in fp32 relative to
// floor(z) (absolute error < 4e-7 * (r + 1) voxels); every integer that is further than `eps` from all
// end points is decided by them alone (eps is > 5x that error bound and the float64 rounding of d2 is
// ~1e-12), so those votes are cast without any per-voxel arithmetic.  An integer within eps of an end
// point (probability ~1e-3 per column) is decided by the exact float64 predicate T_lo <= d2 < T_hi.
// The resulting counts are therefore identical to the reference's float64 loop.
//
// r04: the kernel issues VALU instructions on 0.76 of all SIMD cycles, so its time is its instruction count
// (110 per (point, column) before).  Now ~45: a column inside the rim band (T_lo <= sxy: one interval) is
// treated as two intervals that meet at z (inner root 0) -- the integer nearest z, if within eps, goes through
// the exact predicate like any other end point -- so there is one code path; the four end points are two
// packed fp32 additions; the "too close to an integer" tests are two packed subtractions and one maximum of
// absolute values; counters are 32-bit (the 1024-thread workgroup owns its CU, 120 of the 160 KB of LDS), so a
// vote is one shift-add and one ds_add_u32, and the second voxel of an interval is the same address with an
// immediate offset.  Columns are an odd number of cells apart (the lanes of a wave vote at similar k in 64 different
// columns: with the even stride 60 % of the LDS cycles were bank conflicts) and the waves of a workgroup take the
// 64-point batches of a tile from a counter instead of a fixed round-robin (survivors per batch vary: the static deal
// left the slowest wave 10 % over the mean).  Together: 1.30 -> 0.78 ms per 15.5k-point / D = 229 item.
// ---------------------------------------------------------------------------------------------
constexpr int NONE_K = -0x40000000;
typedef float vote_f2 __attribute__((ext_vector_type(2)));

// tile arg-max (first maximum in C order) + tie count + optional copy of the tile to the HBM grid.
// Waves walk columns, lanes walk k.  Must be called by all VOTE_THREADS threads after a barrier.
__device__ __forceinline__ void tile_epilogue(const uint32_t *cells, uint64_t *wred, int *wredi, const AccHeader &H, int tile,
                                              uint32_t *grid_out, uint64_t *tile_best, uint32_t *tile_ties)
{
    const int lane = lane_id(), wv = wave_id(), tid = threadIdx.x;
    const int D = H.dim, TI = H.TI, TJ = H.TJ, TK = H.TK, TKp = H.TKp, ncols = TI * TJ;
    const int tk = tile % H.ntk, tj = (tile / H.ntk) % H.ntj, ti = tile / (H.ntk * H.ntj);
    const int i0 = ti * TI, j0 = tj * TJ, k0 = tk * TK;
    const int i1 = min(i0 + TI, D), j1 = min(j0 + TJ, D), k1 = min(k0 + TK, D);
    uint64_t best = 0;
    for (int col = wv; col < ncols; col += VOTE_WAVES) {
        const int il = col / TJ, jl = col - il * TJ;
        const int i = i0 + il, j = j0 + jl;
        if (i >= i1 || j >= j1) continue;
        const uint32_t fbase = ((uint32_t)i * (uint32_t)D + (uint32_t)j) * (uint32_t)D + (uint32_t)k0;
        for (int kk = lane; kk < k1 - k0; kk += 64) {
            const uint32_t v = cells[col * TKp + kk];
            const uint64_t key = ((uint64_t)v << 32) | (uint64_t)(0xFFFFFFFFu - (fbase + kk));
            best = key > best ? key : best;
            if (grid_out) grid_out[fbase + kk] = v;
        }
    }
    best = wave_max_u64(best);
    if (lane == 0) wred[wv] = best;
    __syncthreads();
    best = wred[0];
    for (int q = 1; q < VOTE_WAVES; q++) best = wred[q] > best ? wred[q] : best;
    const uint32_t vmax = (uint32_t)(best >> 32);
    int ties = 0;
    for (int col = wv; col < ncols; col += VOTE_WAVES) {
        const int il = col / TJ, jl = col - il * TJ;
        if (i0 + il >= i1 || j0 + jl >= j1) continue;
        for (int kk = lane; kk < k1 - k0; kk += 64) ties += cells[col * TKp + kk] == vmax;
    }
    ties = wave_sum_i32(ties);
    if (lane == 0) wredi[wv] = ties;
    __syncthreads();
    if (tid == 0) {
        int tsum = 0;
        for (int q = 0; q < VOTE_WAVES; q++) tsum += wredi[q];
        tile_best[tile] = best;
        tile_ties[tile] = (uint32_t)tsum;
    }
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float readlane_f32(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// one point against one tile, wave-uniform
struct VotePoint {
    double x, y, w, thi;    // voxel coordinates of the point (fast_for's frame), T_hi - T_lo, T_hi
    float zpe, zme, lim;    // frac(z) + eps, frac(z) - eps, 1 - 2 eps
    int zi;                 // floor(z)
    int rlo, rhi;           // the tile's k-range relative to floor(z)
};

// one (point, column): the votes of the shell of the point along column (ci, cj) of the tile; `ey` = T_hi - (cj - y)^2 (shared by
// the columns of a lane that lie in one grid row), `cbyte` = byte offset of the column's cell k = 0 in the LDS image (may be
// negative: k0 > 0), `zsrc` / `lsrc` / `L`: where the exact path finds the point's z and T_lo.  The end points only have to be
// good to ~1e-6, so a = T_hi - sxy is two fused multiply-adds here; the exact path forms sxy in the reference's order.  (A straight-line
// version that computed the lane's two columns as interleaved chains before any vote measured 1.28 instead of 0.93 ms per item.)
__device__ __forceinline__ void vote_column(unsigned char *cells, double ci, double cj, double ey, int cbyte, const VotePoint &q,
                                            double zsrc, double lsrc, int L)
{
    const double dx = ci - q.x;
    const double a = __builtin_fma(-dx, dx, ey);
    if (!(a > 0.0)) return;
    const double b = a - q.w;  // T_lo - sxy; <= 0 in the rim band: inner root 0, the two intervals meet at z
    const float zh = __builtin_amdgcn_sqrtf((float)a);
    const float zl = __builtin_amdgcn_sqrtf(fmaxf((float)b, 0.f));
    // End points relative to floor(z), moved inwards by eps: lower interval [tA.x, tA.y], upper [tB.y, tB.x].  Sure votes are
    // the integers ceil(low) .. floor(high).  An integer within eps of an end point shows up as a rounded value that moved by
    // almost a full unit: |round(t) - t| > 1 - 2 eps.
    const vote_f2 zz = {zh, zl};
    const vote_f2 tA = vote_f2{q.zpe, q.zme} - zz;
    const vote_f2 tB = vote_f2{q.zme, q.zpe} + zz;
    const vote_f2 rA = {ceilf(tA.x), floorf(tA.y)};
    const vote_f2 rB = {floorf(tB.x), ceilf(tB.y)};
    const vote_f2 dA = rA - tA, dB = rB - tB;
    const bool unc = fmaxf(fmaxf(fabsf(dA.x), fabsf(dA.y)), fmaxf(fabsf(dB.x), fabsf(dB.y))) > q.lim;
    const int a0 = max((int)rA.x, q.rlo), a1 = min((int)rA.y, q.rhi);
    const int b0 = max((int)rB.y, q.rlo), b1 = min((int)rB.x, q.rhi);
    unsigned char *col = cells + (cbyte + q.zi * 4);
    const int na = a1 - a0, nb = b1 - b0;  // votes of the interval - 1
    // cap columns cross the shell in 1-2 voxels per interval: those are voted inline, the rest in a loop
    if (na >= 0) {
        uint32_t *c = (uint32_t *)(col + a0 * 4);
        atomicAdd(c, 1u);
        if (na >= 1) atomicAdd(c + 1, 1u);
    }
    if (nb >= 0) {
        uint32_t *c = (uint32_t *)(col + b0 * 4);
        atomicAdd(c, 1u);
        if (nb >= 1) atomicAdd(c + 1, 1u);
    }
    const int nmax = max(na, nb);
    if (nmax >= 2) {  // rim columns: both intervals are long, one loop serves both
        uint32_t *ca = (uint32_t *)(col + a0 * 4), *cb = (uint32_t *)(col + b0 * 4);
        for (int t = 2; t <= nmax; t++) {
            if (t <= na) atomicAdd(ca + t, 1u);
            if (t <= nb) atomicAdd(cb + t, 1u);
        }
    }
    if (unc) {  // some integer is too close to an end point: decide those with the exact float64 predicate
        // candidates: the integer just outside each sure range (at most one per end point, eps < 0.5).  Typically ONE lane of the
        // wave has ONE such end point: every candidate's block starts with the wave-wide question whether anybody has it.
        const double pz = readlane_f64(zsrc, L), tlo = readlane_f64(lsrc, L);
        const double dy = cj - q.y;
        const double sxy = dx * dx + dy * dy;  // (i-x)**2 + (j-y)**2, reference order (:337)
        const float moved[4] = {fabsf(dA.x), fabsf(dA.y), fabsf(dB.y), fabsf(dB.x)};
        const float edge[4] = {rA.x, rA.y, rB.y, rB.x};
        int seen[4] = {NONE_K, NONE_K, NONE_K, NONE_K};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const bool flagged = moved[t] > q.lim;
            if (!__builtin_amdgcn_ballot_w64(flagged)) continue;
            const int r = flagged ? (int)edge[t] + ((t & 1) ? 1 : -1) : NONE_K;  // below a lower end / above an upper end
            seen[t] = r;
            bool go = r != NONE_K && r >= q.rlo && r <= q.rhi;
#pragma unroll
            for (int e = 0; e < t; e++) go = go && r != seen[e];  // two end points within eps of the same integer
            go = go && !((r >= a0 && r <= a1) || (r >= b0 && r <= b1));  // already a sure vote (cannot happen in exact arithmetic)
            if (go) {
                const double dz = (double)(q.zi + r) - pz;
                const double d2 = sxy + dz * dz;  // ... + (k-z)**2
                if (d2 >= tlo && d2 < q.thi) atomicAdd((uint32_t *)(col + r * 4), 1u);
            }
        }
    }
}

// The waves of a workgroup are independent between the two barriers that frame a tile: wave w takes the
// 64-point batches w, w+16, ... of the point list, culls them against the tile footprint (one point per
// lane, loads of the next batch already in flight) and rasterises the survivors one after the other, the
// point broadcast to the wave with v_readlane, the lanes spread over the tile's (<= 128) columns.
__device__ __forceinline__ void d_vote(AccHeader *hdr, const double *P, int cap, unsigned char *replicas,
                                                       uint64_t *tile_best, uint32_t *tile_ties, uint32_t *grid_out,
                                                       unsigned long long grid_cells)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *cells = (uint32_t *)smem;
    uint64_t *wred = (uint64_t *)(smem + CELL_BYTES);                    // [VOTE_WAVES]
    int *wredi = (int *)(smem + CELL_BYTES + 8 * VOTE_WAVES);            // [VOTE_WAVES]
    int *s_tile_p = (int *)(smem + CELL_BYTES + 12 * VOTE_WAVES);
    int *s_next = s_tile_p + 1;  // next 64-point batch of the tile (the waves take batches as they finish: survivors per batch vary)

    const AccHeader H = *hdr;
    if (H.status != RCV_OK || H.bb_done) return;
    const int D = H.dim, n = H.n, TI = H.TI, TJ = H.TJ, TK = H.TK, TKp = H.TKp;
    const int lane = lane_id(), tid = threadIdx.x;
    const int ncols = TI * TJ;  // <= 128: two 64-column passes per point
    const int nwords = ncols * TKp;
    const bool flush = grid_out != nullptr && (unsigned long long)D * D * D <= grid_cells;
    const double *Px = P, *Py = P + (size_t)cap, *Pz = P + 2 * (size_t)cap, *Plo = P + 3 * (size_t)cap,
                 *Phi = P + 4 * (size_t)cap;
    const int nbatch = (n + 63) / 64;

    for (;;) {
        // tiles are handed out through one counter, heaviest (central) first
        if (tid == 0) {
            *s_tile_p = (int)atomicAdd(&hdr->tile_counter, 1u);
            *s_next = 0;
        }
        __syncthreads();
        const int torder = *s_tile_p;
        __syncthreads();
        if (torder >= H.ntiles * H.nparts) break;
        const int part = torder % H.nparts;
        const int tile = tile_from_order(torder / H.nparts, H);
        const int tk = tile % H.ntk, tj = (tile / H.ntk) % H.ntj, ti = tile / (H.ntk * H.ntj);
        const int i0 = ti * TI, j0 = tj * TJ, k0 = tk * TK;
        const int i1 = min(i0 + TI, D), j1 = min(j0 + TJ, D), k1 = min(k0 + TK, D);
        for (int w = tid; w < (nwords + 3) / 4; w += VOTE_THREADS) ((uint4 *)cells)[w] = uint4{0u, 0u, 0u, 0u};

        // this lane's two columns
        double ci[2], cj[2];
        int cbyte[2];
        bool cvalid[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int col = p * 64 + lane;
            const int il = col / TJ, jl = col - il * TJ;
            cvalid[p] = col < ncols && i0 + il < i1 && j0 + jl < j1;
            ci[p] = (double)(i0 + il);
            cj[p] = (double)(j0 + jl);
            cbyte[p] = (col * TKp - k0) * 4;
        }
        const bool same_j = 64 % TJ == 0;  // the lane's two columns lie in one grid row (j): they share (j - y)^2
        const double fi0 = (double)i0, fi1 = (double)(i1 - 1), fj0 = (double)j0, fj1 = (double)(j1 - 1);
        const double fk0 = (double)k0, fk1 = (double)(k1 - 1);
        __syncthreads();

        // software pipeline: raw loads of the next batch are issued before the current one is rasterised; batch q of this part
        // is batch part + nparts * q of the list, q from the tile's counter
        double nx = 0, ny = 0, nz = 0, nlo = 0, nhi = 0;
        auto grab = [&]() {
            int q = 0;
            if (lane == 0) q = atomicAdd(s_next, 1);
            return part + H.nparts * __builtin_amdgcn_readfirstlane(q);
        };
        int batch = grab();
        {
            const int p = batch * 64 + lane;
            if (p < n) { nx = Px[p]; ny = Py[p]; nz = Pz[p]; nlo = Plo[p]; nhi = Phi[p]; }
        }
        while (batch < nbatch) {
            // fast_for sees xyz_mm - mean - zero_boundary (AccumulatorSpace.py:384-386,397)
            const double x = (nx - H.mean[0]) - H.shift, y = (ny - H.mean[1]) - H.shift, z = (nz - H.mean[2]) - H.shift;
            const double tlo = nlo, thi = nhi;
            batch = grab();
            {
                const int p = batch * 64 + lane;
                nhi = 0;
                if (p < n) { nx = Px[p]; ny = Py[p]; nz = Pz[p]; nlo = Plo[p]; nhi = Phi[p]; }
            }
            // cull: the disc of radius < sqrt(T_hi) <= r around (x, y) against the tile footprint, the k-extent
            const double ex = fmax(fmax(fi0 - x, x - fi1), 0.0), ey = fmax(fmax(fj0 - y, y - fj1), 0.0);
            const double rr = ceil(sqrt(thi));
            const bool keep = thi > 0.0 && (ex * ex + ey * ey) * (1.0 - 1e-9) < thi && z + rr >= fk0 && z - rr <= fk1;
            const double zf = floor(z);
            const int zi = (int)zf;
            const float zfr = (float)(z - zf);
            const float eps = 2.0e-6f * ((float)rr + 8.f);
            const float zpe = zfr + eps, zme = zfr - eps, lim = 1.f - 2.f * eps;
            const double wd = thi - tlo;
            uint64_t mask = __ballot(keep);
            while (mask) {
                const int L = __ffsll((unsigned long long)mask) - 1;
                mask &= mask - 1;
                VotePoint q;
                q.x = readlane_f64(x, L); q.y = readlane_f64(y, L);
                q.w = readlane_f64(wd, L); q.thi = readlane_f64(thi, L);
                q.zpe = readlane_f32(zpe, L); q.zme = readlane_f32(zme, L); q.lim = readlane_f32(lim, L);
                q.zi = __builtin_amdgcn_readlane(zi, L);
                q.rlo = k0 - q.zi; q.rhi = k1 - 1 - q.zi;  // tile k-range relative to floor(z)
                const double dy0 = cj[0] - q.y;
                const double ey0 = __builtin_fma(-dy0, dy0, q.thi);
                double ey1 = ey0;
                if (!same_j) {
                    const double dy1 = cj[1] - q.y;
                    ey1 = __builtin_fma(-dy1, dy1, q.thi);
                }
                if (cvalid[0]) vote_column(smem, ci[0], cj[0], ey0, cbyte[0], q, z, tlo, L);
                if (cvalid[1]) vote_column(smem, ci[1], cj[1], ey1, cbyte[1], q, z, tlo, L);
            }
        }
        __syncthreads();

        if (H.nparts > 1) {
            // park this workgroup's share of the tile; k_merge adds the shares and takes the arg-max
            uint4 *dst = (uint4 *)(replicas + ((size_t)tile * H.nparts + part) * CELL_BYTES);
            const uint4 *src = (const uint4 *)cells;
            for (int w = tid; w < (nwords + 3) / 4; w += VOTE_THREADS) dst[w] = src[w];
        } else {
            tile_epilogue(cells, wred, wredi, H, tile, flush ? grid_out : nullptr, tile_best, tile_ties);
        }
        __syncthreads();
    }
}

// sums the nparts LDS images of every tile and takes its arg-max (only launched work when nparts > 1)
__device__ __forceinline__ void d_merge(const AccHeader *hdr, const unsigned char *replicas,
                                                        uint64_t *tile_best, uint32_t *tile_ties, uint32_t *grid_out,
                                                        unsigned long long grid_cells)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *cells = (uint32_t *)smem;
    uint64_t *wred = (uint64_t *)(smem + CELL_BYTES);
    int *wredi = (int *)(smem + CELL_BYTES + 8 * VOTE_WAVES);
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || H.nparts <= 1 || H.bb_done) return;
    const int nvec = (H.TI * H.TJ * H.TKp + 3) / 4;
    const bool flush = grid_out != nullptr && (unsigned long long)H.dim * H.dim * H.dim <= grid_cells;
    for (int tile = blockIdx.x; tile < H.ntiles; tile += gridDim.x) {
        const uint4 *src = (const uint4 *)(replicas + (size_t)tile * H.nparts * CELL_BYTES);
        for (int w = threadIdx.x; w < nvec; w += VOTE_THREADS) {
            uint4 acc = src[w];
            for (int q = 1; q < H.nparts; q++) {
                const uint4 v = src[(size_t)q * (CELL_BYTES / 16) + w];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            ((uint4 *)cells)[w] = acc;
        }
        __syncthreads();
        tile_epilogue(cells, wred, wredi, H, tile, flush ? grid_out : nullptr, tile_best, tile_ties);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Exact branch-and-bound search of the arg-max (the common case: one sharp peak)
//
// The hot path only needs argwhere(VoteMap == VoteMap.max()) (AccumulatorSpace.py:406), not the map.
// For an axis-aligned block of voxels, a point can vote inside it only if its shell can reach the
// block: dmin^2 < T_hi and dmax^2 >= T_lo (box distances).  The number of such points is an UPPER bound
// of every voxel count in the block; the exact count of one voxel of the block (its centre) is a LOWER
// bound of the peak.  Blocks whose upper bound is below the best lower bound cannot hold a maximum
// (nor a tie) and are dropped; the others are split in 8 down to single voxels, where the bound is the
// exact reference predicate.  The voxel attaining the maximum is never dropped (its ancestors' bounds
// are >= the peak >= every lower bound), so peak, tie count and first index are exact.  Cost: a few
// 10^7 box tests instead of ~10^8.5 shell/column evaluations when the peak is sharp.  When it is not
// (lists overflow, or nothing votes) the LDS rasteriser below decides; it is also the path that
// produces the full grid.  All arithmetic is float64; the bound tests carry a 1e-12 relative slack.
// ---------------------------------------------------------------------------------------------
struct BBEntry {
    uint16_t i, j, k, pad_;
    int32_t ub, lb;
};
static_assert(sizeof(BBEntry) == 16, "BBEntry");

// conservative float32 slack on squared distances: coordinates lie in [0, D), block edges within 2 * BB_ROOT of
// them, so every per-axis distance is below D + 2 * BB_ROOT and the sum of the three below 3 * (D + 2 * BB_ROOT)
__device__ __forceinline__ float bb_f32_slack(int D)
{
    return 1.0e-6f * (float)(D + 2 * BB_ROOT) * 3.f * (float)(D + 2 * BB_ROOT) + 1e-5f;
}

__device__ __forceinline__ void d_bb_init(const AccHeader *hdr, const double *P, int cap, double *Q, BBEntry *roots, float *F,
                                          size_t capF)
{
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled) return;
    const int nb = (H.dim + BB_ROOT - 1) / BB_ROOT;
    const int nroot = nb * nb * nb;
    const int npad = (H.n + 1023) & ~1023;  // the wide levels stream whole 1024-point chunks
    const int top = npad > nroot ? npad : nroot;
    const float slack = bb_f32_slack(H.dim);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < top; t += gridDim.x * 256) {
        if (t < H.n) {  // fast_for sees xyz_mm - mean - zero_boundary (AccumulatorSpace.py:384-386,397)
            double q[3];
            for (int a = 0; a < 3; a++) {
                q[a] = (P[(size_t)a * cap + t] - H.mean[a]) - H.shift;
                Q[(size_t)a * cap + t] = q[a];
                F[(size_t)a * capF + t] = (float)q[a];
            }
            const double tlo = P[3 * (size_t)cap + t], thi = P[4 * (size_t)cap + t];
            F[3 * capF + t] = (float)tlo * (1.f - 1e-6f) - slack;
            F[4 * capF + t] = thi > 0.0 ? (float)thi * (1.f + 1e-6f) + slack : -1.f;  // < 0: never counts
        } else if (t < npad) {
            for (int a = 0; a < 4; a++) F[(size_t)a * capF + t] = 0.f;
            F[4 * capF + t] = -1.f;
        }
        if (t < nroot) {
            BBEntry e;
            e.i = (uint16_t)((t / (nb * nb)) * BB_ROOT);
            e.j = (uint16_t)(((t / nb) % nb) * BB_ROOT);
            e.k = (uint16_t)((t % nb) * BB_ROOT);
            e.pad_ = 0;
            e.ub = 0x7FFFFFFF;
            e.lb = 0;
            roots[t] = e;
        }
    }
}

// One level of the search: every surviving parent block is split in 8 children; a workgroup scores the
// children of one parent against all points.
//   upper bound  (levels above voxels): float32 box distances relative to the parent origin, widened by an
//                absolute slack that covers the float32 rounding (so it never under-counts)
//   lower bound  (WITH_LB: probe and voxel level): exact float64 reference predicate on the child's centre
//                voxel; at the voxel level it IS the vote count.
// Points that cannot reach the parent at all are compacted away first.  NT = 1024: one parent finishes fast
// (few parents: latency matters).
constexpr int BB_CHUNK = 4096;

template <bool LAST, bool WITH_LB, int NT>
__device__ __forceinline__ void d_bb_level(AccHeader *hdr, const double *Q, const double *P, int cap,
                                                 const BBEntry *in_list, const uint32_t *in_count, int in_cap,
                                                 BBEntry *out_list, uint32_t *out_count, int out_cap, int level)
{
    __shared__ int red[NT / 64][16];
    __shared__ int s_skip, s_cnt;
    __shared__ int s_idx[BB_CHUNK];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled) return;
    const int D = H.dim, n = H.n;
    const int S = BB_ROOT >> level, s = S >> 1;  // parent / child edge
    const unsigned nin = min(*in_count, (uint32_t)in_cap);
    const double *Qx = Q, *Qy = Q + (size_t)cap, *Qz = Q + 2 * (size_t)cap;
    const double *Plo = P + 3 * (size_t)cap, *Phi = P + 4 * (size_t)cap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float slack_k = 1.0e-6f * (float)(D + 2 * BB_ROOT);  // float32 rounding of coordinates up to D + block
    for (unsigned pi = blockIdx.x; pi < nin; pi += gridDim.x) {
        const BBEntry e = in_list[pi];
        __syncthreads();
        if (tid == 0)  // one decision per workgroup (the bound moves while other workgroups run)
            s_skip = e.ub < __hip_atomic_load(&hdr->bb_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
                     __hip_atomic_load(&hdr->bb_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_skip) continue;
        const int org[3] = {e.i, e.j, e.k};
        const double forg[3] = {(double)e.i, (double)e.j, (double)e.k};
        // the two halves of the parent along each axis, relative to the parent origin: voxel range [lo, hi], centre c
        float lo[3][2], hi[3][2];
        double ce[3][2];
        bool ok[3][2];
        float phi_[3];  // the parent's own upper corner (for the compaction test; its lower corner is 0)
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int l = org[a] + h * s, u = min(l + s - 1, D - 1);
                ok[a][h] = l <= D - 1;
                lo[a][h] = (float)(l - org[a]);
                hi[a][h] = (float)(u - org[a]);
                ce[a][h] = (double)min(l + s / 2, D - 1);
            }
            phi_[a] = (float)(min(org[a] + S - 1, D - 1) - org[a]);
        }
        int ub[8], lb[8];
#pragma unroll
        for (int c = 0; c < 8; c++) { ub[c] = 0; lb[c] = 0; }

        for (int base = 0; base < n; base += BB_CHUNK) {
            const int cend = min(base + BB_CHUNK, n);
            __syncthreads();
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            // pass 1: points of this chunk whose shell can reach the parent block
            for (int p0 = base + wv * 64; p0 < cend; p0 += NT) {
                const int p = p0 + lane;
                bool keep = false;
                if (p < cend) {
                    const float thi = (float)Phi[p] * (1.f + 1e-6f);
                    if (thi > 0.f) {
                        const float tlo = (float)Plo[p] * (1.f - 1e-6f);
                        const float qr[3] = {(float)(Qx[p] - forg[0]), (float)(Qy[p] - forg[1]), (float)(Qz[p] - forg[2])};
                        float smin = 0.f, smax = 0.f, asum = 0.f;
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            const float e1 = -qr[a], e2 = qr[a] - phi_[a];
                            const float dmin = fmaxf(fmaxf(e1, e2), 0.f), dmax = fmaxf(fabsf(e1), fabsf(e2));
                            smin += dmin * dmin; smax += dmax * dmax; asum += dmax;
                        }
                        const float slack = slack_k * asum + 1e-5f;
                        keep = (smin - slack < thi) && (smax + slack >= tlo);
                    }
                }
                const uint64_t m = __ballot(keep);
                int wbase = 0;
                if (lane == 0 && m) wbase = atomicAdd(&s_cnt, __popcll(m));
                wbase = __shfl(wbase, 0);
                if (keep) s_idx[wbase + __popcll(m & ((1ull << lane) - 1ull))] = p;
            }
            __syncthreads();
            const int cnt = s_cnt;
            // pass 2: the 8 children against the compacted points
            for (int t = tid; t < cnt; t += NT) {
                const int p = s_idx[t];
                const double q[3] = {Qx[p], Qy[p], Qz[p]};
                const double thi_d = Phi[p], tlo_d = Plo[p];
                if (!LAST) {
                    const float thi = (float)thi_d * (1.f + 1e-6f), tlo = (float)tlo_d * (1.f - 1e-6f);
                    float mn2[3][2], mx2[3][2], ax[3][2];
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const float qr = (float)(q[a] - forg[a]);
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const float e1 = lo[a][h] - qr, e2 = qr - hi[a][h];
                            const float dmin = fmaxf(fmaxf(e1, e2), 0.f), dmax = fmaxf(fabsf(e1), fabsf(e2));
                            mn2[a][h] = dmin * dmin; mx2[a][h] = dmax * dmax; ax[a][h] = dmax;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int hx = (c >> 2) & 1, hy = (c >> 1) & 1, hz = c & 1;
                        const float smin = mn2[0][hx] + mn2[1][hy] + mn2[2][hz];
                        const float smax = mx2[0][hx] + mx2[1][hy] + mx2[2][hz];
                        const float slack = slack_k * (ax[0][hx] + ax[1][hy] + ax[2][hz]) + 1e-5f;
                        ub[c] += (smin - slack < thi) && (smax + slack >= tlo);
                    }
                }
                if (WITH_LB && LAST) {
                    double c2[3][2];
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const double dc = ce[a][h] - q[a];  // (i - x), reference operand order (:337)
                            c2[a][h] = dc * dc;
                        }
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int hx = (c >> 2) & 1, hy = (c >> 1) & 1, hz = c & 1;
                        const double d2 = (c2[0][hx] + c2[1][hy]) + c2[2][hz];  // exact reference d2 of the centre voxel
                        lb[c] += d2 >= tlo_d && d2 < thi_d;
                    }
                }
                if (WITH_LB && !LAST) {
                    // above the voxel level the centre count only ranks blocks for the probe (it breaks the ties among the many
                    // blocks every shell crosses); it is not used as a bound, so float32 is enough
                    const float thi = (float)thi_d, tlo = (float)tlo_d;
                    float c2[3][2];
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const float dc = (float)(ce[a][h] - forg[a]) - (float)(q[a] - forg[a]);
                            c2[a][h] = dc * dc;
                        }
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int hx = (c >> 2) & 1, hy = (c >> 1) & 1, hz = c & 1;
                        const float d2 = (c2[0][hx] + c2[1][hy]) + c2[2][hz];
                        lb[c] += d2 >= tlo && d2 < thi;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int u = wave_sum_i32(LAST ? lb[c] : ub[c]), l = wave_sum_i32(lb[c]);
            if (lane == 0) { red[wv][c] = u; red[wv][8 + c] = l; }
        }
        __syncthreads();
        if (tid < 8) {
            const int c = tid, hx = (c >> 2) & 1, hy = (c >> 1) & 1, hz = c & 1;
            int u = 0, l = 0;
            for (int w = 0; w < NT / 64; w++) { u += red[w][c]; l += red[w][8 + c]; }
            if (ok[0][hx] && ok[1][hy] && ok[2][hz]) {
                int best = __hip_atomic_load(&hdr->bb_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (WITH_LB && LAST) {  // exact counts only: a lower bound of the peak and a start for the hill-climb
                    best = max(atomicMax(&hdr->bb_best, l), l);
                    const unsigned long long flat = ((unsigned long long)ce[0][hx] * D + (unsigned long long)ce[1][hy]) * D +
                                                    (unsigned long long)ce[2][hz];
                    atomicMax(&hdr->bb_best_key, ((unsigned long long)l << 32) | flat);
                }
                if (u > 0 && u >= best) {
                    const unsigned slot = atomicAdd(out_count, 1u);
                    if (slot < (unsigned)out_cap) {
                        BBEntry o;
                        o.i = (uint16_t)(org[0] + hx * s); o.j = (uint16_t)(org[1] + hy * s); o.k = (uint16_t)(org[2] + hz * s);
                        o.pad_ = 0; o.ub = u; o.lb = l;
                        out_list[slot] = o;
                    } else {
                        atomicExch(&hdr->bb_overflow, 1);
                    }
                }
            }
        }
    }
}

// Wide levels of the search (upper bounds only).  A 256-thread workgroup scores the children of BBW_NP parents per
// pass over the point list; the list is the float32 copy written by k_bb_init (slack folded into the thresholds,
// padded to whole chunks) and arrives in LDS by LDS-DMA (global_load_lds, 16 B per lane), double buffered: the
// chunk after the current one is in flight while this one is scored, one barrier per chunk.  Same conservative
// float32 box test as k_bb_level (absolute voxel coordinates).
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr int BBW_THREADS = 256;
constexpr int BBW_CHUNK = 1024;
constexpr int BBW_NP = 4;

__device__ __forceinline__ void bbw_dma16(const float *gptr, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gptr,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// One child pair of the inner loop: the two z halves of child (hx, hy) of parent u as packed sign tests.  A point reaches a
// child iff below < 0 and reach < 0; the two cannot both be positive (smin >= T_hi > T_lo >= smax >= smin is impossible), so
// "both negative" is "(-below) * reach is negative" -- ONE packed multiply for the two z halves instead of two v_and_b32, then
// one v_alignbit_b32 per child shifts the product's sign bit into the child's vote history (r04: 3 instead of 4 instructions per
// child pair).  The zero cases agree with the AND form: below = +0 (d2 == T_hi) gives -0 * negative = +0, reach = +0 gives +0;
// a product that underflows is -0, sign set, and both factors were negative.
#define RCV_BBW_TAIL(u, c, below, reach)                                                                                   \
    {                                                                                                                      \
        const f32x2 pr_ = (-(below)) * (reach);                                                                            \
        hist[u][2 * (c)] = __builtin_amdgcn_alignbit(hist[u][2 * (c)], (unsigned)__float_as_int(pr_.x), 31);               \
        hist[u][2 * (c) + 1] = __builtin_amdgcn_alignbit(hist[u][2 * (c) + 1], (unsigned)__float_as_int(pr_.y), 31);       \
    }

// VOX: the children are single voxels (last level).
// PAIRED_IN: the input list holds x-adjacent pairs of sibling blocks (entries 2q, 2q+1: same j, k, i one block apart); a pair
//   shares the y and z terms of every point: 43 instead of 54 instructions per parent-point.  60-80 % of the survivors of a
//   level have their x-sibling surviving too, so every level list is kept as two regions -- pairs and single blocks -- and each
//   region gets its own (compile-time specialised) launch.
// PAIR_OUT: the surviving children are sorted into the two regions of the next level's list; otherwise all go to one list.
template <bool VOX, bool PAIRED_IN, bool PAIR_OUT>
__device__ __forceinline__ void d_bb_wide(AccHeader *hdr, const float *F, size_t capF,
                                                         const BBEntry *in_list, const uint32_t *in_count, int in_cap,
                                                         BBEntry *out_list, uint32_t *out_pair_count, uint32_t *out_single_count,
                                                         int out_single_off, int out_cap, int level)
{
    static_assert(!(VOX && PAIRED_IN), "the voxel level takes single blocks");
    __shared__ __attribute__((aligned(16))) float sbuf[2][5][BBW_CHUNK];  // [stage][x, y, z, T_lo, T_hi][point]
    __shared__ int red[BBW_THREADS / 64][8 * BBW_NP];
    __shared__ int s_valid[BBW_NP], s_org[BBW_NP][3];
    static_assert(BBW_NP == 4, "two pairs per pass");
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled) return;
    const int D = H.dim, n = H.n;
    const int S = BB_ROOT >> level, s = S >> 1;
    const unsigned nin = min(*in_count, (uint32_t)in_cap);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int PT = BBW_CHUNK / BBW_THREADS;
    static_assert(32 % PT == 0, "vote-bit histories are counted every 32 points per thread");
    const int nchunks = (n + BBW_CHUNK - 1) / BBW_CHUNK;
    // wave wv brings floats [wv * 256, wv * 256 + 256) of each of the five arrays of a chunk
    const float *src = F + wv * 256 + lane * 4;
#define RCV_BBW_ISSUE(chunk, stage)                                                                        \
    {                                                                                                      \
        _Pragma("unroll") for (int a = 0; a < 5; a++)                                                      \
            bbw_dma16(src + (size_t)a * capF + (size_t)(chunk) * BBW_CHUNK, &sbuf[stage][a][wv * 256]);    \
    }
    if (blockIdx.x * BBW_NP >= nin) return;
    int g = 0;  // running chunk number of this workgroup: chunk g sits in stage g & 1
    RCV_BBW_ISSUE(0, 0);
    for (unsigned pp = blockIdx.x; pp * BBW_NP < nin; pp += gridDim.x) {
        BBEntry e[BBW_NP];
#pragma unroll
        for (int u = 0; u < BBW_NP; u++) e[u] = in_list[min(pp * BBW_NP + u, nin - 1)];
        __syncthreads();
        if (tid < BBW_NP) {
            const int best = __hip_atomic_load(&hdr->bb_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ovf = __hip_atomic_load(&hdr->bb_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const BBEntry me = in_list[min(pp * BBW_NP + tid, nin - 1)];
            s_valid[tid] = pp * BBW_NP + tid < nin && me.ub >= best && !ovf;
            s_org[tid][0] = me.i; s_org[tid][1] = me.j; s_org[tid][2] = me.k;
        }
        __syncthreads();
        bool valid[BBW_NP], any = false;
#pragma unroll
        for (int u = 0; u < BBW_NP; u++) { valid[u] = s_valid[u] != 0; any = any || valid[u]; }
        if (!any) continue;  // (workgroup-uniform) chunk 0 stays in stage g & 1 for the next pass
        // block ranges: x per parent; y, z per parent -- or, for pairs, per pair (held in the first parent's slot)
        f32x2 lo2[BBW_NP][3], hi2[BBW_NP][3];
#pragma unroll
        for (int u = 0; u < BBW_NP; u++) {
            const int org[3] = {e[u].i, e[u].j, e[u].k};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (PAIRED_IN && a > 0 && (u & 1)) continue;
                const int l0 = org[a], l1 = org[a] + s;
                lo2[u][a] = f32x2{(float)l0, (float)l1};
                hi2[u][a] = f32x2{(float)min(l0 + s - 1, D - 1), (float)min(l1 + s - 1, D - 1)};
            }
        }
        int ub[BBW_NP][8];
        unsigned hist[BBW_NP][8];  // per child: the last <= 32 vote bits, one shifted in per point, counted every 8 chunks
#pragma unroll
        for (int u = 0; u < BBW_NP; u++)
#pragma unroll
            for (int c = 0; c < 8; c++) { ub[u][c] = 0; hist[u][c] = 0u; }
        for (int ch = 0; ch < nchunks; ch++, g++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk g have landed
            __syncthreads();                                  // ... and everybody's; stage (g + 1) & 1 is free again
            RCV_BBW_ISSUE(ch + 1 < nchunks ? ch + 1 : 0, (g + 1) & 1);  // next chunk (or chunk 0 of the next pass)
            const float(*sb)[BBW_CHUNK] = sbuf[g & 1];
#pragma unroll 1
            for (int t = 0; t < PT; t++) {
                const int li = t * BBW_THREADS + tid;
                const float q[3] = {sb[0][li], sb[1][li], sb[2][li]};
                const float tlo = sb[3][li], thi = sb[4][li];
                const f32x2 thi2 = {thi, thi}, tlo2 = {tlo, tlo};
                // dmin / dmax of both halves of one axis of parent u as float2 (v_pk_add_f32); max(e1^2, e2^2) = max(|e1|, |e2|)^2
                // exactly (the |.| are free source modifiers).  r04: the squares are not formed on their own any more -- every sum
                // they enter is one v_pk_fma_f32 (dmin * dmin + rest, rest - dmax * dmax): 16 multiplies less per four parents and
                // point, and one rounding less per term, so the thresholds' float32 slack covers the sums as before
#define RCV_BBW_AXIS(u, a, dn, dx)                                                                                          \
    {                                                                                                                       \
        const f32x2 qq = {q[a], q[a]};                                                                                      \
        const f32x2 e1 = lo2[u][a] - qq, e2 = qq - hi2[u][a];                                                               \
        dn = f32x2{fmaxf(fmaxf(e1.x, e2.x), 0.f), fmaxf(fmaxf(e1.y, e2.y), 0.f)};                                           \
        dx = f32x2{fmaxf(fabsf(e1.x), fabsf(e2.x)), fmaxf(fabsf(e1.y), fabsf(e2.y))};                                       \
    }
#define RCV_SQ_ADD(d, c) __builtin_elementwise_fma((d), (d), (c))      /* d * d + c */
#define RCV_SUB_SQ(c, d) __builtin_elementwise_fma(-(d), (d), (c))     /* c - d * d */
                if constexpr (VOX) {
                    // last level: the children are single voxels (lo == hi), the box distances collapse to the distance
                    // to the voxel centre: d2 per child, 36 instead of 54 instructions per (parent, point)
#pragma unroll
                    for (int u = 0; u < BBW_NP; u++) {
                        f32x2 t[3];
#pragma unroll
                        for (int a = 0; a < 3; a++) t[a] = lo2[u][a] - f32x2{q[a], q[a]};
                        const f32x2 sq0 = t[0] * t[0];
                        const f32x2 xy[2] = {RCV_SQ_ADD(t[1], (f32x2{sq0.x, sq0.x})), RCV_SQ_ADD(t[1], (f32x2{sq0.y, sq0.y}))};
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const float v = (c & 1) ? xy[c >> 1].y : xy[c >> 1].x;
                            const f32x2 d2 = RCV_SQ_ADD(t[2], (f32x2{v, v}));
                            const f32x2 below = d2 - thi2;  // negative <=> d2 < thi
                            const f32x2 reach = tlo2 - d2;  // negative <=> d2 > tlo
                            RCV_BBW_TAIL(u, c, below, reach);
                        }
                    }
                } else if constexpr (PAIRED_IN) {
                    // (smin < thi) && (smax > tlo) as sign bits, the thresholds folded into the z terms, the yz sums shared by
                    // the two parents of a pair and by both x halves
#pragma unroll
                    for (int pr = 0; pr < BBW_NP / 2; pr++) {
                        const int uA = 2 * pr;
                        f32x2 dny, dxy, dnz, dxz;
                        RCV_BBW_AXIS(uA, 1, dny, dxy);
                        RCV_BBW_AXIS(uA, 2, dnz, dxz);
                        const f32x2 zb = RCV_SQ_ADD(dnz, -thi2), zr = RCV_SUB_SQ(tlo2, dxz);
                        const f32x2 byz[2] = {RCV_SQ_ADD((f32x2{dny.x, dny.x}), zb), RCV_SQ_ADD((f32x2{dny.y, dny.y}), zb)};
                        const f32x2 ryz[2] = {RCV_SUB_SQ(zr, (f32x2{dxy.x, dxy.x})), RCV_SUB_SQ(zr, (f32x2{dxy.y, dxy.y}))};
#pragma unroll
                        for (int u = uA; u <= uA + 1; u++) {
                            f32x2 dnx, dxx;
                            RCV_BBW_AXIS(u, 0, dnx, dxx);
#pragma unroll
                            for (int c = 0; c < 4; c++) {  // c = hx * 2 + hy
                                const float dn = (c >> 1) ? dnx.y : dnx.x, dx = (c >> 1) ? dxx.y : dxx.x;
                                const f32x2 below = RCV_SQ_ADD((f32x2{dn, dn}), byz[c & 1]);
                                const f32x2 reach = RCV_SUB_SQ(ryz[c & 1], (f32x2{dx, dx}));
                                RCV_BBW_TAIL(u, c, below, reach);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < BBW_NP; u++) {
                        f32x2 dn2[3], dx2[3];
#pragma unroll
                        for (int a = 0; a < 3; a++) RCV_BBW_AXIS(u, a, dn2[a], dx2[a]);
                        // the xy partial sums are shared by the two z halves, the thresholds are folded into the z terms
                        const f32x2 mn0 = dn2[0] * dn2[0], mx0 = dx2[0] * dx2[0];
                        const f32x2 mnxy[2] = {RCV_SQ_ADD(dn2[1], (f32x2{mn0.x, mn0.x})), RCV_SQ_ADD(dn2[1], (f32x2{mn0.y, mn0.y}))};
                        const f32x2 mxxy[2] = {RCV_SQ_ADD(dx2[1], (f32x2{mx0.x, mx0.x})), RCV_SQ_ADD(dx2[1], (f32x2{mx0.y, mx0.y}))};
                        const f32x2 zb = RCV_SQ_ADD(dn2[2], -thi2);
                        const f32x2 zr = RCV_SUB_SQ(tlo2, dx2[2]);
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const float mn = (c & 1) ? mnxy[c >> 1].y : mnxy[c >> 1].x;
                            const float mx = (c & 1) ? mxxy[c >> 1].y : mxxy[c >> 1].x;
                            const f32x2 below = f32x2{mn, mn} + zb;  // negative <=> smin < thi
                            const f32x2 reach = zr - f32x2{mx, mx};  // negative <=> smax > tlo
                            RCV_BBW_TAIL(u, c, below, reach);
                        }
                    }
                }
#undef RCV_BBW_AXIS
#undef RCV_SQ_ADD
#undef RCV_SUB_SQ
            }
            if ((ch & (32 / PT - 1)) == 32 / PT - 1) {
#pragma unroll
                for (int u = 0; u < BBW_NP; u++)
#pragma unroll
                    for (int c = 0; c < 8; c++) { ub[u][c] += __popc(hist[u][c]); hist[u][c] = 0u; }
            }
        }
#pragma unroll
        for (int u = 0; u < BBW_NP; u++)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int v = wave_sum_i32(ub[u][c] + __popc(hist[u][c]));
                if (lane == 0) red[wv][u * 8 + c] = v;
            }
        __syncthreads();
        if (tid < 8 * BBW_NP) {  // lanes 0..31 of wave 0: parent u = tid / 8, child c = hx * 4 + hy * 2 + hz
            const int u = tid >> 3, c = tid & 7, hx = (c >> 2) & 1, hy = (c >> 1) & 1, hz = c & 1;
            int cnt = 0;
            for (int w = 0; w < BBW_THREADS / 64; w++) cnt += red[w][tid];
            const int oi = s_org[u][0] + hx * s, oj = s_org[u][1] + hy * s, ok_ = s_org[u][2] + hz * s;
            const int best = __hip_atomic_load(&hdr->bb_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool alive = s_valid[u] && oi <= D - 1 && oj <= D - 1 && ok_ <= D - 1 && cnt > 0 && cnt >= best;
            BBEntry o;
            o.i = (uint16_t)oi; o.j = (uint16_t)oj; o.k = (uint16_t)ok_;
            o.pad_ = 0; o.ub = cnt; o.lb = 0;
            if constexpr (PAIR_OUT) {
                const unsigned m8 = (unsigned)((__ballot(alive) >> (u * 8)) & 0xFFull);  // this parent's survivors, bit = c
                const unsigned pairs = m8 & (m8 >> 4) & 0xFu;                             // (hy, hz) groups with both x halves alive
                const unsigned singles = m8 & ~(pairs | (pairs << 4));
                const int npairs = __popc(pairs), nsingles = __popc(singles);
                unsigned bp = 0, bs = 0;
                if (c == 0 && npairs > 0) bp = atomicAdd(out_pair_count, (unsigned)(2 * npairs));
                if (c == 0 && nsingles > 0) bs = atomicAdd(out_single_count, (unsigned)nsingles);
                bp = (unsigned)__shfl((int)bp, u * 8);
                bs = (unsigned)__shfl((int)bs, u * 8);
                if (alive) {
                    const int grp = c & 3;
                    if ((pairs >> grp) & 1u) {
                        const unsigned slot = bp + 2 * __popc(pairs & ((1u << grp) - 1u)) + hx;
                        if (bp + 2 * npairs <= (unsigned)out_cap) out_list[slot] = o;
                        else atomicExch(&hdr->bb_overflow, 1);
                    } else {
                        const unsigned slot = bs + __popc(singles & ((1u << c) - 1u));
                        if (bs + nsingles <= (unsigned)out_cap) out_list[out_single_off + slot] = o;
                        else atomicExch(&hdr->bb_overflow, 1);
                    }
                }
            } else if (alive) {
                const unsigned slot = atomicAdd(out_single_count, 1u);
                if (slot < (unsigned)out_cap) out_list[out_single_off + slot] = o;
                else atomicExch(&hdr->bb_overflow, 1);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead chunk of the last pass
#undef RCV_BBW_ISSUE
}
#undef RCV_BBW_TAIL

// Where to start climbing.  Every point says "the keypoint lies r_p voxels from me": |c - q_p|^2 = r_p^2, and the difference
// to the mean equation is linear in c -- the 4 x 4 normal equations of (c, |c|^2) over the item's points give the
// multilateration estimate in one pass (coordinates taken relative to the grid centre for conditioning).  With the radii
// the network is trained to produce it lands within two voxels of the vote peak (bench frames: 0.4 ... 2.0), which one
// 5 x 5 x 5 round of exact counts turns into the peak itself; the greedy probe alone needed ten rounds of two voxels
// each.  Only a starting point for exact counts: a poor estimate (outliers, the RGB-only variant) costs nothing but its
// round.  One workgroup per item, sums in a fixed order (the start voxel is the same from run to run).
__device__ __forceinline__ void d_bb_seed(AccHeader *hdr, const double *Q, const double *P, const int32_t *ri, int cap)
{
    constexpr int NS = 15;  // upper triangle of A^T A (10) + A^T b (4), rows a_p = (-2x, -2y, -2z, 1), b_p = r^2 - |q|^2; + the row count
    // 1024 threads (r04; 256 before): the passes are chains of dependent loads, and one wave per SIMD hid none of their latency
    // (465 us per launch with the chip idle around it); sums per wave by shuffles, the 16 wave sums added in a fixed order
    constexpr int NT = BB_SEED_THREADS, NW = NT / 64;
    __shared__ double part[NS][NW];
    __shared__ double tot[NS];
    __shared__ double s_c[3];   // the current estimate (relative to the grid centre)
    __shared__ int s_have;      // an estimate exists
    __shared__ int s_hist[512]; // |residual| of the points against the current estimate, 1-voxel bins
    __shared__ double s_tau;
    __shared__ double M[4][5], sol[4];  // thread 0's elimination: pivoting indexes rows at run time -- in LDS, not in scratch memory (176 bytes per thread until r05)
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled) return;
    if (threadIdx.x == 0) { hdr->seed = 0; s_have = 0; }
    if (H.variant != 0) return;
    const int n = H.n;
    const double t0 = 0.5 * (double)H.dim;
    const double *Phi = P + 4 * (size_t)cap;
    // Five passes (r03): the plain least-squares fit, then four trimmed re-fits over the better half of the points -- those whose
    // shell passes within tau of the current estimate, tau = the median |residual| (from a 1-voxel histogram), at least 3
    // voxels.  Gross outliers (false-positive segmentation pixels on the background, wrong radii) enter the first fit with
    // the square of their radius and drag it tens of voxels away -- the hill-climb then starts on a side lobe, the search
    // prunes with a lower bound of a quarter of the peak and visits 20 x the blocks (tools/vote_sweep.py: 141 instead of 18 ms
    // per 216 items at 10 % false positives).  A re-fit that fails numerically keeps the previous estimate.
    for (int pass = 0; pass < 5; pass++) {
        __syncthreads();  // s_c / s_have of the previous pass
        const bool trim = pass > 0;
        if (trim && !s_have) break;
        const double cx = s_c[0], cy = s_c[1], cz = s_c[2];
        if (trim) {
            for (int t = threadIdx.x; t < 512; t += NT) s_hist[t] = 0;
            __syncthreads();
            for (int p = threadIdx.x; p < n; p += NT) {
                if (!(Phi[p] > 0.0)) continue;
                const double dx = cx - (Q[p] - t0), dy = cy - (Q[(size_t)cap + p] - t0), dz = cz - (Q[2 * (size_t)cap + p] - t0);
                const double res = fabs(sqrt((dx * dx + dy * dy) + dz * dz) - (double)ri[p]);
                atomicAdd(&s_hist[res < 511.0 ? (int)res : 511], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int total = 0;
                for (int t = 0; t < 512; t++) total += s_hist[t];
                int run = 0, t = 0;
                for (; t < 511; t++) { run += s_hist[t]; if (2 * run >= total) break; }
                s_tau = fmax(3.0, (double)(t + 1));
            }
            __syncthreads();
        }
        const double tau = trim ? s_tau : 0.0;
        double sm[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) sm[k] = 0.0;
#pragma unroll 2
        for (int p = threadIdx.x; p < n; p += NT) {
            if (!(Phi[p] > 0.0)) continue;  // never votes
            const double x = Q[p] - t0, y = Q[(size_t)cap + p] - t0, z = Q[2 * (size_t)cap + p] - t0, r = (double)ri[p];
            if (trim) {
                const double dx = cx - x, dy = cy - y, dz = cz - z;
                if (!(fabs(sqrt((dx * dx + dy * dy) + dz * dz) - r) <= tau)) continue;
            }
            const double a[4] = {-2.0 * x, -2.0 * y, -2.0 * z, 1.0};
            const double b = r * r - ((x * x + y * y) + z * z);
            int k = 0;
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = u; v < 4; v++) sm[k++] += a[u] * a[v];
#pragma unroll
            for (int u = 0; u < 4; u++) sm[10 + u] += a[u] * b;
            sm[14] += 1.0;
        }
#pragma unroll
        for (int k = 0; k < NS; k++) {
            double v = sm[k];
            for (int m = 32; m; m >>= 1) v += shfl_xor_f64(v, m);
            if (lane_id() == 0) part[k][wave_id()] = v;
        }
        __syncthreads();
        if (threadIdx.x < NS) {
            double acc = 0.0;
            for (int t = 0; t < NW; t++) acc += part[threadIdx.x][t];
            tot[threadIdx.x] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            bool ok = !trim || (tot[14] >= 16.0 && tot[14] * 16.0 >= (double)n);
            {
                int k = 0;
                for (int u = 0; u < 4; u++)
                    for (int v = u; v < 4; v++) { M[u][v] = tot[k]; M[v][u] = tot[k]; k++; }
                for (int u = 0; u < 4; u++) M[u][4] = tot[10 + u];
            }
            for (int c = 0; c < 4 && ok; c++) {  // Gaussian elimination, partial pivoting
                int piv = c;
                for (int r = c + 1; r < 4; r++)
                    if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
                if (!(fabs(M[piv][c]) > 0.0)) { ok = false; break; }
                for (int v = 0; v < 5; v++) { const double tmp = M[c][v]; M[c][v] = M[piv][v]; M[piv][v] = tmp; }
                for (int r = c + 1; r < 4; r++) {
                    const double f = M[r][c] / M[c][c];
                    for (int v = c; v < 5; v++) M[r][v] -= f * M[c][v];
                }
            }
            if (ok) {
                for (int c = 3; c >= 0; c--) {
                    double v = M[c][4];
                    for (int u = c + 1; u < 4; u++) v -= M[c][u] * sol[u];
                    sol[c] = v / M[c][c];
                }
                for (int a = 0; a < 3; a++) ok = ok && sol[a] > -1.0e6 && sol[a] < 1.0e6;  // NaN / far outside: not an estimate
                if (ok) { s_c[0] = sol[0]; s_c[1] = sol[1]; s_c[2] = sol[2]; s_have = 1; }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x != 0 || !s_have) return;
    long long vox[3];
    for (int a = 0; a < 3; a++) {
        vox[a] = llrint(s_c[a] + t0);
        vox[a] = vox[a] < 0 ? 0 : (vox[a] >= H.dim ? H.dim - 1 : vox[a]);
    }
    hdr->seed = 1ull + (unsigned long long)((vox[0] * H.dim + vox[1]) * H.dim + vox[2]);
}

// Hill-climb on exact voxel counts around the best voxel found so far (5 x 5 x 5 neighbourhood, one workgroup
// per voxel).  The greedy probe lands near the peak; a few rounds of this bring the lower bound to the peak
// itself (or next to it), which is what lets the wide levels prune.  Purely an accelerator: any exact count
// is a valid lower bound.
constexpr int BB_CLIMB_R = 2;
constexpr int BB_CLIMB_W = 2 * BB_CLIMB_R + 1;
constexpr int BB_CLIMB_ROUNDS = RCV_BB_CLIMB_ROUNDS;

// one workgroup per i-offset of the (2R+1)^3 neighbourhood; every thread scores the (2R+1)^2 voxels of that slice for
// its points (dx^2 shared), then the slice's counts are reduced and folded into the running best
__device__ __forceinline__ void d_bb_climb(AccHeader *hdr, const double *Q, const double *P, int cap, int round)
{
    constexpr int W = BB_CLIMB_W, NV = W * W;
    __shared__ int red[4][NV];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled || H.bb_overflow) return;  // (handed over to the rasteriser: nothing left to bound)
    // the round's centre: round 0 looks around the least-squares estimate; round r > 0 around the best voxel as of the end of
    // round r - 1 (climb_key[(r - 1) & 1], which no workgroup of THIS round writes: k_bb_gate folds the probe's finds into slot
    // 0 before round 1)
    const bool from_seed = round == 0;
    if (from_seed && H.seed == 0) return;
    // (the two slots are read through the pointer: indexing the header COPY with a run-time subscript keeps the whole 400-byte
    // struct in scratch memory -- every thread of every workgroup loaded and stored it, 19 us per launch even for the rounds that
    // return at once; r04)
    const int prev = (round - 1) & 1;
    const unsigned long long frozen = from_seed ? 0ull : hdr->climb_key[prev];
    if (!from_seed && frozen == 0) return;
    const int D = H.dim, n = H.n;
    const unsigned long long flat = from_seed ? H.seed - 1 : (frozen & 0xFFFFFFFFull);
    // this round's slot starts from the previous round's best (it still holds round r - 2's, which is not larger)
    if (!from_seed && threadIdx.x == 0) atomicMax(&hdr->climb_key[round & 1], frozen);
    // converged: the best voxel is the one the previous round was centred on -- its whole neighbourhood has been counted
    // and nothing in it was better.  (Every workgroup of the round computes the same centre; the first one records it.)
    if (round > 0 && hdr->climb_center[prev] == flat + 1) {
        if (blockIdx.x == 0 && threadIdx.x == 0) hdr->climb_center[round & 1] = flat + 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr->climb_center[round & 1] = flat + 1;
    const int ci = (int)(flat / ((unsigned long long)D * D)) + (int)blockIdx.x - BB_CLIMB_R;
    const int cj0 = (int)((flat / D) % D) - BB_CLIMB_R, ck0 = (int)(flat % D) - BB_CLIMB_R;
    if (ci < 0 || ci >= D) return;
    const double fi = (double)ci;
    const double *Qx = Q, *Qy = Q + (size_t)cap, *Qz = Q + 2 * (size_t)cap;
    const double *Plo = P + 3 * (size_t)cap, *Phi = P + 4 * (size_t)cap;
    int cnt[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) cnt[v] = 0;
    for (int p = threadIdx.x; p < n; p += 256) {
        const double thi = Phi[p];
        if (!(thi > 0.0)) continue;
        const double tlo = Plo[p], qy = Qy[p], qz = Qz[p];
        const double dx = fi - Qx[p], dx2 = dx * dx;
        double y2[W], z2[W];
#pragma unroll
        for (int o = 0; o < W; o++) {
            const double dy = (double)(cj0 + o) - qy, dz = (double)(ck0 + o) - qz;  // (j - y), (k - z)
            y2[o] = dy * dy; z2[o] = dz * dz;
        }
#pragma unroll
        for (int a = 0; a < W; a++) {
            const double xy = dx2 + y2[a];  // reference operand order (:337)
#pragma unroll
            for (int b = 0; b < W; b++) {
                const double d2 = xy + z2[b];
                const int below = (int)(__double_as_longlong(d2 - thi) >> 32);
                const int reach = ~(int)(__double_as_longlong(d2 - tlo) >> 32);
                cnt[a * W + b] += (int)((unsigned)(below & reach) >> 31);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int c = wave_sum_i32(cnt[v]);
        if (lane_id() == 0) red[wave_id()][v] = c;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int v = threadIdx.x, cj = cj0 + v / W, ck = ck0 + v % W;
        if (cj >= 0 && ck >= 0 && cj < D && ck < D) {
            const int c = red[0][v] + red[1][v] + red[2][v] + red[3][v];
            const unsigned long long key = ((unsigned long long)c << 32) | (((unsigned long long)ci * D + cj) * D + ck);
            atomicMax(&hdr->bb_best, c);
            atomicMax(&hdr->bb_best_key, key);
            atomicMax(&hdr->climb_key[round & 1], key);
        }
    }
}

// Line search along the sheet normal (r03).  The vote landscape around the keypoint is a sheet about one voxel thick, normal
// to the direction from the points' centroid to the keypoint: an estimate that lies three voxels off the sheet
// ALONG that normal -- which is where outliers push the least-squares fit -- has nothing but side lobes in its 5 x 5 x 5
// neighbourhood, and the hill-climb ends on one of them.  When the round around the seed found no strong peak, the exact counts
// of 125 voxels on the line seed + t * u, t = -62 ... 62, u = unit(seed - centroid), find the sheet crossing; the climb
// rounds then walk inside the sheet.  Five workgroups of 25 positions; any exact count is a valid lower bound.
__device__ __forceinline__ void d_bb_line(AccHeader *hdr, const double *Q, const double *P, int cap, int mode)
{
    constexpr int NV = 25;
    __shared__ int red[4][NV];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled || H.bb_overflow || H.variant != 0) return;
    // mode 0: through the least-squares seed, when its round found no strong peak; mode 1: through the restart voxel (k_bb_restart)
    if (mode == 0 ? (H.seed == 0 || H.probe_skip) : H.restart == 0) return;
    const int D = H.dim, n = H.n;
    const int slot = mode == 0 ? 0 : 1;
    const unsigned long long flat = mode == 0 ? H.seed - 1 : H.restart - 1;
    const double s0 = (double)(flat / ((unsigned long long)D * D)), s1 = (double)((flat / D) % D), s2 = (double)(flat % D);
    // centroid of the recentred points: Q = voxel coordinates - mean - shift
    const double c0 = -H.shift;
    double u0 = s0 - c0, u1 = s1 - c0, u2 = s2 - c0;
    const double nrm = sqrt((u0 * u0 + u1 * u1) + u2 * u2);
    if (!(nrm >= 1.0)) return;
    u0 /= nrm; u1 /= nrm; u2 /= nrm;
    double vi[NV], vj[NV], vk[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const double t = (double)((int)blockIdx.x * NV + v - 62);
        vi[v] = rint(s0 + t * u0); vj[v] = rint(s1 + t * u1); vk[v] = rint(s2 + t * u2);
    }
    const double *Qx = Q, *Qy = Q + (size_t)cap, *Qz = Q + 2 * (size_t)cap;
    const double *Plo = P + 3 * (size_t)cap, *Phi = P + 4 * (size_t)cap;
    int cnt[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) cnt[v] = 0;
    for (int p = threadIdx.x; p < n; p += 256) {
        const double thi = Phi[p];
        if (!(thi > 0.0)) continue;
        const double tlo = Plo[p], qx = Qx[p], qy = Qy[p], qz = Qz[p];
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const double dx = vi[v] - qx, dy = vj[v] - qy, dz = vk[v] - qz;
            const double d2 = (dx * dx + dy * dy) + dz * dz;  // reference operand order (:337)
            const int below = (int)(__double_as_longlong(d2 - thi) >> 32);
            const int reach = ~(int)(__double_as_longlong(d2 - tlo) >> 32);
            cnt[v] += (int)((unsigned)(below & reach) >> 31);
        }
    }
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int c = wave_sum_i32(cnt[v]);
        if (lane_id() == 0) red[wave_id()][v] = c;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        // (position v = threadIdx.x recomputed: a run-time subscript into vi / vj / vk / inside would put the four arrays into scratch
        // memory -- 656 bytes per thread until r05)
        const int v = threadIdx.x;
        const double t = (double)((int)blockIdx.x * NV + v - 62);
        const double wi = rint(s0 + t * u0), wj = rint(s1 + t * u1), wk = rint(s2 + t * u2);
        if (wi >= 0.0 && wj >= 0.0 && wk >= 0.0 && wi < (double)D && wj < (double)D && wk < (double)D) {
            const int c = red[0][v] + red[1][v] + red[2][v] + red[3][v];
            const unsigned long long key = ((unsigned long long)c << 32) | (((unsigned long long)wi * D + (unsigned long long)wj) * D + (unsigned long long)wk);
            atomicMax(&hdr->bb_best, c);
            atomicMax(&hdr->bb_best_key, key);
            atomicMax(&hdr->climb_key[slot], key);
        }
    }
}

// A second start for the lower bound, from the search's own upper bounds (r03).  The seed and the greedy probe can both end on a
// side lobe (outliers drag the least-squares fit along the viewing direction, tools/vote_sweep.py; the probe ranks 32-voxel blocks
// whose bounds are all ~N by their centre counts, i.e. by noise).  Once a level has produced 8- or 4-voxel blocks, their bounds
// do discriminate: the block most shells cross is at the keypoint.  This kernel picks the candidate block with the largest upper
// bound (first in list order among equals: the list order is deterministic up to the atomics that built it, and any start is
// valid); the line search through its centre voxel and two climb rounds follow.  If the lower bound already sits on the peak,
// they find it again.
__device__ __forceinline__ void d_bb_restart(AccHeader *hdr, const BBEntry *list, uint32_t n_pairs, uint32_t n_singles, int edge)
{
    __shared__ unsigned long long s_key[16];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled || H.bb_overflow || H.variant != 0) return;
    const uint32_t na = min(n_pairs, (uint32_t)BB_HALF), nb = min(n_singles, (uint32_t)BB_HALF), n = na + nb;
    // Cost-based hand-over (r04).  The search certifies the arg-max by scoring every surviving block against all N points; on a
    // flat vote landscape (radial noise of several voxels) the survivors of the fine levels outgrow the rasteriser's work, which
    // is one interval evaluation per (point, column inside the shell's disc) ~ N * pi * r^2 whatever the noise.  Measured on the
    // bench cloud (N = 15.5k, radii 50 ... 94 voxels; tools/vote_sweep.py): 1.1 ps per (block, point) against 4.8 ps per (point,
    // column) (8 before the rasteriser's r04 rework), and the 2-voxel level sees about twice the 4-voxel level's survivors -- so
    // once MORE than 2.5 r_max^2 four-voxel candidates survive (22k on that cloud), finishing the search costs more than voting
    // the whole grid: the search gives up
    // here (the fallback every search exit takes: k_vote decides; identical result by construction).  Both costs are
    // proportional to N; the grid edge does not enter (false positives on a background plane widen the grid, not the shells).
    if (edge == 4 && (double)n > BB_HANDOVER_R2 * H.rmax * H.rmax) {
        if (threadIdx.x == 0) {
            hdr->bb_overflow = 1;
            hdr->bb_handover = 1;
        }
        return;
    }
    unsigned long long best = 0;
    __shared__ int s_box[6];
    if (threadIdx.x < 6) s_box[threadIdx.x] = (threadIdx.x & 1) ? -1 : 0x7fffffff;
    __syncthreads();
    int bx[6] = {0x7fffffff, -1, 0x7fffffff, -1, 0x7fffffff, -1};
    for (uint32_t t = threadIdx.x; t < n; t += 1024) {
        const uint32_t idx = t < na ? t : (uint32_t)BB_HALF + (t - na);
        const BBEntry e = list[idx];
        if (e.ub >= H.bb_best) {  // (stale entries below the current bound would be skipped by the level kernel)
            bx[0] = min(bx[0], (int)e.i); bx[1] = max(bx[1], (int)e.i + edge - 1);
            bx[2] = min(bx[2], (int)e.j); bx[3] = max(bx[3], (int)e.j + edge - 1);
            bx[4] = min(bx[4], (int)e.k); bx[5] = max(bx[5], (int)e.k + edge - 1);
        }
        // ub (31 bits) | inverted position (so that the first entry wins among equals) | + 1: 0 means none
        const unsigned long long key = ((unsigned long long)(uint32_t)min(e.ub, 0x7FFFFFFF) << 32) | (unsigned long long)(0xFFFFFFu - t);
        if (key + 1 > best) best = key + 1;
    }
    best = wave_max_u64(best);
    if (lane_id() == 0) s_key[wave_id()] = best;
    for (int a = 0; a < 6; a++) {
        if (a & 1) atomicMax(&s_box[a], bx[a]);
        else atomicMin(&s_box[a], bx[a]);
    }
    __syncthreads();
    if (threadIdx.x < 6) hdr->box[edge == 8 ? 0 : 1][threadIdx.x] = s_box[threadIdx.x];
    if (threadIdx.x != 0) return;
    for (int w = 1; w < 16; w++) best = s_key[w] > best ? s_key[w] : best;
    // both slots start empty: the restart's first round (even) folds its neighbourhood into slot 0, and the round after it must
    // re-centre on THIS restart's best, not on the older side lobe still standing there (bb_best / bb_best_key keep the bound)
    hdr->climb_key[0] = 0;
    hdr->climb_key[1] = 0;
    hdr->climb_center[1] = 0;
    hdr->restart = 0;
    // a lower bound that an eighth of the points vote for sits on the peak's sheet already (the gate of the probe): nothing to restart
    if (!best || (long long)H.bb_best * 8 >= (long long)H.n) return;
    const uint32_t t = 0xFFFFFFu - (uint32_t)((best - 1) & 0xFFFFFFull);
    const BBEntry e = list[t < na ? t : (uint32_t)BB_HALF + (t - na)];
    const int D = H.dim;
    const unsigned long long ci = min((int)e.i + edge / 2, D - 1), cj = min((int)e.j + edge / 2, D - 1), ck = min((int)e.k + edge / 2, D - 1);
    hdr->restart = 1ull + (ci * D + cj) * D + ck;
}

// Exact counts of the candidate voxels.  The last wide level (2^3 parents, single-voxel children) leaves the voxels
// whose float32 upper bound reaches the best exact count: a few dozen around the peak.  Here every candidate gets
// the reference predicate itself, float64 throughout: d2 = ((i-x)^2 + (j-y)^2) + (k-z)^2 in the reference's order,
// vote <=> T_lo <= d2 < T_hi, decided on the signs of the two float64 differences (a floating-point subtraction has
// the sign of the exact difference, and x - x is +0).  A workgroup scores 8 candidates per pass over the point list
// (streamed through LDS) and writes the counts back into their list entries (ub = lb = votes).
constexpr int BBL_THREADS = 256;
constexpr int BBL_CHUNK = 1024;
constexpr int BBL_NV = 8;
constexpr int BBL_PARTS = 8;

__device__ __forceinline__ void d_bb_exact(AccHeader *hdr, const double *Q, const double *P, int cap, BBEntry *list,
                                           const uint32_t *count, int list_cap)
{
    __shared__ double sx[BBL_CHUNK], sy[BBL_CHUNK], sz[BBL_CHUNK], slo[BBL_CHUNK], shi[BBL_CHUNK];
    __shared__ int red[BBL_THREADS / 64][BBL_NV];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled || H.bb_overflow) return;
    const int n = H.n;
    const unsigned nin = min(*count, (uint32_t)list_cap);
    const double *Qx = Q, *Qy = Q + (size_t)cap, *Qz = Q + 2 * (size_t)cap;
    const double *Plo = P + 3 * (size_t)cap, *Phi = P + 4 * (size_t)cap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int PT = BBL_CHUNK / BBL_THREADS;
    // workgroup = (group of BBL_NV candidates, every BBL_PARTS-th chunk of the point list): the partial counts are added to the
    // entries' lb fields (zero on entry) and k_bb_final takes the maximum.  (r02: one workgroup per candidate group walked the
    // whole list -- with 1 ... 13 candidates per item that was one or two busy workgroups per item, 0.47 ms per 216 items.)
    const int part = (int)(blockIdx.x % BBL_PARTS);
    for (unsigned pp = blockIdx.x / BBL_PARTS; pp * BBL_NV < nin; pp += gridDim.x / BBL_PARTS) {
        double ci[BBL_NV], cj[BBL_NV], ck[BBL_NV];
#pragma unroll
        for (int v = 0; v < BBL_NV; v++) {
            const BBEntry e = list[min(pp * BBL_NV + v, nin - 1)];
            ci[v] = (double)e.i; cj[v] = (double)e.j; ck[v] = (double)e.k;
        }
        int cnt[BBL_NV];
        unsigned hist[BBL_NV];  // vote bits of the last <= 32 points, as in k_bb_wide
#pragma unroll
        for (int v = 0; v < BBL_NV; v++) { cnt[v] = 0; hist[v] = 0u; }
        int chunk_no = 0;
        for (int base = part * BBL_CHUNK; base < n; base += BBL_PARTS * BBL_CHUNK, chunk_no++) {
            __syncthreads();
            {
                double rx[PT], ry[PT], rz[PT], rl[PT], rh[PT];
#pragma unroll
                for (int t = 0; t < PT; t++) {
                    const int p = min(base + t * BBL_THREADS + tid, n - 1);
                    rx[t] = Qx[p]; ry[t] = Qy[p]; rz[t] = Qz[p]; rl[t] = Plo[p]; rh[t] = Phi[p];
                }
#pragma unroll
                for (int t = 0; t < PT; t++) {
                    const int li = t * BBL_THREADS + tid;
                    const bool live = base + li < n && rh[t] > 0.0;
                    sx[li] = rx[t]; sy[li] = ry[t]; sz[li] = rz[t];
                    slo[li] = rl[t];
                    shi[li] = live ? rh[t] : -1.0;  // d2 < -1 never holds
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int t = 0; t < PT; t++) {
                const int li = t * BBL_THREADS + tid;
                const double qx = sx[li], qy = sy[li], qz = sz[li];
                const double tlo = slo[li], thi = shi[li];
#pragma unroll
                for (int v = 0; v < BBL_NV; v++) {
                    const double dx = ci[v] - qx, dy = cj[v] - qy, dz = ck[v] - qz;  // (i - x), reference operand order (:337)
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    const int below = (int)(__double_as_longlong(d2 - thi) >> 32);   // negative <=> d2 < T_hi
                    const int reach = ~(int)(__double_as_longlong(d2 - tlo) >> 32);  // sign set <=> d2 >= T_lo
                    hist[v] = __builtin_amdgcn_alignbit(hist[v], (unsigned)(below & reach), 31);
                }
            }
            if ((chunk_no & (32 / PT - 1)) == 32 / PT - 1) {
#pragma unroll
                for (int v = 0; v < BBL_NV; v++) { cnt[v] += __popc(hist[v]); hist[v] = 0u; }
            }
        }
#pragma unroll
        for (int v = 0; v < BBL_NV; v++) {
            const int u = wave_sum_i32(cnt[v] + __popc(hist[v]));
            if (lane == 0) red[wv][v] = u;
        }
        __syncthreads();
        if (tid < BBL_NV && pp * BBL_NV + tid < nin) {
            int u = 0;
            for (int w = 0; w < BBL_THREADS / 64; w++) u += red[w][tid];
            if (u) atomicAdd(&list[pp * BBL_NV + tid].lb, u);
        }
        __syncthreads();  // red[] is rewritten by the next pass
    }
}

// Ranking counts for the probe: the (float32, approximate) number of votes at the centre voxel of every block of a list.
// Many blocks of a coarse level are crossed by every shell (equal upper bounds); the centre counts break those ties so
// that the probe descends into the block that holds the peak.  Never used as a bound.  A workgroup takes 8 blocks per
// pass over the float32 point list (read straight from L2: 20 bytes per point).
__device__ __forceinline__ void d_bb_centres(const AccHeader *hdr, const float *F, size_t capF, BBEntry *list, const uint32_t *count,
                                             int list_cap, int level)
{
    __shared__ int red[4][8];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled) return;
    const int D = H.dim, s = (BB_ROOT >> level) >> 1;  // edge of the blocks in the list (children of `level`)
    const unsigned nin = min(*count, (uint32_t)list_cap);
    const int npad = (H.n + 1023) & ~1023;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (unsigned pp = blockIdx.x; pp * 8 < nin; pp += gridDim.x) {
        float cx[8], cy[8], cz[8];
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const BBEntry e = list[min(pp * 8 + v, nin - 1)];
            cx[v] = (float)min((int)e.i + s / 2, D - 1); cy[v] = (float)min((int)e.j + s / 2, D - 1); cz[v] = (float)min((int)e.k + s / 2, D - 1);
        }
        int cnt[8];
#pragma unroll
        for (int v = 0; v < 8; v++) cnt[v] = 0;
        for (int p = tid; p < npad; p += 256) {
            const float x = F[p], y = F[capF + p], z = F[2 * capF + p], tlo = F[3 * capF + p], thi = F[4 * capF + p];
#pragma unroll
            for (int v = 0; v < 8; v++) {
                const float dx = cx[v] - x, dy = cy[v] - y, dz = cz[v] - z;
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                cnt[v] += (d2 >= tlo) & (d2 < thi);
            }
        }
        __syncthreads();  // red[] of the previous pass has been read
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const int c = wave_sum_i32(cnt[v]);
            if (lane == 0) red[wv][v] = c;
        }
        __syncthreads();
        if (tid < 8 && pp * 8 + tid < nin) list[pp * 8 + tid].lb = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}

// the BB_BEAM entries with the largest upper bound (ties: larger lower bound), for the greedy probe
// (the list may be two regions: entries [0, count_a) and [off_b, off_b + count_b); count_b may be null)
__device__ __forceinline__ void d_bb_topk(const AccHeader *hdr, const BBEntry *list, const uint32_t *count_a, const uint32_t *count_b,
                                                  int off_b, int cap, BBEntry *sel, uint32_t *sel_count)
{
    __shared__ unsigned long long s_key[16];
    __shared__ unsigned long long s_taken[BB_BEAM];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled) return;
    const uint32_t na = min(*count_a, (uint32_t)cap), nb = count_b ? min(*count_b, (uint32_t)cap) : 0u, n = na + nb;
    const int tid = threadIdx.x;
    int picked = 0;
    for (int round = 0; round < BB_BEAM && round < (int)n; round++) {
        unsigned long long best = 0;
        for (uint32_t t = tid; t < n; t += 1024) {
            const uint32_t idx = t < na ? t : (uint32_t)off_b + (t - na);
            const BBEntry e = list[idx];
            // key: ub (31 bits) | lb (clamped to 12 bits) | index (20 bits) + 1 so that 0 means "none"
            unsigned long long key = ((unsigned long long)(uint32_t)min(e.ub, 0x7FFFFFFF) << 32) |
                                     ((unsigned long long)min(e.lb, 4095) << 20) | (unsigned long long)(idx & 0xFFFFF);
            bool taken = false;
            for (int q = 0; q < round; q++) taken |= s_taken[q] == key;
            if (!taken && key + 1 > best) best = key + 1;
        }
        best = wave_max_u64(best);
        if (lane_id() == 0) s_key[wave_id()] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = 0;
            for (int w = 0; w < 16; w++) b = s_key[w] > b ? s_key[w] : b;
            s_taken[round] = b ? b - 1 : ~0ull;
            if (b) sel[round] = list[(uint32_t)((b - 1) & 0xFFFFF)];
        }
        __syncthreads();
        if (s_taken[round] != ~0ull) picked = round + 1;
    }
    if (tid == 0) *sel_count = (uint32_t)picked;
}

// row 0 of the reference's `center` from the voxel index of the first maximum
__device__ __forceinline__ double center_component(const AccHeader &H, int idx, int a)
{
    if (H.variant) {
        // Accumulator_3D_no_depth (AccumulatorSpace.py:468-477): `center` is an int64 array; center + zero_boundary stays
        // integer, `center[0,a] += mean` stores a float into an int64 element (C cast, toward zero), then center * 5
        long long c = idx;
        if (H.zb < 0) c += H.zb;
        const double shifted = (double)c + H.nd_mean[a];
        return (double)((long long)shifted * 5ll);
    }
    double c = (double)idx;
    if (H.zb < 0) c = c + (double)H.zb;                 // center + zero_boundary, :409-410
    return ((c + H.mean[a]) + 0.5) * H.unit;            // (center + mean + 0.5) * acc_unit, :413-415
}

__device__ __forceinline__ void d_bb_final(AccHeader *hdr, const BBEntry *voxels, rcv_acc_result *out)
{
    __shared__ uint32_t s_first[16];
    __shared__ int s_ties[16];
    const AccHeader H = *hdr;
    if (H.status != RCV_OK || !H.bb_enabled || H.bb_overflow || H.bb_best <= 0) return;  // the rasteriser decides
    const uint32_t nv = H.bs_count[BB_LEVELS];
    if (nv == 0 || nv > (uint32_t)BB_HALF) return;
    // the peak: the largest exact count among the surviving voxels (k_bb_exact left it in lb).  The voxel that gave the search
    // its lower bound is one of them, so anything below that bound means a broken list: the rasteriser decides then.
    __shared__ int s_peak[16];
    int peak = 0;
    for (uint32_t t = threadIdx.x; t < nv; t += 1024) peak = max(peak, voxels[t].lb);
    for (int m = 32; m; m >>= 1) peak = max(peak, __shfl_xor(peak, m));
    if (lane_id() == 0) s_peak[wave_id()] = peak;
    __syncthreads();
    for (int w = 0; w < 16; w++) peak = max(peak, s_peak[w]);
    if (peak < H.bb_best) return;
    const uint32_t D = (uint32_t)H.dim;
    uint32_t first = 0xFFFFFFFFu;
    int ties = 0;
    for (uint32_t t = threadIdx.x; t < nv; t += 1024) {
        const BBEntry e = voxels[t];
        if (e.lb == peak) {
            ties++;
            const uint32_t flat = ((uint32_t)e.i * D + e.j) * D + e.k;
            first = flat < first ? flat : first;
        }
    }
    for (int m = 32; m; m >>= 1) { const uint32_t o = __shfl_xor(first, m); first = o < first ? o : first; }
    ties = wave_sum_i32(ties);
    if (lane_id() == 0) { s_first[wave_id()] = first; s_ties[wave_id()] = ties; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 1; w < 16; w++) { first = s_first[w] < first ? s_first[w] : first; ties += s_ties[w]; }
    if (ties <= 0) return;  // cannot happen (the peak voxel is in the list); leave it to the rasteriser
    rcv_acc_result R;
    memset(&R, 0, sizeof(R));
    R.dim = H.dim; R.zero_boundary = H.zb; R.n_points = H.n; R.n_radii = H.m; R.status = RCV_OK;
    R.radius_max = H.rmax;
    const int idx[3] = {(int)(first / (D * D)), (int)((first / D) % D), (int)(first % D)};
    for (int a = 0; a < 3; a++) {
        R.mean_vox[a] = H.variant ? H.nd_mean[a] : H.mean[a];
        R.first_idx[a] = idx[a];
        R.center_mm[a] = center_component(H, idx[a], a);
    }
    R.peak_votes = peak;
    R.n_ties = ties;
    *out = R;
    hdr->bb_done = 1;
}

__device__ __forceinline__ void d_final(const AccHeader *hdr, const uint64_t *tile_best, const uint32_t *tile_ties,
                                               int grid_given, unsigned long long grid_cells, rcv_acc_result *out)
{
    __shared__ uint64_t sb[4];
    __shared__ unsigned long long st[4];
    const AccHeader H = *hdr;
    if (H.bb_done) return;  // k_bb_final already wrote the result
    rcv_acc_result R;
    memset(&R, 0, sizeof(R));
    R.dim = H.dim; R.zero_boundary = H.zb; R.n_points = H.n; R.n_radii = H.m; R.status = H.status;
    R.radius_max = H.rmax;
    for (int a = 0; a < 3; a++) R.mean_vox[a] = H.variant ? H.nd_mean[a] : H.mean[a];
    if (H.status != RCV_OK) {
        if (threadIdx.x == 0) *out = R;
        return;
    }
    uint64_t best = 0;
    for (int t = threadIdx.x; t < H.ntiles; t += 256) best = tile_best[t] > best ? tile_best[t] : best;
    best = wave_max_u64(best);
    if (lane_id() == 0) sb[wave_id()] = best;
    __syncthreads();
    best = sb[0];
    for (int q = 1; q < 4; q++) best = sb[q] > best ? sb[q] : best;
    const uint32_t vmax = (uint32_t)(best >> 32);
    unsigned long long ties = 0;
    for (int t = threadIdx.x; t < H.ntiles; t += 256)
        if ((uint32_t)(tile_best[t] >> 32) == vmax) ties += tile_ties[t];
    for (int m = 32; m; m >>= 1) ties += shfl_xor_u64(ties, m);
    if (lane_id() == 0) st[wave_id()] = ties;
    __syncthreads();
    if (threadIdx.x != 0) return;
    ties = st[0] + st[1] + st[2] + st[3];
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull);
    const uint32_t D = (uint32_t)H.dim;
    const int idx[3] = {(int)(flat / (D * D)), (int)((flat / D) % D), (int)(flat % D)};
    for (int a = 0; a < 3; a++) {
        R.first_idx[a] = idx[a];
        R.center_mm[a] = center_component(H, idx[a], a);
    }
    R.peak_votes = (int32_t)vmax;
    R.n_ties = ties > 2147483647ull ? 2147483647 : (int32_t)ties;
    if (grid_given && (unsigned long long)D * D * D > grid_cells) R.status = RCV_E_GRID_BUFFER;
    *out = R;
}

__global__ void k_debug_points(const AccHeader *hdr, const double *P, const int32_t *ri, int cap, int out_cap,
                               double *points_vox, int32_t *radius_int)
{
    const AccHeader H = *hdr;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H.n || i >= out_cap) return;
    for (int a = 0; a < 3; a++) points_vox[3 * (size_t)i + a] = (P[(size_t)a * cap + i] - H.mean[a]) - H.shift;
    if (radius_int) radius_int[i] = ri[i];
}

// argwhere(grid == peak), ascending flat order; one workgroup walks the grid in order (rare path)
__global__ __launch_bounds__(1024) void k_collect_ties(const uint32_t *grid, unsigned long long total, uint32_t peak,
                                                       int64_t *out, int cap, int32_t *count_out)
{
    __shared__ int wc[16];
    __shared__ int running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (unsigned long long base = 0; base < total; base += 1024) {
        unsigned long long c = base + threadIdx.x;
        bool hit = c < total && grid[c] == peak;
        uint64_t b = __ballot(hit);
        if (lane_id() == 0) wc[wave_id()] = __popcll(b);
        __syncthreads();
        int before = running;
        for (int q = 0; q < wave_id(); q++) before += wc[q];
        if (hit) {
            int dst = before + __popcll(b & ((1ull << lane_id()) - 1ull));
            if (dst < cap) out[dst] = (int64_t)c;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int s = 0; for (int q = 0; q < 16; q++) s += wc[q]; running += s; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = running;
}

// ---------------------------------------------------------------------------------------------
// Kernels.  Every launch covers a batch of independent (frame, keypoint) items: blockIdx.y = frame,
// blockIdx.z = keypoint; each buffer is addressed as base + frame * stride_frame + keypoint * stride_kp
// (stride 0 = shared, e.g. the depth map of a frame is shared by its keypoints).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct SP {
    T *p;
    long long sb, sg;
    __device__ __forceinline__ T *get() const
    {
        return p ? (T *)((char *)const_cast<typename std::remove_const<T>::type *>(p) + (long long)blockIdx.y * sb +
                         (long long)blockIdx.z * sg)
                 : nullptr;
    }
};

template <typename RT>
__global__ __launch_bounds__(256) void k_seam_count(SP<const float> seg, SP<const RT> rad, SP<const uint16_t> depth, int HW,
                                                    float thr, SP<const double> rmax, double rmax_scalar, int mode,
                                                    SP<unsigned char> ws, Layout L)
{
    d_seam_count<RT>(seg.get(), rad.get(), depth.get(), HW, thr, rmax.p ? *rmax.get() : rmax_scalar, mode,
                     (int2 *)(ws.get() + L.seam_counts));
}
__global__ __launch_bounds__(1024) void k_seam_scan(int nb, SP<int32_t> counts, SP<unsigned char> ws, Layout L, int want_nonfinite)
{
    unsigned char *w = ws.get();
    d_seam_scan((const int2 *)(w + L.seam_counts), nb, (int2 *)(w + L.seam_offsets), counts.get(), want_nonfinite);
}
template <typename RT>
__global__ __launch_bounds__(256) void k_seam_scatter(SP<const float> seg, SP<const RT> rad, SP<const uint16_t> depth, int HW,
                                                      int W, double fx, double fy, double cx, double cy, float thr,
                                                      SP<const double> rmax, double rmax_scalar, int mode, SP<double> xyz,
                                                      SP<float> radial_list, int cap, SP<unsigned char> ws, Layout L)
{
    d_seam_scatter<RT>(seg.get(), rad.get(), depth.get(), HW, W, fx, fy, cx, cy, thr, rmax.p ? *rmax.get() : rmax_scalar, mode,
                   (const int2 *)(ws.get() + L.seam_offsets), xyz.get(), radial_list.get(), cap);
}
template <typename RT>
__global__ __launch_bounds__(PREP_THREADS) void k_prep(SP<const double> xyz, SP<const RT> radial, SP<const int32_t> counts,
                                                       int n_arg, int m_arg, int cap, double unit, const double *thr,
                                                       SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_prep<RT>(xyz.get(), radial.get(), counts.get(), n_arg, m_arg, cap, unit, thr, (double *)(w + L.P), (int32_t *)(w + L.ri),
               (PrepPartial *)(w + L.partials));
}
__global__ __launch_bounds__(1024) void k_setup(SP<const int32_t> counts, int n_arg, int m_arg, int cap, double unit,
                                                int bb_enable, long long replica_slots, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_setup((const PrepPartial *)(w + L.partials), (const double *)(w + L.P), counts.get(), n_arg, m_arg, cap, unit, bb_enable,
            replica_slots, (AccHeader *)(w + L.header));
}
__global__ __launch_bounds__(256) void k_bb_init(int cap, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_bb_init((const AccHeader *)(w + L.header), (const double *)(w + L.P), cap, (double *)(w + L.Q),
              (BBEntry *)(w + L.bb_lists) + BB_HALF /* the roots are single blocks */, (float *)(w + L.F), L.capF);
}
// which candidate lists a level kernel reads / writes (all live in the item's workspace)
struct BBIo {
    int in_sel;      // 0: breadth-first list `level & 1`, 1: probe selection
    int out_sel;     // 0: breadth-first list `(level + 1) & 1`, 1: probe children list `level & 1`
};
__device__ __forceinline__ void bb_resolve(unsigned char *w, const Layout &L, int level, BBIo io, const BBEntry *&in,
                                           const uint32_t *&in_count, int &in_cap, BBEntry *&out, uint32_t *&out_count,
                                           int &out_cap)
{
    AccHeader *hdr = (AccHeader *)(w + L.header);
    BBEntry *la = (BBEntry *)(w + L.bb_lists), *lb = la + BB_CAP;
    BBEntry *psel = (BBEntry *)(w + L.bb_probe);
    BBEntry *pkids[2] = {psel + 16, psel + 16 + 8 * BB_BEAM};
    if (io.in_sel) { in = psel; in_count = &hdr->sel_count[level]; in_cap = BB_BEAM; }
    else { in = (level & 1) ? lb : la; in_count = &hdr->bb_count[level]; in_cap = BB_CAP; }
    if (io.out_sel) { out = pkids[level & 1]; out_count = &hdr->pb_count[level]; out_cap = 8 * BB_BEAM; }
    else { out = (level & 1) ? la : lb; out_count = &hdr->bb_count[level + 1]; out_cap = BB_CAP; }
}
template <bool LAST, bool WITH_LB, int NT>
__global__ __launch_bounds__(NT) void k_bb_level(int cap, int level, BBIo io, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    const BBEntry *in; const uint32_t *in_count; int in_cap; BBEntry *out; uint32_t *out_count; int out_cap;
    bb_resolve(w, L, level, io, in, in_count, in_cap, out, out_count, out_cap);
    if (io.in_sel && ((const AccHeader *)(w + L.header))->probe_skip) return;  // a probe launch, and the probe is off
    d_bb_level<LAST, WITH_LB, NT>((AccHeader *)(w + L.header), (const double *)(w + L.Q), (const double *)(w + L.P), cap, in,
                                  in_count, in_cap, out, out_count, out_cap, level);
}
// PROBE: from the probe's selection to the probe's children list (plain lists); otherwise level list -> next level list
template <bool VOX, bool PAIRED_IN, bool PAIR_OUT, bool PROBE = false>
__global__ __launch_bounds__(BBW_THREADS) void k_bb_wide(int cap, int level, SP<unsigned char> ws, Layout L)
{
    static_assert(!PROBE || (!PAIRED_IN && !PAIR_OUT), "the probe's lists are plain");
    unsigned char *w = ws.get();
    AccHeader *hdr = (AccHeader *)(w + L.header);
    BBEntry *la = (BBEntry *)(w + L.bb_lists), *lb = la + BB_CAP;
    BBEntry *psel = (BBEntry *)(w + L.bb_probe);
    BBEntry *pkids[2] = {psel + 16, psel + 16 + 8 * BB_BEAM};
    if constexpr (PROBE) {
        if (hdr->probe_skip) return;
        d_bb_wide<VOX, false, false>(hdr, (const float *)(w + L.F), L.capF, psel, &hdr->sel_count[level], BB_BEAM, pkids[level & 1], nullptr,
                                     &hdr->pb_count[level], 0, 8 * BB_BEAM, level);
    } else {
        const BBEntry *in = ((level & 1) ? lb : la) + (PAIRED_IN ? 0 : BB_HALF);
        const uint32_t *in_count = PAIRED_IN ? &hdr->bb_count[level] : &hdr->bs_count[level];
        d_bb_wide<VOX, PAIRED_IN, PAIR_OUT>(hdr, (const float *)(w + L.F), L.capF, in, in_count, BB_HALF, (level & 1) ? la : lb,
                                            &hdr->bb_count[level + 1], &hdr->bs_count[level + 1], BB_HALF, BB_HALF, level);
    }
}
__global__ __launch_bounds__(BB_SEED_THREADS) void k_bb_seed(int cap, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_bb_seed((AccHeader *)(w + L.header), (const double *)(w + L.Q), (const double *)(w + L.P), (const int32_t *)(w + L.ri), cap);
}
// After the round around the seed: a voxel that an eighth of all points vote for is the peak's neighbourhood for every purpose
// of the search (its count is the lower bound the wide levels prune with) -- the greedy probe, which exists to find such a
// voxel, is skipped.  Otherwise (no seed, outliers, several weak maxima) the probe runs as before.
//
// Flat-landscape decision (r05; mode 2, launched behind the second start of the lower bound -- d_bb_restart + line search + two climb
// rounds on the 8-voxel candidates -- and before the wide level that scores their children, the first expensive one).  What the
// fine levels cost is set by how sharply the peak stands out: with f = (best exact count) / N the search scores about 500 / f^2
// blocks against all N points (sweep, 216 items on the bench cloud: f = 0.33 -> 4.6 k blocks, 0.16 -> 16 k, 0.086 -> 60 k; radial
// noise of sigma_v voxels gives f ~ 0.43 / (sqrt(2 pi) sigma_v)), at BB_PS_PER_BLOCK_POINT each, while the rasteriser evaluates one
// interval per (point, column inside the shell's disc) ~ N pi 0.6 r_max^2 at RAST_PS_PER_POINT_COLUMN, whatever the noise.  N
// cancels: the search is the dearer engine when 500 / f^2 * 1.1 > 4.8 pi 0.6 r_max^2, i.e. f r_max < 7.8; then it stops here and
// k_vote decides (identical result by construction).  Why not earlier: before the restart the lower bound can sit on a side
// lobe -- with 10 % false-positive pixels two thirds of the items reach k_bb_gate(1) at f ~ 0.04 and end at f = 0.29; deciding
// there sent them to the rasteriser (33 -> 136 ms per 216 items, profiles/r05_vote_sweep_gate1.json).  d_bb_restart's survivor
// count (BB_HANDOVER_R2 r_max^2 four-voxel candidates, from the same constants) stays as the second line.
__global__ void k_bb_gate(int mode, SP<unsigned char> ws, Layout L)
{
    AccHeader *hdr = (AccHeader *)(ws.get() + L.header);
    if (threadIdx.x == 0) {
        if (mode == 0) hdr->probe_skip = (hdr->seed != 0 && (long long)hdr->bb_best * 8 >= (long long)hdr->n) ? 1 : 0;
        else if (mode == 1) {  // before round 1: the probe's best
            if (hdr->bb_best_key > hdr->climb_key[0]) hdr->climb_key[0] = hdr->bb_best_key;
        } else {               // mode 2: the flat-landscape decision (above)
            hdr->gate_best = hdr->bb_best;
            if (hdr->status == RCV_OK && hdr->bb_enabled && hdr->variant == 0 && !hdr->bb_overflow &&
                (double)hdr->bb_best * hdr->rmax < BB_FLAT_F_RMAX * (double)hdr->n) {
                hdr->bb_overflow = 1;  // (every later kernel of the search skips its work; k_bb_final leaves the item to k_vote)
                hdr->bb_handover = 2;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_bb_climb(int cap, int round, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_bb_climb((AccHeader *)(w + L.header), (const double *)(w + L.Q), (const double *)(w + L.P), cap, round);
}
__global__ __launch_bounds__(256) void k_bb_line(int cap, int mode, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_bb_line((AccHeader *)(w + L.header), (const double *)(w + L.Q), (const double *)(w + L.P), cap, mode);
}
// the candidate list that level `level` is about to consume (blocks of edge BB_ROOT >> level)
__global__ __launch_bounds__(1024) void k_bb_restart(int level, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    AccHeader *hdr = (AccHeader *)(w + L.header);
    BBEntry *la = (BBEntry *)(w + L.bb_lists), *lb = la + BB_CAP;
    d_bb_restart(hdr, (level & 1) ? lb : la, hdr->bb_count[level], hdr->bs_count[level], BB_ROOT >> level);
}
__global__ __launch_bounds__(BBL_THREADS) void k_bb_exact(int cap, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    AccHeader *hdr = (AccHeader *)(w + L.header);
    BBEntry *la = (BBEntry *)(w + L.bb_lists), *lb = la + BB_CAP;
    d_bb_exact(hdr, (const double *)(w + L.Q), (const double *)(w + L.P), cap, ((BB_LEVELS & 1) ? lb : la) + BB_HALF,
               &hdr->bs_count[BB_LEVELS], BB_HALF);
}
__global__ __launch_bounds__(256) void k_bb_centres(int level, int probe, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    AccHeader *hdr = (AccHeader *)(w + L.header);
    BBEntry *la = (BBEntry *)(w + L.bb_lists), *lb = la + BB_CAP;
    BBEntry *psel = (BBEntry *)(w + L.bb_probe);
    BBEntry *pkids[2] = {psel + 16, psel + 16 + 8 * BB_BEAM};
    if (hdr->probe_skip) return;  // the centre counts only rank blocks for the probe
    // the list that level `level` wrote: the probe's children or the breadth-first list
    if (probe) d_bb_centres(hdr, (const float *)(w + L.F), L.capF, pkids[level & 1], &hdr->pb_count[level], 8 * BB_BEAM, level);
    else {
        BBEntry *lst = ((level + 1) & 1) ? lb : la;
        d_bb_centres(hdr, (const float *)(w + L.F), L.capF, lst, &hdr->bb_count[level + 1], BB_HALF, level);
        d_bb_centres(hdr, (const float *)(w + L.F), L.capF, lst + BB_HALF, &hdr->bs_count[level + 1], BB_HALF, level);
    }
}
// probe selection before level `level`: from the breadth-first list (level 1) or the previous probe children
__global__ __launch_bounds__(1024) void k_bb_topk(int level, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    AccHeader *hdr = (AccHeader *)(w + L.header);
    BBEntry *la = (BBEntry *)(w + L.bb_lists), *lb = la + BB_CAP;
    BBEntry *psel = (BBEntry *)(w + L.bb_probe);
    BBEntry *pkids[2] = {psel + 16, psel + 16 + 8 * BB_BEAM};
    if (hdr->probe_skip) return;
    if (level == 1) d_bb_topk(hdr, lb, &hdr->bb_count[1], &hdr->bs_count[1], BB_HALF, BB_HALF, psel, &hdr->sel_count[level]);
    else d_bb_topk(hdr, pkids[(level - 1) & 1], &hdr->pb_count[level - 1], nullptr, 0, 8 * BB_BEAM, psel, &hdr->sel_count[level]);
}
__global__ __launch_bounds__(1024) void k_bb_final(SP<rcv_acc_result> out, SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    BBEntry *la = (BBEntry *)(w + L.bb_lists), *lb = la + BB_CAP;
    d_bb_final((AccHeader *)(w + L.header), ((BB_LEVELS & 1) ? lb : la) + BB_HALF, out.get());
}
__global__ __launch_bounds__(VOTE_THREADS) void k_vote(int cap, SP<uint32_t> grid, unsigned long long grid_cells,
                                                       SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_vote((AccHeader *)(w + L.header), (const double *)(w + L.P), cap, w + L.replicas, (uint64_t *)(w + L.tile_best),
                 (uint32_t *)(w + L.tile_ties), grid.get(), grid_cells);
}
__global__ __launch_bounds__(VOTE_THREADS) void k_merge(SP<uint32_t> grid, unsigned long long grid_cells, SP<unsigned char> ws,
                                                        Layout L)
{
    unsigned char *w = ws.get();
    d_merge((const AccHeader *)(w + L.header), w + L.replicas, (uint64_t *)(w + L.tile_best), (uint32_t *)(w + L.tile_ties),
                  grid.get(), grid_cells);
}
__global__ __launch_bounds__(256) void k_final(int grid_given, unsigned long long grid_cells, SP<rcv_acc_result> out,
                                               SP<unsigned char> ws, Layout L)
{
    unsigned char *w = ws.get();
    d_final((const AccHeader *)(w + L.header), (const uint64_t *)(w + L.tile_best), (const uint32_t *)(w + L.tile_ties), grid_given,
            grid_cells, out.get());
}


// ---------------------------------------------------------------------------------------------
// RGB-only variant (SURVEY f-5): rgbd_to_point_cloud_no_depth (AccumulatorSpace.py:100-120) and
// Accumulator_3D_no_depth / fast_for_no_depth (:421-479).  Every pixel is back-projected at each millimetre of a depth
// range; all Z * N hypotheses vote with the (unrounded) radius of their pixel.  The arithmetic the reference actually
// performs on the [Z,N,3] array is kept, quirks included (include/rcvpose_hip.h: rcv_accumulate_3d_no_depth).
// The voting itself runs in the engines above: a hypothesis is a point with its own float64 threshold pair.
// ---------------------------------------------------------------------------------------------
struct NdRange {
    double zmin, zmax;
    int32_t n, pad_;
};

// nonzero pixels of a float64 depth map in row-major order (np.nonzero) + their depth range.  One workgroup walks the
// image in order (a data-preparation step off the hot path).
__global__ __launch_bounds__(1024) void k_nd_pixels(const double *depth, int HW, int32_t *pix_out, int cap, NdRange *range)
{
    __shared__ int wc[16];
    __shared__ int running;
    __shared__ double smin[16], smax[16];
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double mn = INF, mx = -INF;
    for (int base = 0; base < HW; base += 1024) {
        const int p = base + threadIdx.x;
        const bool hit = p < HW && depth[p] != 0.0;
        if (hit) { mn = fmin(mn, depth[p]); mx = fmax(mx, depth[p]); }
        const uint64_t b = __ballot(hit);
        if (lane_id() == 0) wc[wave_id()] = __popcll(b);
        __syncthreads();
        int before = running;
        for (int q = 0; q < wave_id(); q++) before += wc[q];
        if (hit) {
            const int dst = before + __popcll(b & ((1ull << lane_id()) - 1ull));
            if (dst < cap) pix_out[dst] = p;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < 16; q++) t += wc[q]; running += t; }
        __syncthreads();
    }
    mn = wave_min(mn); mx = wave_max(mx);
    if (lane_id() == 0) { smin[wave_id()] = mn; smax[wave_id()] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 16; q++) { mn = fmin(mn, smin[q]); mx = fmax(mx, smax[q]); }
        range->zmin = mn; range->zmax = mx; range->n = running; range->pad_ = 0;
    }
}

// pts[i][q] = (((u - cx) * z) / fx, ((v - cy) * z) / fy, z) with z = zmin + i * 0.001  (AccumulatorSpace.py:110-113)
__global__ __launch_bounds__(256) void k_nd_hypotheses(const int32_t *pix, int n, int Z, int W, double fx, double fy, double cx, double cy,
                                                       double zmin, double *pts)
{
    const long long total = (long long)Z * n;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int i = (int)(t / n), q = (int)(t - (long long)i * n);
        const int p = pix[q], v = p / W, u = p - v * W;
        const double z = zmin + (double)i * 0.001;
        pts[3 * t + 0] = (((double)u - cx) * z) / fx;
        pts[3 * t + 1] = (((double)v - cy) * z) / fy;
        pts[3 * t + 2] = z;
    }
}

// the shell of a real radius R as a threshold pair on d2 (float64): vote <=> T_lo <= d2 < T_hi, where
//   T_hi = min{ d : sqrt(d) >= R }      (`radius - distance > 0`  <=>  d2 < T_hi)
//   T_lo = min{ d : R - sqrt(d) < f }   (`radius - distance < factor`)
// found with the device's correctly rounded float64 sqrt and subtraction -- the operations of the reference's predicate
// (:431-432) -- by bisection on the bit pattern (both predicates are monotone in d).
__device__ void nd_thresholds(double R, double f, double &tlo, double &thi)
{
    tlo = 0.0; thi = 0.0;  // R <= 0 (or NaN): `radius - distance > 0` never holds
    if (!(R > 0.0)) return;
    if (!(R < 1.0e150)) { tlo = 0.0; thi = __longlong_as_double(0x7FEFFFFFFFFFFFFFll); return; }
    // predicate true at hi_u, false at lo_u
    unsigned long long lo_u = 0ull, hi_u = (unsigned long long)__double_as_longlong(R * R * 1.0000001 + 1e-300);
    while (hi_u - lo_u > 1) {
        const unsigned long long mid = lo_u + (hi_u - lo_u) / 2;
        if (sqrt(__longlong_as_double((long long)mid)) >= R) hi_u = mid; else lo_u = mid;
    }
    thi = __longlong_as_double((long long)hi_u);
    lo_u = 0ull;  // at d = +0: R - 0 < f ?  (tiny radii: every d2 < T_hi votes)
    if (R - 0.0 < f) { tlo = 0.0; return; }
    hi_u = (unsigned long long)__double_as_longlong(thi);  // at T_hi: R - sqrt(d) <= 0 < f
    while (hi_u - lo_u > 1) {
        const unsigned long long mid = lo_u + (hi_u - lo_u) / 2;
        if (R - sqrt(__longlong_as_double((long long)mid)) < f) hi_u = mid; else lo_u = mid;
    }
    tlo = __longlong_as_double((long long)hi_u);
}

// step 1: xyz * 200 into the SoA point arrays, the three points the reference recentres gathered for their means,
// per-pixel radii (* 20, float32 lists stay float32) and their threshold pairs, radius maximum
template <typename RT>
__global__ __launch_bounds__(PREP_THREADS) void k_nd_prep(const double *xyz_m, const RT *radial, int Z, int N, int m, int cap, double scale,
                                                          double *P, double *Q, double *Tq, PrepPartial *partials)
{
    __shared__ double red[PREP_THREADS / 64];
    const long long total = (long long)Z * N;
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    volatile double three = 3.0;
    const double f = sqrt(three) / 4.0;  // (3**0.5)/4, :423
    double rmax = -INF;
    for (long long t = (long long)blockIdx.x * PREP_THREADS + threadIdx.x; t < total || t < m; t += (long long)PREP_BLOCKS * PREP_THREADS) {
        if (t < total) {
            const int z = (int)(t / N), p = (int)(t - (long long)z * N);
            for (int c = 0; c < 3; c++) {
                const double v = xyz_m[3 * t + c] * scale;  // xyz*200, :438
                P[(size_t)c * cap + t] = v;
                if (p < 3) Q[(size_t)p * cap + 3 * z + c] = v;  // xyz_mm[:, p] as numpy flattens it (C order)
            }
        }
        if (t < m) {
            const RT rv = radial[t] * (RT)20;  // radial_list*20, :447 (weak python scalar: float32 stays float32)
            const double rd = (double)rv;
            if (rd > rmax) rmax = rd;
            if (t < N) {
                double tlo, thi;
                nd_thresholds(rd, f, tlo, thi);
                Tq[2 * t] = tlo;   // per pixel; replicated over the hypotheses by k_nd_finish
                Tq[2 * t + 1] = thi;
            }
        }
    }
    rmax = wave_max(rmax);
    if (lane_id() == 0) red[wave_id()] = rmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PREP_THREADS / 64; w++) rmax = fmax(rmax, red[w]);
        PrepPartial pp;
        memset(&pp, 0, sizeof(pp));
        pp.rmax = rmax;
        partials[blockIdx.x] = pp;
    }
}

// step 2 (one workgroup): the three means, the recentring of points 0..2 of every hypothesis, extrema, grid
__global__ __launch_bounds__(1024) void k_nd_setup(int Z, int N, int m, int cap, int bb_enable, long long replica_slots, double *P,
                                                   const double *Q, const PrepPartial *partials, AccHeader *hdr)
{
    __shared__ double total[3];
    __shared__ double s_mn[16], s_mx[16];
    const int t = threadIdx.x;
    np_sums3(Q, (size_t)cap, 3 * Z, total);   // np.mean(xyz_mm[:, a]) over the [Z,3] view, a = 0..2  (:440-442)
    double mean[3];
    for (int a = 0; a < 3; a++) mean[a] = total[a] / (double)(3 * Z);
    for (int e = t; e < 9 * Z; e += 1024) {   // xyz_mm[:, a] -= mean_a  (:443-445): points 0..2, all coordinates
        const int z = e / 9, a = (e / 3) % 3, c = e % 3;
        P[(size_t)c * cap + (size_t)z * N + a] -= mean[a];
    }
    __syncthreads();
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double mn = INF, mx = -INF;
    const long long n = (long long)Z * N;
    for (int c = 0; c < 3; c++)
        for (long long i = t; i < n; i += 1024) { const double v = P[(size_t)c * cap + i]; mn = fmin(mn, v); mx = fmax(mx, v); }
    mn = wave_min(mn); mx = wave_max(mx);
    if (lane_id() == 0) { s_mn[wave_id()] = mn; s_mx[wave_id()] = mx; }
    __syncthreads();
    if (t != 0) return;
    for (int q = 1; q < 16; q++) { mn = fmin(mn, s_mn[q]); mx = fmax(mx, s_mx[q]); }
    double rmax = -INF;
    for (int b = 0; b < PREP_BLOCKS; b++) rmax = fmax(rmax, partials[b].rmax);
    __shared__ AccHeader sH;  // (in LDS, not in scratch memory: d_setup)
    AccHeader &H = sH;
    memset(&H, 0, sizeof(H));
    H.variant = 1;
    H.n = (int)n; H.m = m; H.unit = 5.0;
    for (int a = 0; a < 3; a++) H.nd_mean[a] = mean[a];
    H.rmax = rmax;
    const int zb = clamp_to_int(mn - rmax) + 1;            // :453
    H.zb = zb;
    const double pmax = zb < 0 ? mx - (double)zb : mx;     // :456-457 (the shift itself is applied by k_nd_finish)
    const long long dim = (long long)clamp_to_int(pmax) + 1;  // :459 -- no radius margin
    finish_header(H, dim, (int)n, false, bb_enable, replica_slots);
    *hdr = H;
}

// step 3: the zero_boundary shift into the stored coordinates (the engines then see mean = shift = 0) and the pixel's
// threshold pair for every hypothesis
__global__ __launch_bounds__(256) void k_nd_finish(int Z, int N, int cap, double *P, const double *Tq, const AccHeader *hdr)
{
    const AccHeader H = *hdr;
    const long long n = (long long)Z * N;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        if (H.zb < 0)
            for (int c = 0; c < 3; c++) P[(size_t)c * cap + t] = P[(size_t)c * cap + t] - (double)H.zb;  // xyz_mm - zero_boundary
        const int p = (int)(t % N);
        P[(size_t)3 * cap + t] = Tq[2 * p];
        P[(size_t)4 * cap + t] = Tq[2 * p + 1];
    }
}

constexpr size_t VOTE_LDS_BYTES = CELL_BYTES + 12 * VOTE_WAVES + 64;
bool g_vote_attr_set = false;

}  // namespace
}  // namespace rcv

using namespace rcv;

extern "C" {

int rcv_init(void) { return ensure_init(); }

size_t rcv_acc_workspace_bytes(int max_points)
{
    if (max_points < 1) max_points = 1;
    return make_layout(max_points).total;
}

size_t rcv_acc_workspace_bytes_min(int max_points)
{
    if (max_points < 1) max_points = 1;
    return make_layout(max_points).min_total;
}

static int seam_batch(rcv_strided seg, rcv_strided rad, rcv_strided depth, int H, int W, const double K[4], float seg_thr,
                      rcv_strided rmax, double rmax_scalar, int mode, rcv_strided xyz, rcv_strided rlist, rcv_strided counts,
                      int cap, rcv_strided ws, size_t ws_bytes, int n_frames, int n_kp, void *stream)
{
    if (!depth.ptr || !K || !xyz.ptr || !counts.ptr || !ws.ptr) return RCV_E_INVALID;
    const bool rad_f16 = (mode & RCV_SEAM_RADIAL_F16) != 0;
    const int want_nonfinite = (mode & RCV_SEAM_COUNT_NONFINITE) != 0;
    mode &= ~(RCV_SEAM_RADIAL_F16 | RCV_SEAM_COUNT_NONFINITE);
    if (mode < 0 || mode > 3) return RCV_E_INVALID;
    if (mode != 2 && (!rad.ptr || !rlist.ptr)) return RCV_E_INVALID;
    if ((mode == 0 || mode == 3) && !seg.ptr) return RCV_E_INVALID;
    if (H <= 0 || W <= 0 || cap <= 0 || n_frames <= 0 || n_kp <= 0 || n_frames > 65535 || n_kp > 65535) return RCV_E_INVALID;
    const long long HW = (long long)H * W;
    const int nb = (int)((HW + SEAM_PIX_PER_BLOCK - 1) / SEAM_PIX_PER_BLOCK);
    if (nb > SEAM_MAX_BLOCKS) return RCV_E_TOO_LARGE;
    const Layout L = make_layout(cap);
    if (ws_bytes < L.tile_best) return RCV_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!seg.ptr) seg = rad;  // never dereferenced when mode != 0
    const SP<const float> sseg{(const float *)seg.ptr, seg.stride_frame, seg.stride_kp};
    const SP<const float> srad{(const float *)rad.ptr, rad.stride_frame, rad.stride_kp};
    const SP<const _Float16> srad16{(const _Float16 *)rad.ptr, rad.stride_frame, rad.stride_kp};
    const SP<const uint16_t> sdep{(const uint16_t *)depth.ptr, depth.stride_frame, depth.stride_kp};
    const SP<const double> srmax{(const double *)rmax.ptr, rmax.stride_frame, rmax.stride_kp};
    const SP<unsigned char> sws{(unsigned char *)ws.ptr, ws.stride_frame, ws.stride_kp};
    if (rad_f16)
        hipLaunchKernelGGL(k_seam_count<_Float16>, dim3(nb, n_frames, n_kp), dim3(256), 0, st, sseg, srad16, sdep, (int)HW, seg_thr,
                           srmax, rmax_scalar, mode, sws, L);
    else
        hipLaunchKernelGGL(k_seam_count<float>, dim3(nb, n_frames, n_kp), dim3(256), 0, st, sseg, srad, sdep, (int)HW, seg_thr, srmax,
                           rmax_scalar, mode, sws, L);
    RCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_seam_scan, dim3(1, n_frames, n_kp), dim3(1024), 0, st, nb,
                       SP<int32_t>{(int32_t *)counts.ptr, counts.stride_frame, counts.stride_kp}, sws, L, want_nonfinite);
    RCV_LAUNCH_CHECK();
    const SP<double> sxyz{(double *)xyz.ptr, xyz.stride_frame, xyz.stride_kp};
    const SP<float> srl{(float *)rlist.ptr, rlist.stride_frame, rlist.stride_kp};
    if (rad_f16)
        hipLaunchKernelGGL(k_seam_scatter<_Float16>, dim3(nb, n_frames, n_kp), dim3(256), 0, st, sseg, srad16, sdep, (int)HW, W, K[0],
                           K[1], K[2], K[3], seg_thr, srmax, rmax_scalar, mode, sxyz, srl, cap, sws, L);
    else
        hipLaunchKernelGGL(k_seam_scatter<float>, dim3(nb, n_frames, n_kp), dim3(256), 0, st, sseg, srad, sdep, (int)HW, W, K[0], K[1],
                           K[2], K[3], seg_thr, srmax, rmax_scalar, mode, sxyz, srl, cap, sws, L);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

// the arg-max engines on a prepared item (header + point arrays in the workspace): exact peak search, LDS rasteriser
static int run_engines(int cap, int bb_enable, SP<rcv_acc_result> sres, SP<uint32_t> sgrid, size_t grid_cells, bool grid_given,
                       SP<unsigned char> sws, const Layout &L, int n_frames, int n_kp, hipStream_t st)
{
    const dim3 items(1, n_frames, n_kp);
    auto grid3 = [&](int x) { return dim3(x, n_frames, n_kp); };
    const int n_items = n_frames * n_kp;
    if (bb_enable) {
        hipLaunchKernelGGL(k_bb_init, grid3(256), dim3(256), 0, st, cap, sws, L);
        RCV_LAUNCH_CHECK();
        // a least-squares estimate of the keypoint's voxel, one round of exact counts around it, and the decision whether the
        // greedy probe is still needed (r02; see d_bb_seed / k_bb_gate)
        hipLaunchKernelGGL(k_bb_seed, grid3(1), dim3(BB_SEED_THREADS), 0, st, cap, sws, L);
        RCV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_bb_climb, grid3(BB_CLIMB_W), dim3(256), 0, st, cap, 0, sws, L);
        RCV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_bb_gate, items, dim3(64), 0, st, 0, sws, L);
        RCV_LAUNCH_CHECK();
        // no strong peak around the seed: look for the vote sheet along its normal through the seed, and decide again
        hipLaunchKernelGGL(k_bb_line, grid3(5), dim3(256), 0, st, cap, 0, sws, L);
        RCV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_bb_gate, items, dim3(64), 0, st, 0, sws, L);
        RCV_LAUNCH_CHECK();
        // level 0 of the breadth-first search scores the 32-voxel blocks (children of the 64-voxel roots): upper bounds with
        // the wide kernel, then the centre counts that rank them for the probe (r02: was one k_bb_level launch with an exact
        // centre count per child, 2.0 ms per 216 items)
        const int wide_grid0 = n_items >= 8 ? 64 : 2048 / n_items;
        hipLaunchKernelGGL((k_bb_wide<false, false, true>), grid3(wide_grid0), dim3(BBW_THREADS), 0, st, cap, 0, sws, L);
        RCV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_bb_centres, grid3(64), dim3(256), 0, st, 0, 0, sws, L);
        RCV_LAUNCH_CHECK();
        // greedy probe: follow the BB_BEAM most promising blocks of each level down to voxels; the exact
        // counts found there give the search a lower bound close to the peak before it widens
        for (int level = 1; level < BB_LEVELS; level++) {
            hipLaunchKernelGGL(k_bb_topk, items, dim3(1024), 0, st, level, sws, L);
            RCV_LAUNCH_CHECK();
            if (level == BB_LEVELS - 1) {  // voxel level: exact counts (the lower bound of the search)
                hipLaunchKernelGGL((k_bb_level<true, true, 1024>), grid3(BB_BEAM), dim3(1024), 0, st, cap, level, BBIo{1, 1}, sws, L);
                RCV_LAUNCH_CHECK();
            } else {  // upper bounds of the selected blocks' children + the centre counts that rank them
                hipLaunchKernelGGL((k_bb_wide<false, false, false, true>), grid3((BB_BEAM + BBW_NP - 1) / BBW_NP), dim3(BBW_THREADS), 0, st,
                                   cap, level, sws, L);
                RCV_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_bb_centres, grid3(BB_BEAM), dim3(256), 0, st, level, 1, sws, L);
                RCV_LAUNCH_CHECK();
            }
        }
        hipLaunchKernelGGL(k_bb_gate, items, dim3(64), 0, st, 1, sws, L);  // freeze the start of round 1 (see AccHeader::climb_key)
        RCV_LAUNCH_CHECK();
        for (int round = 1; round < BB_CLIMB_ROUNDS; round++) {  // exact hill-climb from the best voxel so far
            hipLaunchKernelGGL(k_bb_climb, grid3(BB_CLIMB_W), dim3(256), 0, st, cap, round, sws, L);
            RCV_LAUNCH_CHECK();
        }
        // the lower bound now sits at (or next to) the peak: the wide levels only need upper bounds
        const int wide_grid = n_items >= 8 ? 256 : 2048 / n_items;
        // (the last of them, on 2^3 parents, bounds single voxels)
        int extra_round = BB_CLIMB_ROUNDS;  // (even: the restart's line search fills slot 1, which round `extra_round` reads)
        static_assert(BB_CLIMB_ROUNDS % 2 == 0, "restart rounds read climb_key[1]");
        for (int level = 1; level < BB_LEVELS; level++) {
            if (level == 3 || level == 4) {  // the 8- and 4-voxel candidates exist: second start of the lower bound (d_bb_restart)
                hipLaunchKernelGGL(k_bb_restart, items, dim3(1024), 0, st, level, sws, L);
                RCV_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_bb_line, grid3(5), dim3(256), 0, st, cap, 1, sws, L);
                RCV_LAUNCH_CHECK();
                for (int r = 0; r < 2; r++) {
                    hipLaunchKernelGGL(k_bb_climb, grid3(BB_CLIMB_W), dim3(256), 0, st, cap, extra_round + r, sws, L);
                    RCV_LAUNCH_CHECK();
                }
                extra_round += 2;
                if (level == 3) {  // the lower bound is as good as it gets: flat landscape -> the rasteriser (k_bb_gate, mode 2)
                    hipLaunchKernelGGL(k_bb_gate, items, dim3(64), 0, st, 2, sws, L);
                    RCV_LAUNCH_CHECK();
                }
            }
            // every level list is two regions, x-sibling pairs and single blocks, each with its own launch; the level before
            // the voxel level and the voxel level itself write single entries only
            if (level == BB_LEVELS - 1) {
                hipLaunchKernelGGL((k_bb_wide<true, false, false>), grid3(wide_grid), dim3(BBW_THREADS), 0, st, cap, level, sws, L);
            } else if (level == BB_LEVELS - 2) {
                hipLaunchKernelGGL((k_bb_wide<false, true, false>), grid3(wide_grid), dim3(BBW_THREADS), 0, st, cap, level, sws, L);
                RCV_LAUNCH_CHECK();
                hipLaunchKernelGGL((k_bb_wide<false, false, false>), grid3(wide_grid), dim3(BBW_THREADS), 0, st, cap, level, sws, L);
            } else {
                hipLaunchKernelGGL((k_bb_wide<false, true, true>), grid3(wide_grid), dim3(BBW_THREADS), 0, st, cap, level, sws, L);
                RCV_LAUNCH_CHECK();
                hipLaunchKernelGGL((k_bb_wide<false, false, true>), grid3(wide_grid), dim3(BBW_THREADS), 0, st, cap, level, sws, L);
            }
            RCV_LAUNCH_CHECK();
        }
        // exact counts of the surviving voxels (float32 bound >= best exact count): typically a few dozen
        hipLaunchKernelGGL(k_bb_exact, grid3((n_items >= 8 ? 4 : 32) * BBL_PARTS), dim3(BBL_THREADS), 0, st, cap, sws, L);
        RCV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_bb_final, items, dim3(1024), 0, st, sres, sws, L);
        RCV_LAUNCH_CHECK();
    }
    const int vote_blocks = n_items >= 4 ? 128 : 512 / n_items;  // work items are handed out by a per-item counter
    hipLaunchKernelGGL(k_vote, grid3(vote_blocks), dim3(VOTE_THREADS), VOTE_LDS_BYTES, st, cap, sgrid,
                       (unsigned long long)grid_cells, sws, L);
    RCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_merge, grid3(vote_blocks), dim3(VOTE_THREADS), VOTE_LDS_BYTES, st, sgrid,
                       (unsigned long long)grid_cells, sws, L);
    RCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_final, items, dim3(256), 0, st, grid_given ? 1 : 0, (unsigned long long)grid_cells, sres, sws, L);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

static int accumulate_batch(rcv_strided xyz, rcv_strided radial, int radial_is_f64, rcv_strided counts, int n_points, int n_radii,
                            double acc_unit_mm, int engine, rcv_strided result, rcv_strided grid, size_t grid_cells, rcv_strided ws,
                            size_t ws_bytes, int n_frames, int n_kp, void *stream)
{
    if (!result.ptr || !ws.ptr) return RCV_E_INVALID;
    if (n_points < 0 || n_radii < 0 || !(acc_unit_mm > 0.0)) return RCV_E_INVALID;
    if (engine != RCV_VOTE_AUTO && engine != RCV_VOTE_RASTER) return RCV_E_INVALID;
    if ((n_points > 0 && !xyz.ptr) || (n_radii > 0 && !radial.ptr)) return RCV_E_INVALID;
    if (n_frames <= 0 || n_kp <= 0 || n_frames > 65535 || n_kp > 65535) return RCV_E_INVALID;
    int rc = ensure_init();
    if (rc != RCV_OK) return rc;
    const int cap = n_points > n_radii ? (n_points > 0 ? n_points : 1) : (n_radii > 0 ? n_radii : 1);
    const Layout L = make_layout(cap);
    if (ws_bytes < L.min_total) return RCV_E_WORKSPACE;
    const long long replica_slots = (long long)((ws_bytes - L.replicas) / CELL_BYTES);
    hipStream_t st = (hipStream_t)stream;
    if (!g_vote_attr_set) {
        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_vote, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VOTE_LDS_BYTES));
        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VOTE_LDS_BYTES));
        g_vote_attr_set = true;
    }
    const SP<unsigned char> sws{(unsigned char *)ws.ptr, ws.stride_frame, ws.stride_kp};
    const SP<const double> sxyz{(const double *)xyz.ptr, xyz.stride_frame, xyz.stride_kp};
    const SP<const int32_t> scnt{(const int32_t *)counts.ptr, counts.stride_frame, counts.stride_kp};
    const SP<rcv_acc_result> sres{(rcv_acc_result *)result.ptr, result.stride_frame, result.stride_kp};
    const SP<uint32_t> sgrid{(uint32_t *)grid.ptr, grid.stride_frame, grid.stride_kp};
    const dim3 items(1, n_frames, n_kp);
    auto grid3 = [&](int x) { return dim3(x, n_frames, n_kp); };
    if (radial_is_f64)
        hipLaunchKernelGGL(k_prep<double>, grid3(PREP_BLOCKS), dim3(PREP_THREADS), 0, st, sxyz,
                           SP<const double>{(const double *)radial.ptr, radial.stride_frame, radial.stride_kp}, scnt, n_points,
                           n_radii, cap, acc_unit_mm, g_thr_dev, sws, L);
    else
        hipLaunchKernelGGL(k_prep<float>, grid3(PREP_BLOCKS), dim3(PREP_THREADS), 0, st, sxyz,
                           SP<const float>{(const float *)radial.ptr, radial.stride_frame, radial.stride_kp}, scnt, n_points,
                           n_radii, cap, acc_unit_mm, g_thr_dev, sws, L);
    RCV_LAUNCH_CHECK();
    // the exact peak search runs first unless the caller wants the full grid (or forced the rasteriser)
    const int bb_enable = (engine == RCV_VOTE_AUTO && grid.ptr == nullptr) ? 1 : 0;
    hipLaunchKernelGGL(k_setup, items, dim3(1024), 0, st, scnt, n_points, n_radii, cap, acc_unit_mm, bb_enable, replica_slots, sws,
                       L);
    RCV_LAUNCH_CHECK();
    return run_engines(cap, bb_enable, sres, sgrid, grid_cells, grid.ptr != nullptr, sws, L, n_frames, n_kp, st);
}

int rcv_union_depth(const float *seg_dev, const float *radial_dev, long long kp_stride, const uint16_t *depth_dev, int H, int W,
                    float seg_thr, const double *rmax_dm_dev, int mode, int n_keypoints, uint16_t *depth_out_dev, void *stream)
{
    if (!radial_dev || !depth_dev || !depth_out_dev || !rmax_dm_dev || H <= 0 || W <= 0 || n_keypoints <= 0) return RCV_E_INVALID;
    if ((mode != 0 && mode != 1) || (mode == 0 && !seg_dev)) return RCV_E_INVALID;
    const int HW = H * W;
    hipLaunchKernelGGL(k_union_depth, dim3(std::min((HW + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       seg_dev, radial_dev, kp_stride, depth_dev, HW, seg_thr, rmax_dm_dev, mode, n_keypoints, depth_out_dev);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int rcv_seam_compact(const float *seg_dev, const float *radial_dev, const uint16_t *depth_dev, int H, int W,
                     const double K[4], float seg_thr, double rmax_dm, int mode, double *xyz_m_dev,
                     float *radial_list_dev, int32_t *counts_dev, int cap, void *ws, size_t ws_bytes, void *stream)
{
    const rcv_strided none{nullptr, 0, 0};
    return seam_batch(rcv_strided{(void *)seg_dev, 0, 0}, rcv_strided{(void *)radial_dev, 0, 0}, rcv_strided{(void *)depth_dev, 0, 0}, H,
                      W, K, seg_thr, none, rmax_dm, mode, rcv_strided{xyz_m_dev, 0, 0}, rcv_strided{radial_list_dev, 0, 0},
                      rcv_strided{counts_dev, 0, 0}, cap, rcv_strided{ws, 0, 0}, ws_bytes, 1, 1, stream);
}

int rcv_seam_compact_batch(rcv_strided seg, rcv_strided radial, rcv_strided depth, int H, int W, const double K[4], float seg_thr,
                           rcv_strided rmax_dm, int mode, rcv_strided xyz_m, rcv_strided radial_list, rcv_strided counts, int cap,
                           rcv_strided ws, size_t ws_bytes, int n_frames, int n_keypoints, void *stream)
{
    if (!rmax_dm.ptr && mode != 2) return RCV_E_INVALID;
    return seam_batch(seg, radial, depth, H, W, K, seg_thr, rmax_dm, 0.0, mode, xyz_m, radial_list, counts, cap, ws, ws_bytes,
                      n_frames, n_keypoints, stream);
}

int rcv_accumulate_3d(const double *xyz_m_dev, const void *radial_dm_dev, int radial_is_f64,
                      const int32_t *counts_dev, int n_points, int n_radii, double acc_unit_mm, int engine,
                      rcv_acc_result *result_dev, uint32_t *vote_grid_dev, size_t vote_grid_cells, void *ws,
                      size_t ws_bytes, void *stream)
{
    return accumulate_batch(rcv_strided{(void *)xyz_m_dev, 0, 0}, rcv_strided{(void *)radial_dm_dev, 0, 0}, radial_is_f64,
                            rcv_strided{(void *)counts_dev, 0, 0}, n_points, n_radii, acc_unit_mm, engine, rcv_strided{result_dev, 0, 0},
                            rcv_strided{vote_grid_dev, 0, 0}, vote_grid_cells, rcv_strided{ws, 0, 0}, ws_bytes, 1, 1, stream);
}

int rcv_accumulate_3d_batch(rcv_strided xyz_m, rcv_strided radial_dm, int radial_is_f64, rcv_strided counts, int cap,
                            double acc_unit_mm, int engine, rcv_strided result, rcv_strided ws, size_t ws_bytes, int n_frames,
                            int n_keypoints, void *stream)
{
    if (!counts.ptr) return RCV_E_INVALID;
    return accumulate_batch(xyz_m, radial_dm, radial_is_f64, counts, cap, cap, acc_unit_mm, engine, result, rcv_strided{nullptr, 0, 0},
                            0, ws, ws_bytes, n_frames, n_keypoints, stream);
}

int rcv_collect_ties(const uint32_t *vote_grid_dev, int dim, uint32_t peak, int64_t *flat_idx_dev, int cap,
                     int32_t *count_dev, void *stream)
{
    if (!vote_grid_dev || !flat_idx_dev || !count_dev || dim <= 0 || cap <= 0) return RCV_E_INVALID;
    hipLaunchKernelGGL(k_collect_ties, dim3(1), dim3(1024), 0, (hipStream_t)stream, vote_grid_dev,
                       (unsigned long long)dim * dim * dim, peak, flat_idx_dev, cap, count_dev);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int rcv_acc_debug_points(const void *ws, double *points_vox_dev, int32_t *radius_int_dev, int cap, void *stream)
{
    if (!ws || !points_vox_dev || cap <= 0) return RCV_E_INVALID;
    const Layout L = make_layout(cap);
    const char *w = (const char *)ws;
    hipLaunchKernelGGL(k_debug_points, dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const AccHeader *)(w + L.header), (const double *)(w + L.P), (const int32_t *)(w + L.ri), cap, cap,
                       points_vox_dev, radius_int_dev);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int rcv_backproject_no_depth(const double *depth_m_dev, int H, int W, const double K[4], double *pts_dev, long long cap_points,
                             int32_t *zn_out, void *ws, size_t ws_bytes, void *stream)
{
    if (!depth_m_dev || !K || !zn_out || !ws || H <= 0 || W <= 0) return RCV_E_INVALID;
    const long long HW = (long long)H * W;
    if (HW > 0x7fffffffLL) return RCV_E_TOO_LARGE;
    if (ws_bytes < 256 + sizeof(int32_t) * (size_t)HW) return RCV_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    NdRange *range = (NdRange *)ws;
    int32_t *pix = (int32_t *)((char *)ws + 256);
    hipLaunchKernelGGL(k_nd_pixels, dim3(1), dim3(1024), 0, st, depth_m_dev, (int)HW, pix, (int)HW, range);
    RCV_LAUNCH_CHECK();
    // the shape of the result depends on the data (as the reference's return value does): one synchronisation
    NdRange h;
    RCV_HIP_CHECK(hipMemcpyAsync(&h, range, sizeof(h), hipMemcpyDeviceToHost, st));
    RCV_HIP_CHECK(hipStreamSynchronize(st));
    zn_out[0] = 0; zn_out[1] = h.n;
    if (h.n <= 0) return RCV_E_EMPTY;  // the reference raises on zs.min() of an empty array (:103)
    const long long Z = (long long)(h.zmax * 1000.0) + 1 - (long long)(h.zmin * 1000.0);  // :105
    if (Z <= 0 || Z > 0x7fffffffLL / h.n) return RCV_E_TOO_LARGE;
    zn_out[0] = (int32_t)Z;
    if (!pts_dev) return RCV_OK;  // size query
    if (cap_points < Z * h.n) return RCV_E_WORKSPACE;
    const long long total = Z * h.n;
    hipLaunchKernelGGL(k_nd_hypotheses, dim3((unsigned)std::min<long long>((total + 255) / 256, 65535)), dim3(256), 0, st, pix, h.n, (int)Z, W,
                       K[0], K[1], K[2], K[3], h.zmin, pts_dev);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int rcv_accumulate_3d_no_depth(const double *xyz_m_dev, const void *radial_dm_dev, int radial_is_f64, int n_hypotheses, int n_pixels,
                               int n_radii, int engine, rcv_acc_result *result_dev, uint32_t *vote_grid_dev, size_t vote_grid_cells,
                               void *ws, size_t ws_bytes, void *stream)
{
    if (!xyz_m_dev || !radial_dm_dev || !result_dev || !ws) return RCV_E_INVALID;
    if (engine != RCV_VOTE_AUTO && engine != RCV_VOTE_RASTER) return RCV_E_INVALID;
    if (n_hypotheses < 1 || n_pixels < 3 || n_radii < n_pixels) return RCV_E_INVALID;  // the reference raises IndexError below 3 pixels
    if ((long long)n_hypotheses * n_pixels > 0x3fffffffLL) return RCV_E_TOO_LARGE;
    const int cap = n_hypotheses * n_pixels;
    const Layout L = make_layout(cap);
    if (ws_bytes < L.min_total) return RCV_E_WORKSPACE;
    const long long replica_slots = (long long)((ws_bytes - L.replicas) / CELL_BYTES);
    hipStream_t st = (hipStream_t)stream;
    if (!g_vote_attr_set) {
        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_vote, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VOTE_LDS_BYTES));
        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VOTE_LDS_BYTES));
        g_vote_attr_set = true;
    }
    unsigned char *w = (unsigned char *)ws;
    double *P = (double *)(w + L.P), *Q = (double *)(w + L.Q), *Tq = (double *)(w + L.F);
    PrepPartial *partials = (PrepPartial *)(w + L.partials);
    AccHeader *hdr = (AccHeader *)(w + L.header);
    if (radial_is_f64)
        hipLaunchKernelGGL(k_nd_prep<double>, dim3(PREP_BLOCKS), dim3(PREP_THREADS), 0, st, xyz_m_dev, (const double *)radial_dm_dev, n_hypotheses,
                           n_pixels, n_radii, cap, 200.0, P, Q, Tq, partials);
    else
        hipLaunchKernelGGL(k_nd_prep<float>, dim3(PREP_BLOCKS), dim3(PREP_THREADS), 0, st, xyz_m_dev, (const float *)radial_dm_dev, n_hypotheses,
                           n_pixels, n_radii, cap, 200.0, P, Q, Tq, partials);
    RCV_LAUNCH_CHECK();
    const int bb_enable = (engine == RCV_VOTE_AUTO && vote_grid_dev == nullptr) ? 1 : 0;
    hipLaunchKernelGGL(k_nd_setup, dim3(1), dim3(1024), 0, st, n_hypotheses, n_pixels, n_radii, cap, bb_enable, replica_slots, P, Q, partials, hdr);
    RCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nd_finish, dim3(std::min((cap + 255) / 256, 4096)), dim3(256), 0, st, n_hypotheses, n_pixels, cap, P, Tq, hdr);
    RCV_LAUNCH_CHECK();
    return run_engines(cap, bb_enable, SP<rcv_acc_result>{result_dev, 0, 0}, SP<uint32_t>{vote_grid_dev, 0, 0}, vote_grid_cells,
                       vote_grid_dev != nullptr, SP<unsigned char>{w, 0, 0}, L, 1, 1, st);
}

}  // extern "C"
