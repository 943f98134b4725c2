/*
 * rcv_oracle.c -- CPU restatement of the RCVPose voting path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP kernels in rcvpose_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path
 * never links, imports or calls anything in oracle/.
 *
 * Every function restates (in plain C, own code) the arithmetic of the reference
 * aaronWool/rcvpose, file AccumulatorSpace.py, and cites the lines it follows.  It is pinned
 * to the reference by the .npz files of tests/golden, which were produced by running the reference's own
 * source (tests/golden/make_golden.py): exact vote grids, exact tie lists, centres.
 *
 * Arithmetic contract (all IEEE-754 binary64 unless noted, no FMA contraction: build with
 * -ffp-contract=off):
 *   - x**2 is x*x, x**0.5 is sqrt(x) (what numba emits for AccumulatorSpace.py:337).
 *   - np.around is round-half-to-even (rint / rintf), int() truncates toward zero.
 *   - np.mean over a strided float64 column is numpy's pairwise summation divided by n.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_E_EMPTY -1     /* reference raises ValueError (min of an empty array), :390 */
#define ORC_E_BADGRID -2   /* reference raises (negative dimension / max of empty grid), :401,:406 */

typedef struct {
    double mean[3];      /* x_mean_mm, y_mean_mm, z_mean_mm (voxel units), :381-383 */
    double radius_max;   /* radial_list_mm.max() widened to double, :392 */
    int32_t zero_boundary; /* :394 */
    int32_t length;      /* :399 */
    int32_t dim;         /* length + int(radius_max), :401 */
    int32_t status;
} orc_grid_info;

/* numpy's pairwise summation (numpy/core/src/umath/loops_utils.h.src, pairwise_sum_DOUBLE):
 * blocks of <=128 with 8 interleaved partial sums, halves split at a multiple of 8. */
static double pw_sum(const double *a, long n, long stride)
{
    if (n < 8) {
        double r = 0.;
        for (long i = 0; i < n; i++) r += a[i * stride];
        return r;
    }
    if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i * stride];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pw_sum(a, n2, stride) + pw_sum(a + n2 * stride, n - n2, stride);
}

/* np.mean(a[:, c]) for a float64 column view of a C-contiguous [n,3] array: add.reduce starts
 * from the identity 0 and feeds the strided (hence buffered) operand to the inner loop in
 * chunks of the ufunc buffer size, 8192 elements; each chunk is pairwise-summed and added to
 * the running total; the mean is total / n  (numpy/_core/_methods.py:_mean).  Verified
 * bit-for-bit against numpy for n up to 1e6 in tests/test_oracle_golden.py. */
#define NPY_BUFSIZE 8192
double orc_mean_strided(const double *a, long n, long stride)
{
    double s = 0.;
    for (long off = 0; off < n; off += NPY_BUFSIZE) {
        long len = n - off < NPY_BUFSIZE ? n - off : NPY_BUFSIZE;
        s = s + pw_sum(a + off * stride, len, stride);
    }
    return s / (double)n;
}

/* rgbd_to_point_cloud, AccumulatorSpace.py:77-85.  depth: int64 [H,W] (mm) after the mask
 * multiply; K = {fx, fy, cx, cy}.  Output xyz_mm [n,3] in row-major pixel order. */
long orc_backproject(const int64_t *depth, int H, int W, const double *K, double *xyz_mm)
{
    long n = 0;
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            int64_t z = depth[(long)v * W + u];
            if (z == 0) continue;
            double zs = (double)z;
            xyz_mm[3 * n + 0] = (((double)u - K[2]) * zs) / K[0];
            xyz_mm[3 * n + 1] = (((double)v - K[3]) * zs) / K[1];
            xyz_mm[3 * n + 2] = zs;
            n++;
        }
    return n;
}

/* The mask / gather seam, AccumulatorSpace.py:603-610 and :619 (using_ckpts branch):
 *   sem = where(sem>0.8,1,0); sem = where(radial<=rmax, sem, 0); depth *= sem;
 *   xyz_mm = backproject(depth); radial_list = where(radial<=rmax, radial, 0)[sem==1]; xyz = xyz_mm/1000
 * sem, radial: float32 [H,W]; depth uint16 [H,W]; rmax_dm is a float64 (max_radii_dm[k]), the
 * comparison radial<=rmax therefore happens in float64.  Returns the two (possibly different,
 * SURVEY a-4) list lengths through n_points / n_radii. */
void orc_seam(const float *sem, const float *radial, const uint16_t *depth, int H, int W, const double *K,
              double seg_thr, double rmax_dm, double *xyz_m, float *radial_list, long *n_points, long *n_radii)
{
    long np_ = 0, nr = 0;
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            long p = (long)v * W + u;
            /* sem>0.8: python float vs float32 array compares in float32 (weak scalar);
             * radial<=max_radii_dm[k]: float32 array vs np.float64 scalar compares in float64
             * (numpy >= 2 promotion, the numpy the golden vectors were generated with). */
            int m = (sem[p] > (float)seg_thr) && ((double)radial[p] <= rmax_dm);
            if (!m) continue;
            radial_list[nr++] = radial[p];
            int64_t z = (int64_t)depth[p];
            if (z == 0) continue;
            double zs = (double)z;
            double xs = (((double)u - K[2]) * zs) / K[0];
            double ys = (((double)v - K[3]) * zs) / K[1];
            xyz_m[3 * np_ + 0] = xs / 1000.;
            xyz_m[3 * np_ + 1] = ys / 1000.;
            xyz_m[3 * np_ + 2] = zs / 1000.;
            np_++;
        }
    *n_points = np_;
    *n_radii = nr;
}

/* The using_ckpts=False seam, AccumulatorSpace.py:612-619: radial_est = where(est<=rmax, est, 0);
 * sem = (radial_est != 0); depth *= sem; xyz from depth.nonzero(); radial_list = radial_est[depth.nonzero()]. */
void orc_seam_radial_only(const float *radial_est, const uint16_t *depth, int H, int W, const double *K,
                          double rmax_dm, double *xyz_m, float *radial_list, long *n_points)
{
    long n = 0;
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            long p = (long)v * W + u;
            float r = ((double)radial_est[p] <= rmax_dm) ? radial_est[p] : 0.f;
            if (r == 0.f || depth[p] == 0) continue;
            double zs = (double)depth[p];
            xyz_m[3 * n + 0] = ((((double)u - K[2]) * zs) / K[0]) / 1000.;
            xyz_m[3 * n + 1] = ((((double)v - K[3]) * zs) / K[1]) / 1000.;
            xyz_m[3 * n + 2] = zs / 1000.;
            radial_list[n] = r;
            n++;
        }
    *n_points = n;
}

/* The Occlusion-LINEMOD seam with pre-computed radial maps, AccumulatorSpace.py:849-851 and :858-864:
 *     radial_out = where(radial_out <= max_radii_dm[k], radial_out, 0);  sem_out = where(radial_out > 0, 1, 0)
 *     depth_map *= sem_out;  pixel_coor = where(sem_out == 1);  radial_list = radial_out[pixel_coor]
 *     xyz_mm = rgbd_to_point_cloud(linemod_K, depth_map);  xyz = xyz_mm / 1000
 * -- radii over the mask, points over mask & depth != 0 (lists may differ in length, as in the LINEMOD seam).
 * Returns 0 when radial_out.max() == 0 (the frame's keypoint is skipped, :858). */
int orc_seam_lmo_radial(const float *radial_est, const uint16_t *depth, int H, int W, const double *K, double rmax_dm,
                        double *xyz_m, float *radial_list, long *n_points, long *n_radii)
{
    long np_ = 0, nr = 0;
    float mx = 0.f;
    int any = 0;
    for (long p = 0; p < (long)H * W; p++) {
        float r = ((double)radial_est[p] <= rmax_dm) ? radial_est[p] : 0.f;
        if (!any || r > mx) { mx = r; any = 1; }
    }
    *n_points = 0;
    *n_radii = 0;
    if (!(mx != 0.f)) return 0;
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            long p = (long)v * W + u;
            float r = ((double)radial_est[p] <= rmax_dm) ? radial_est[p] : 0.f;
            if (!(r > 0.f)) continue;
            radial_list[nr++] = r;
            if (depth[p] == 0) continue;
            double zs = (double)depth[p];
            xyz_m[3 * np_ + 0] = ((((double)u - K[2]) * zs) / K[0]) / 1000.;
            xyz_m[3 * np_ + 1] = ((((double)v - K[3]) * zs) / K[1]) / 1000.;
            xyz_m[3 * np_ + 2] = zs / 1000.;
            np_++;
        }
    *n_points = np_;
    *n_radii = nr;
    return 1;
}

/* The YCB-Video seam, AccumulatorSpace.py:1050-1057:
 *     sem_out = where(sem_out > 0.8, 1, 0);  depth_map = depth_map1 * sem_out / factor_depth
 *     radial_list = radial_out[where(sem_out == 1)];  xyz = rgbd_to_point_cloud(intrinsic_matrix, depth_map)
 * -- no radius bound, the frame's own intrinsics, xyz in the unit of depth / factor_depth (metres), NOT divided by 1000. */
void orc_seam_ycb(const float *sem, const float *radial, const uint16_t *depth, int H, int W, const double *K, double seg_thr,
                  double factor_depth, double *xyz, float *radial_list, long *n_points, long *n_radii)
{
    long np_ = 0, nr = 0;
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            long p = (long)v * W + u;
            if (!(sem[p] > (float)seg_thr)) continue;
            radial_list[nr++] = radial[p];
            if (depth[p] == 0) continue;
            double zs = (double)depth[p] / factor_depth;   /* int64 product / uint16 (1,1) array -> float64 */
            xyz[3 * np_ + 0] = (((double)u - K[2]) * zs) / K[0];
            xyz[3 * np_ + 1] = (((double)v - K[3]) * zs) / K[1];
            xyz[3 * np_ + 2] = zs;
            np_++;
        }
    *n_points = np_;
    *n_radii = nr;
}

/* Accumulator_3D up to the call of fast_for, AccumulatorSpace.py:374-401.
 * xyz_m [n,3] metres (not modified); radial: float32 or float64 [m] decimetres (m >= n).
 * Outputs: points_vox [n,3] (recentred, shifted, voxel units), radius_vox[m] as double
 * (the float32 case is computed in float32 and widened exactly), info. */
int orc_acc_prepare(const double *xyz_m, long n, const void *radial, int radial_is_f64, long m, double acc_unit,
                    double *points_vox, double *radius_vox, orc_grid_info *info)
{
    memset(info, 0, sizeof(*info));
    if (n <= 0 || m <= 0) {
        info->status = ORC_E_EMPTY;
        return ORC_E_EMPTY;
    }
    for (long i = 0; i < 3 * n; i++) points_vox[i] = (xyz_m[i] * 1000.) / acc_unit;        /* :376 */
    for (int c = 0; c < 3; c++) info->mean[c] = orc_mean_strided(points_vox + c, n, 3);     /* :381-383 */
    for (long i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) points_vox[3 * i + c] -= info->mean[c];               /* :384-386 */
    double rmax;
    if (radial_is_f64) {                                                                   /* :388 */
        const double *r = (const double *)radial;
        for (long i = 0; i < m; i++) radius_vox[i] = (r[i] * 100.) / acc_unit;
    } else {
        const float *r = (const float *)radial;
        float unit = (float)acc_unit; /* python int 5 is a weak scalar: float32 arithmetic */
        for (long i = 0; i < m; i++) radius_vox[i] = (double)((r[i] * 100.f) / unit);
    }
    rmax = radius_vox[0];
    for (long i = 1; i < m; i++) if (radius_vox[i] > rmax) rmax = radius_vox[i];          /* :392 */
    double pmin = points_vox[0], pmax;
    for (long i = 1; i < 3 * n; i++) if (points_vox[i] < pmin) pmin = points_vox[i];      /* :390 */
    int32_t zb = (int32_t)(pmin - rmax) + 1;                                               /* :394 */
    if (zb < 0)
        for (long i = 0; i < 3 * n; i++) points_vox[i] -= (double)zb;                      /* :396-397 */
    pmax = points_vox[0];
    for (long i = 1; i < 3 * n; i++) if (points_vox[i] > pmax) pmax = points_vox[i];
    info->radius_max = rmax;
    info->zero_boundary = zb;
    info->length = (int32_t)pmax;                                                          /* :399 */
    info->dim = info->length + (int32_t)rmax;                                              /* :401 */
    if (info->dim <= 0) {
        info->status = ORC_E_BADGRID;
        return ORC_E_BADGRID;
    }
    return ORC_OK;
}

/* round-half-even of the voxel radius then int(), AccumulatorSpace.py:332 */
static inline long radius_int(double rv) { return (long)rint(rv); }

/* fast_for, AccumulatorSpace.py:325-341: every point against every voxel.
 * votes: uint32 [D,D,D] (C order), must be zeroed by the caller.  The reference parallelises the
 * point loop with an unsynchronised += (a data race); the race-free result (what the
 * sequential loop gives) is reproduced here by parallelising the i-slabs instead.  The amount
 * of arithmetic (n*D^3 distance tests) is the reference's. */
void orc_fast_for(const double *points_vox, const double *radius_vox, long n, int D, uint32_t *votes, int threads)
{
    const double factor = sqrt(3.0) / 4.0; /* (3**0.5)/4, :328 */
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < D; i++) {
        for (long p = 0; p < n; p++) {
            const double x = points_vox[3 * p], y = points_vox[3 * p + 1], z = points_vox[3 * p + 2];
            const double radius = (double)radius_int(radius_vox[p]);
            const double dx = (double)i - x, dx2 = dx * dx;
            for (int j = 0; j < D; j++) {
                const double dy = (double)j - y, sxy = dx2 + dy * dy;
                uint32_t *row = votes + ((long)i * D + j) * D;
                for (int k = 0; k < D; k++) {
                    const double dz = (double)k - z;
                    const double distance = sqrt(sxy + dz * dz);
                    const double t = radius - distance;
                    if (t < factor && t > 0) row[k] += 1;
                }
            }
        }
    }
}

/* Same votes as orc_fast_for (proved equal on the golden/small cases by tests/test_oracle.py),
 * but only the voxels near each shell are tested -- used to check the GPU at sizes where the
 * brute force would take minutes.  The predicate applied to each candidate is identical. */
/* votes of all points into the i-th plane of the grid (plane: uint32 [D][D], indexed [j][k]) */
static void sparse_plane(const double *points_vox, const double *radius_vox, long n, int D, int i, uint32_t *plane)
{
    const double factor = sqrt(3.0) / 4.0;
    for (long p = 0; p < n; p++) {
        const double x = points_vox[3 * p], y = points_vox[3 * p + 1], z = points_vox[3 * p + 2];
        const long ri = radius_int(radius_vox[p]);
        if (ri <= 0) continue;
        const double radius = (double)ri;
        const double dx = (double)i - x, dx2 = dx * dx;
        if (dx2 >= radius * radius + 1.0) continue;
        int j0 = (int)floor(y - radius - 1.0), j1 = (int)ceil(y + radius + 1.0);
        if (j0 < 0) j0 = 0;
        if (j1 > D - 1) j1 = D - 1;
        for (int j = j0; j <= j1; j++) {
            const double dy = (double)j - y, sxy = dx2 + dy * dy;
            const double hi = radius * radius - sxy;
            if (hi < -1.0) continue;
            const double inner = radius - factor;
            const double lo = inner * inner - sxy;
            const double zb_ = sqrt(hi > 0 ? hi : 0) + 1.0, za_ = lo > 0 ? sqrt(lo) - 1.0 : -1.0;
            uint32_t *row = plane + (long)j * D;
            /* candidate k ranges: [z - zb_, z - za_] and [z + za_, z + zb_] (merged when they touch) */
            int a0 = (int)floor(z - zb_), a1 = (int)ceil(z - za_);
            int b0 = (int)floor(z + za_), b1 = (int)ceil(z + zb_);
            if (b0 <= a1) { a1 = b1; b0 = b1 + 1; }
            for (int pass = 0; pass < 2; pass++) {
                int k0 = pass ? b0 : a0, k1 = pass ? b1 : a1;
                if (k0 < 0) k0 = 0;
                if (k1 > D - 1) k1 = D - 1;
                for (int k = k0; k <= k1; k++) {
                    const double dz = (double)k - z;
                    const double distance = sqrt(sxy + dz * dz);
                    const double t = radius - distance;
                    if (t < factor && t > 0) row[k] += 1;
                }
            }
        }
    }
}

void orc_fast_for_sparse(const double *points_vox, const double *radius_vox, long n, int D, uint32_t *votes, int threads)
{
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < D; i++) sparse_plane(points_vox, radius_vox, n, D, i, votes + (long)i * D * D);
}

/* VoteMap_3D.max() and argwhere(VoteMap_3D == max) (AccumulatorSpace.py:406) WITHOUT holding the grid: for grids too
 * large to allocate (1 mm voxels: D ~ 1000, 4 GB as uint32).  Every i-plane is voted (the same candidate walk and the
 * same predicate as orc_fast_for_sparse) into a thread-private D x D buffer and scanned for its maximum, the number of
 * voxels attaining it and the first of them; the planes are then combined in i order.  Returns the number of maxima;
 * *first_flat = flat C-order index of the first one.  Optionally the counts of `n_probe` given voxels (flat indices)
 * are written to probe_votes.  Checked against orc_fast_for_sparse + orc_argmax_all in tests/test_oracle_golden.py. */
long orc_peak_planes(const double *points_vox, const double *radius_vox, long n, int D, uint32_t *peak, int64_t *first_flat,
                     const int64_t *probe_flat, long n_probe, uint32_t *probe_votes, int threads)
{
    uint32_t *pmax = (uint32_t *)calloc((size_t)D, sizeof(uint32_t));
    long *pcnt = (long *)calloc((size_t)D, sizeof(long));
    long *pfirst = (long *)calloc((size_t)D, sizeof(long));
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        uint32_t *plane = (uint32_t *)malloc((size_t)D * D * sizeof(uint32_t));
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < D; i++) {
            memset(plane, 0, (size_t)D * D * sizeof(uint32_t));
            sparse_plane(points_vox, radius_vox, n, D, i, plane);
            uint32_t mx = 0;
            long cnt = 0, first = 0;
            const long cells = (long)D * D;
            for (long t = 0; t < cells; t++) if (plane[t] > mx) mx = plane[t];
            for (long t = 0; t < cells; t++)
                if (plane[t] == mx) { if (!cnt) first = t; cnt++; }
            pmax[i] = mx; pcnt[i] = cnt; pfirst[i] = first;
            for (long q = 0; q < n_probe; q++)
                if (probe_flat[q] / cells == i) probe_votes[q] = plane[probe_flat[q] % cells];
        }
        free(plane);
    }
    uint32_t mx = 0;
    for (int i = 0; i < D; i++) if (pmax[i] > mx) mx = pmax[i];
    long total = 0;
    int have_first = 0;
    for (int i = 0; i < D; i++)
        if (pmax[i] == mx) {
            if (!have_first) { *first_flat = (int64_t)i * D * D + pfirst[i]; have_first = 1; }
            total += pcnt[i];
        }
    *peak = mx;
    free(pmax); free(pcnt); free(pfirst);
    return total;
}

/* argwhere(V == V.max()) in C order, AccumulatorSpace.py:406.  Returns the number of maxima;
 * writes up to cap flat indices (ascending). */
long orc_argmax_all(const uint32_t *votes, int D, uint32_t *peak, int64_t *flat_idx, long cap)
{
    long total = (long)D * D * D, cnt = 0;
    uint32_t mx = 0;
    for (long t = 0; t < total; t++) if (votes[t] > mx) mx = votes[t];
    for (long t = 0; t < total; t++)
        if (votes[t] == mx) {
            if (cnt < cap) flat_idx[cnt] = t;
            cnt++;
        }
    *peak = mx;
    return cnt;
}

/* Row 0 of the returned centre in mm, AccumulatorSpace.py:408-415 (other rows stay voxel indices
 * shifted by zero_boundary). */
void orc_center_row0(int64_t flat, const orc_grid_info *info, double acc_unit, double *center_mm)
{
    long D = info->dim;
    double c[3] = {(double)(flat / (D * D)), (double)((flat / D) % D), (double)(flat % D)};
    for (int a = 0; a < 3; a++) {
        if (info->zero_boundary < 0) c[a] = c[a] + (double)info->zero_boundary;
        center_mm[a] = (c[a] + info->mean[a] + 0.5) * acc_unit;
    }
}

/* ------------------------------------------------------------------------------------------
 * RGB-only variant (SURVEY f-5): the depth of every masked pixel is unknown, so each pixel is back-projected at
 * every millimetre of a depth range and all hypotheses vote.
 * ------------------------------------------------------------------------------------------ */

/* rgbd_to_point_cloud_no_depth, AccumulatorSpace.py:100-120.  depth: float64 [H,W] in METRES (the reference converts
 * with *1000 for the hypothesis count only).  pts [Z][N][3] with Z = int(zmax*1000)+1-int(zmin*1000), hypothesis i at
 * depth zmin + i*0.001 for every nonzero pixel (row-major).  Returns Z (0: nothing / buffer too small), *n_out = N. */
long orc_backproject_no_depth(const double *depth, int H, int W, const double *K, double *pts, long cap_points, long *n_out)
{
    long n = 0;
    double zmin = 0, zmax = 0;
    for (long p = 0; p < (long)H * W; p++)
        if (depth[p] != 0.0) {
            if (!n || depth[p] < zmin) zmin = depth[p];
            if (!n || depth[p] > zmax) zmax = depth[p];
            n++;
        }
    *n_out = n;
    if (!n) return 0;
    const long Z = (long)(zmax * 1000.) + 1 - (long)(zmin * 1000.);
    if (Z * n > cap_points) return 0;
    for (long i = 0; i < Z; i++) {
        const double z = zmin + (double)i * 0.001;
        long q = 0;
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                if (depth[(long)v * W + u] == 0.0) continue;
                double *o = pts + 3 * (i * n + q);
                o[0] = (((double)u - K[2]) * z) / K[0];
                o[1] = (((double)v - K[3]) * z) / K[1];
                o[2] = z;
                q++;
            }
    }
    return Z;
}

/* Accumulator_3D_no_depth up to the call of fast_for_no_depth, AccumulatorSpace.py:436-463.  xyz_m [Z][N][3] metres.
 *   :438      xyz_mm = xyz*200
 *   :440-445  x_mean_mm = np.mean(xyz_mm[:,0]) -- with a [Z,N,3] array that is the mean over ALL hypotheses and ALL three
 *             coordinates of POINT 0 (likewise points 1, 2), and the subtraction shifts those three points only.
 *             (np.mean of the [Z,3] strided view = the chunked pairwise sum of its C-order elements, orc_mean_strided.)
 *   :447      radial_list_mm = radial_list*20: NOT rounded later (fast_for_no_depth uses it as is); float32 stays float32
 *   :453-460  zero_boundary / shift as in Accumulator_3D; length = int(max)+1 WITHOUT the radius margin
 * info->dim = length, info->mean = the three means.  N < 3 -> the reference raises IndexError. */
int orc_nd_prepare(const double *xyz_m, long Z, long N, const void *radial, int radial_is_f64, long m, double *points_vox,
                   double *radius_vox, orc_grid_info *info)
{
    memset(info, 0, sizeof(*info));
    if (N < 3 || Z < 1 || m < 1) { info->status = ORC_E_EMPTY; return ORC_E_EMPTY; }
    const long total = 3 * Z * N;
    for (long i = 0; i < total; i++) points_vox[i] = xyz_m[i] * 200.;                       /* :438 */
    double *tmp = (double *)malloc(sizeof(double) * 3 * (size_t)Z);
    for (int a = 0; a < 3; a++) {                                                            /* :440-442 */
        for (long z = 0; z < Z; z++)
            for (int c = 0; c < 3; c++) tmp[3 * z + c] = points_vox[3 * (z * N + a) + c];
        info->mean[a] = orc_mean_strided(tmp, 3 * Z, 1);
    }
    free(tmp);
    for (int a = 0; a < 3; a++)                                                              /* :443-445 */
        for (long z = 0; z < Z; z++)
            for (int c = 0; c < 3; c++) points_vox[3 * (z * N + a) + c] -= info->mean[a];
    if (radial_is_f64) {
        const double *r = (const double *)radial;
        for (long i = 0; i < m; i++) radius_vox[i] = r[i] * 20.;
    } else {
        const float *r = (const float *)radial;
        for (long i = 0; i < m; i++) radius_vox[i] = (double)(r[i] * 20.f);
    }
    double rmax = radius_vox[0];
    for (long i = 1; i < m; i++) if (radius_vox[i] > rmax) rmax = radius_vox[i];
    double pmin = points_vox[0], pmax;
    for (long i = 1; i < total; i++) if (points_vox[i] < pmin) pmin = points_vox[i];
    int32_t zb = (int32_t)(pmin - rmax) + 1;                                                 /* :453 */
    if (zb < 0)
        for (long i = 0; i < total; i++) points_vox[i] -= (double)zb;                        /* :456-457 */
    pmax = points_vox[0];
    for (long i = 1; i < total; i++) if (points_vox[i] > pmax) pmax = points_vox[i];
    info->radius_max = rmax;
    info->zero_boundary = zb;
    info->length = (int32_t)pmax + 1;                                                        /* :459 */
    info->dim = info->length;
    if (info->dim <= 0) { info->status = ORC_E_BADGRID; return ORC_E_BADGRID; }
    return ORC_OK;
}

/* fast_for_no_depth, AccumulatorSpace.py:421-434: every hypothesis of every point against every voxel; the radius of
 * point `count` is radial_list_mm[count], a real number (no rounding), shared by all its hypotheses. */
void orc_fast_for_nd(const double *points_vox, const double *radius_vox, long Z, long N, int D, uint32_t *votes, int threads)
{
    const double factor = sqrt(3.0) / 4.0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < D; i++)
        for (long z = 0; z < Z; z++)
            for (long p = 0; p < N; p++) {
                const double *q = points_vox + 3 * (z * N + p);
                const double radius = radius_vox[p];
                const double dx = (double)i - q[0], dx2 = dx * dx;
                for (int j = 0; j < D; j++) {
                    const double dy = (double)j - q[1], sxy = dx2 + dy * dy;
                    uint32_t *row = votes + ((long)i * D + j) * D;
                    for (int k = 0; k < D; k++) {
                        const double dz = (double)k - q[2];
                        const double distance = sqrt(sxy + dz * dz);
                        const double t = radius - distance;
                        if (t < factor && t > 0) row[k] += 1;
                    }
                }
            }
}

/* `center` of Accumulator_3D_no_depth, AccumulatorSpace.py:468-479: int64 rows; row 0 gets the float means added and is
 * truncated back to int64 by the item assignment; every row is multiplied by 5. */
void orc_nd_center(const int64_t *flat_idx, long n_rows, const orc_grid_info *info, int64_t *center)
{
    const long D = info->dim;
    for (long r = 0; r < n_rows; r++) {
        int64_t c[3] = {flat_idx[r] / (D * D), (flat_idx[r] / D) % D, flat_idx[r] % D};
        for (int a = 0; a < 3; a++) {
            if (info->zero_boundary < 0) c[a] += info->zero_boundary;
            if (r == 0) c[a] = (int64_t)((double)c[a] + info->mean[a]);   /* float into an int64 element: C cast, toward zero */
            center[3 * r + a] = c[a] * 5;
        }
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Horn absolute orientation, util/horn.py:75-181 (lmshorn) with its Jacobi eigen-solver (:12-72),
 * called with the 3 model / estimated keypoints at AccumulatorSpace.py:659-661.
 * 1-based 5x5 work arrays like the reference (a Numerical-Recipes port); same operation order.
 * P1, P2: [n,3] (not modified: the reference subtracts and re-adds the centroids in place).
 * A: [4,4] row-major homogeneous transform with P2 ~ R P1 + T.
 * ------------------------------------------------------------------------------------------ */
static void horn_rotate(double a[5][5], int i, int j, int k, int l, double s, double tau)
{
    double g = a[i][j], h = a[k][l];
    a[i][j] = g - s * (h + g * tau);
    a[k][l] = h + s * (g - h * tau);
}

static void horn_jacobi(double a[5][5], int n, double d[5], double v[5][5])
{
    double b[5] = {0}, z[5] = {0};
    for (int ip = 1; ip <= n; ip++) {
        for (int iq = 1; iq <= n; iq++) v[ip][iq] = 0.0;
        v[ip][ip] = 1.0;
    }
    for (int ip = 1; ip <= n; ip++) { b[ip] = d[ip] = a[ip][ip]; z[ip] = 0.0; }
    for (int i = 1; i <= 50; i++) {
        double sm = 0.0;
        for (int ip = 1; ip < n; ip++)            /* rows 1..n-1, ALL columns (util/horn.py:27-29) */
            for (int iq = 1; iq <= n; iq++) sm += fabs(a[ip][iq]);
        if (sm == 0.0) return;
        double tresh = i < 4 ? 0.2 * sm / (n * n) : 0.0;
        for (int ip = 1; ip < n; ip++)
            for (int iq = ip + 1; iq <= n; iq++) {
                double g = 100.0 * fabs(a[ip][iq]);
                if (i > 4 && fabs(d[ip]) + g == fabs(d[ip]) && fabs(d[iq]) + g == fabs(d[iq])) {
                    a[ip][iq] = 0.0;
                } else if (fabs(a[ip][iq]) > tresh) {
                    double h = d[iq] - d[ip], t;
                    if (fabs(h) + g == fabs(h)) {
                        t = a[ip][iq] / h;
                    } else {
                        double theta = 0.5 * h / a[ip][iq];
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    double c = 1.0 / sqrt(1 + t * t), s = t * c, tau = s / (1.0 + c);
                    h = t * a[ip][iq];
                    z[ip] -= h; z[iq] += h; d[ip] -= h; d[iq] += h;
                    a[ip][iq] = 0.0;
                    for (int j = 1; j < ip; j++) horn_rotate(a, j, ip, j, iq, s, tau);
                    for (int j = ip + 1; j < iq; j++) horn_rotate(a, ip, j, j, iq, s, tau);
                    for (int j = iq + 1; j <= n; j++) horn_rotate(a, ip, j, iq, j, s, tau);
                    for (int j = 1; j <= n; j++) horn_rotate(v, j, ip, j, iq, s, tau);
                }
            }
        for (int ip = 1; ip <= n; ip++) { b[ip] += z[ip]; d[ip] = b[ip]; z[ip] = 0.0; }
    }
}

void orc_horn(const double *P1in, const double *P2in, int n, double *A)
{
    double C1[3] = {0, 0, 0}, C2[3] = {0, 0, 0};
    double P1[64][3], P2[64][3];
    if (n > 64) n = 64;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) { C1[j] += P1in[3 * i + j]; C2[j] += P2in[3 * i + j]; }
    for (int j = 0; j < 3; j++) { C1[j] /= n; C2[j] /= n; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) { P1[i][j] = P1in[3 * i + j] - C1[j]; P2[i][j] = P2in[3 * i + j] - C2[j]; }
    double Sxx = 0, Sxy = 0, Sxz = 0, Syx = 0, Syy = 0, Syz = 0, Szx = 0, Szy = 0, Szz = 0;
    for (int i = 0; i < n; i++) {
        Sxx += P1[i][0] * P2[i][0]; Sxy += P1[i][0] * P2[i][1]; Sxz += P1[i][0] * P2[i][2];
        Syx += P1[i][1] * P2[i][0]; Syy += P1[i][1] * P2[i][1]; Syz += P1[i][1] * P2[i][2];
        Szx += P1[i][2] * P2[i][0]; Szy += P1[i][2] * P2[i][1]; Szz += P1[i][2] * P2[i][2];
    }
    double N[5][5] = {{0}}, D[5] = {0}, V[5][5] = {{0}};
    N[1][1] = Sxx + Syy + Szz; N[1][2] = Syz - Szy; N[1][3] = Szx - Sxz; N[1][4] = Sxy - Syx;
    N[2][1] = Syz - Szy; N[2][2] = Sxx - Syy - Szz; N[2][3] = Sxy + Syx; N[2][4] = Szx + Sxz;
    N[3][1] = Szx - Sxz; N[3][2] = Sxy + Syx; N[3][3] = -Sxx + Syy - Szz; N[3][4] = Syz + Szy;
    N[4][1] = Sxy - Syx; N[4][2] = Szx + Sxz; N[4][3] = Syz + Szy; N[4][4] = -Sxx - Syy + Szz;
    horn_jacobi(N, 4, D, V);
    int mi = 1;
    double mv = D[1];
    for (int i = 2; i < 5; i++) if (D[i] > mv) { mi = i; mv = D[i]; }
    double q0 = V[1][mi], q1 = V[2][mi], q2 = V[3][mi], q3 = V[4][mi];
    double R[3][3];
    R[0][0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; R[0][1] = 2 * (q1 * q2 - q0 * q3); R[0][2] = 2 * (q1 * q3 + q0 * q2);
    R[1][0] = 2 * (q1 * q2 + q0 * q3); R[1][1] = q0 * q0 + q2 * q2 - q1 * q1 - q3 * q3; R[1][2] = 2 * (q2 * q3 - q0 * q1);
    R[2][0] = 2 * (q1 * q3 - q0 * q2); R[2][1] = 2 * (q2 * q3 + q0 * q1); R[2][2] = q0 * q0 + q3 * q3 - q1 * q1 - q2 * q2;
    for (int i = 0; i < 3; i++) {
        A[4 * i + 3] = C2[i] - (R[i][0] * C1[0] + R[i][1] * C1[1] + R[i][2] * C1[2]);
        for (int j = 0; j < 3; j++) A[4 * i + j] = R[i][j];
        A[12 + i] = 0.0;
    }
    A[15] = 1.0;
}
