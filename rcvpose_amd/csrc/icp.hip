// icp.hip -- nearest-neighbour distances (ADD / ADD-S) and point-to-point ICP on the device.
//
// Reference: the reference calls Open3D 0.14.1 for both (AccumulatorSpace.py:689-736):
//   PointCloud.compute_point_cloud_distance(other)          -> rcv_nn_distance
//   pipelines.registration.registration_icp(source, target, max_correspondence_distance, init,
//       TransformationEstimationPointToPoint(), ICPConvergenceCriteria(...))  -> rcv_icp_point_to_point
// Open3D is a third-party dependency that is absent here (rcvpose.yml:176 pins open3d==0.14.1), so this file restates
// its published algorithm (RegistrationICP + GetRegistrationResultAndCorrespondences, Registration.cpp of 0.14):
//   result(T): every source point T*p takes its nearest target point within max_correspondence_distance;
//              fitness = #correspondences / #source, inlier_rmse = sqrt(sum d^2 / #correspondences)
//   loop (<= max_iteration): update = least-squares rigid transform of the correspondences (Umeyama without scale),
//              T = update * T, re-evaluate; stop when |d fitness| < relative_fitness and |d rmse| < relative_rmse.
// Parity is unpinned against Open3D itself (the tests compare with a numpy/scipy restatement of the same algorithm).
// Nearest neighbours are exact (float64 brute force, target streamed through LDS; ties -> lowest index), the rigid
// update is Horn's quaternion solution of the same least-squares problem as Umeyama's SVD.
#include <math.h>
#include <stdlib.h>
#include <cmath>
#include <string.h>

#include <vector>

#include <mutex>

#include "common.hpp"

namespace rcv {
namespace {
static_assert(WAVE == 64, "the butterfly reductions below (__shfl_xor from 32 down) assume 64-lane wavefronts (gfx950)");

constexpr int NN_THREADS = 256, NN_TILE = 1024, RED_BLOCKS = 256;

// nearest b for every a: squared distance and index (first index among equal distances)
__global__ __launch_bounds__(NN_THREADS) void k_nn(const double *a, int n_a, const double *b, int n_b, double *d2_out, int *idx_out)
{
    __shared__ double sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
    const int i = blockIdx.x * NN_THREADS + threadIdx.x;
    const int ii = i < n_a ? i : n_a - 1;
    const double px = a[3 * ii], py = a[3 * ii + 1], pz = a[3 * ii + 2];
    double best = INFINITY;
    int bi = -1;
    for (int base = 0; base < n_b; base += NN_TILE) {
        __syncthreads();
        for (int t = threadIdx.x; t < NN_TILE; t += NN_THREADS) {
            const int j = base + t;
            if (j < n_b) { sx[t] = b[3 * j]; sy[t] = b[3 * j + 1]; sz[t] = b[3 * j + 2]; }
        }
        __syncthreads();
        const int m = min(NN_TILE, n_b - base);
#pragma unroll 4
        for (int t = 0; t < m; t++) {
            const double dx = px - sx[t], dy = py - sy[t], dz = pz - sz[t];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; bi = base + t; }
        }
    }
    if (i < n_a) {
        d2_out[i] = best;
        if (idx_out) idx_out[i] = bi;
    }
}

// The same search with the target cloud cut into gridDim.y slices (r03: 3 000 model points against 15 000 scene points kept
// 12 workgroups busy for 0.2 ms per ICP iteration): slice s scans targets [s * per, (s + 1) * per) and leaves its best
// (squared distance, index) in part_*[s][i]; k_icp_merge_sums folds the slices in order with the same strict `<`, which is the first
// index among equal distances again -- identical results.
constexpr int NN_SLICES = 16;
// T (row-major 4x4, may be null: identity) is applied to the source point on the fly; slice 0 also leaves the transformed points
// in `cur` for the reductions that follow (r03: one launch less per ICP iteration).
__global__ __launch_bounds__(NN_THREADS) void k_nn_sliced(const double *a, int n_a, const double *T, double *cur, const double *b, int n_b,
                                                          double *part_d2, int *part_idx, const int *done)
{
    if (done && *done) return;
    __shared__ double sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
    const int i = blockIdx.x * NN_THREADS + threadIdx.x;
    const int ii = i < n_a ? i : n_a - 1;
    const int per = (n_b + NN_SLICES - 1) / NN_SLICES;
    const int lo = blockIdx.y * per, hi = min(n_b, lo + per);
    double px = a[3 * ii], py = a[3 * ii + 1], pz = a[3 * ii + 2];
    if (T) {
        const double x = px, y = py, z = pz;
        px = T[0] * x + T[1] * y + T[2] * z + T[3];
        py = T[4] * x + T[5] * y + T[6] * z + T[7];
        pz = T[8] * x + T[9] * y + T[10] * z + T[11];
        if (blockIdx.y == 0 && i < n_a) { cur[3 * i] = px; cur[3 * i + 1] = py; cur[3 * i + 2] = pz; }
    }
    double best = INFINITY;
    int bi = -1;
    for (int base = lo; base < hi; base += NN_TILE) {
        __syncthreads();
        for (int t = threadIdx.x; t < NN_TILE; t += NN_THREADS) {
            const int j = base + t;
            if (j < hi) { sx[t] = b[3 * j]; sy[t] = b[3 * j + 1]; sz[t] = b[3 * j + 2]; }
        }
        __syncthreads();
        const int m = min(NN_TILE, hi - base);
#pragma unroll 4
        for (int t = 0; t < m; t++) {
            const double dx = px - sx[t], dy = py - sy[t], dz = pz - sz[t];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; bi = base + t; }
        }
    }
    if (i < n_a) {
        part_d2[(size_t)blockIdx.y * n_a + i] = best;
        part_idx[(size_t)blockIdx.y * n_a + i] = bi;
    }
}

__global__ __launch_bounds__(256) void k_sqrt(const double *d2, int n, double *d)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = sqrt(d2[i]);
}

__device__ __forceinline__ double block_sum(double v, double *s)
{
    for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return s[0] + s[1] + s[2] + s[3];
}

// Fold the slices of the nearest-neighbour search (in order, strict `<`: the first index among equal distances) and, in the same
// pass, the sums over the correspondences (source i <-> target idx[i], d2[i] <= max_d2) that evaluation and update need:
//   part[block][0] = count, [1..3] = sum (p - a), [4..6] = sum (q - b), [7] = sum d2, [8..16] = sum (p - a)(q - b)^T
// with a, b = ref[0..2], ref[3..5]: the correspondence means of the previous evaluation (zeros at first), so that the centred
// covariance S = sum (p - a)(q - b)^T - n (pm - a)(qm - b)^T comes out of ONE pass without cancellation.  Fixed block / thread
// assignment: reproducible.
constexpr int ICP_NSUM = 17;
template <int SL>
__global__ __launch_bounds__(256) void k_icp_merge_sums(const double *part_d2, const int *part_idx, const double *cur, const double *tgt, int n,
                                                        double max_d2, const double *ref, double *d2_out, int *idx_out, double *part,
                                                        const int *done)
{
    if (done && *done) return;
    __shared__ double s[4];
    const double a[3] = {ref[0], ref[1], ref[2]}, b[3] = {ref[3], ref[4], ref[5]};
    double acc[ICP_NSUM];
#pragma unroll
    for (int k = 0; k < ICP_NSUM; k++) acc[k] = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        double best = INFINITY;
        int bi = -1;
        for (int sl = 0; sl < SL; sl++) {
            const double d2 = part_d2[(size_t)sl * n + i];
            if (d2 < best) { best = d2; bi = part_idx[(size_t)sl * n + i]; }
        }
        if (SL > 1) {  // (SL == 1: the grid search wrote d2_out / idx_out itself, part_* alias them)
            d2_out[i] = best;
            idx_out[i] = bi;
        }
        if (bi >= 0 && best <= max_d2) {
            const double p[3] = {cur[3 * i] - a[0], cur[3 * i + 1] - a[1], cur[3 * i + 2] - a[2]};
            const double q[3] = {tgt[3 * bi] - b[0], tgt[3 * bi + 1] - b[1], tgt[3 * bi + 2] - b[2]};
            acc[0] += 1.0;
            acc[1] += p[0]; acc[2] += p[1]; acc[3] += p[2];
            acc[4] += q[0]; acc[5] += q[1]; acc[6] += q[2];
            acc[7] += best;
#pragma unroll
            for (int u = 0; u < 3; u++)
#pragma unroll
                for (int v = 0; v < 3; v++) acc[8 + u * 3 + v] += p[u] * q[v];
        }
    }
#pragma unroll
    for (int k = 0; k < ICP_NSUM; k++) {
        const double v = block_sum(acc[k], s);
        if (threadIdx.x == 0) part[blockIdx.x * ICP_NSUM + k] = v;
    }
}

// ---- nearest neighbour within the correspondence distance through a hashed uniform grid (r03).  The ICP only uses neighbours
// within max_correspondence_distance h: with the target points binned into cells of edge h, the nearest target within h of a
// point lies in the 27 cells around the point's own.  Cells are hashed into M = 2^k >= 2 n_tgt buckets (collisions only add
// candidates); every candidate's squared distance is the brute-force kernel's expression, the minimum is taken with the same
// rule (smaller distance, then smaller target index), so correspondences -- and everything computed from them -- are those of
// the exhaustive search.  Points without a target within h report whatever the 27 cells held (never a correspondence).
__device__ __forceinline__ unsigned grid_hash(long long ix, long long iy, long long iz, unsigned mask)
{
    unsigned long long h = (unsigned long long)ix * 73856093ULL ^ (unsigned long long)iy * 19349663ULL ^ (unsigned long long)iz * 83492791ULL;
    h ^= h >> 29;
    return (unsigned)h & mask;
}
__global__ __launch_bounds__(256) void k_grid_count(const double *b, int n_b, double inv_h, unsigned mask, int *counts, int *max_count)
{
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_b; j += gridDim.x * 256) {
        const unsigned h = grid_hash((long long)floor(b[3 * j] * inv_h), (long long)floor(b[3 * j + 1] * inv_h), (long long)floor(b[3 * j + 2] * inv_h), mask);
        const int c = atomicAdd(&counts[h], 1) + 1;
        atomicMax(max_count, c);
    }
}
// exclusive scan of counts[M] into starts[M + 1], one workgroup (M is a power of two >= 1024)
__global__ __launch_bounds__(1024) void k_grid_scan(const int *counts, int M, int *starts, unsigned long long *sumsq)
{
    __shared__ int s[1024];
    const int per = M / 1024, t = threadIdx.x;
    int sum = 0;
    unsigned long long sq = 0;
    for (int k = 0; k < per; k++) {
        const int c = counts[t * per + k];
        sum += c;
        sq += (unsigned long long)c * (unsigned long long)c;
    }
    atomicAdd(sumsq, sq);  // sum of squared bucket counts / n = the occupancy a target point sees in its own bucket
    s[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? s[t - off] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    int run = s[t] - sum;
    for (int k = 0; k < per; k++) {
        starts[t * per + k] = run;
        run += counts[t * per + k];
    }
    if (t == 1023) starts[M] = run;
}
__global__ __launch_bounds__(256) void k_grid_fill(const double *b, int n_b, double inv_h, unsigned mask, const int *starts, int *cursor,
                                                   double *sorted, int *sorted_idx)
{
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_b; j += gridDim.x * 256) {
        const double x = b[3 * j], y = b[3 * j + 1], z = b[3 * j + 2];
        const unsigned h = grid_hash((long long)floor(x * inv_h), (long long)floor(y * inv_h), (long long)floor(z * inv_h), mask);
        const int pos = starts[h] + atomicAdd(&cursor[h], 1);
        sorted[3 * pos] = x; sorted[3 * pos + 1] = y; sorted[3 * pos + 2] = z;
        sorted_idx[pos] = j;
    }
}
// 32 lanes per source point, one per neighbouring cell (27 of them work): with a thread per point the 6 000-point model gave the
// chip 94 waves that each walked 27 dependent bucket lookups -- 45 us of latency for 600 k distance evaluations.  Two cells that
// hash to one bucket make two lanes scan the same points; the minimum does not care.
__global__ __launch_bounds__(256) void k_nn_grid(const double *a, int n_a, const double *T, double *cur, const double *sorted, const int *sorted_idx,
                                                 const int *starts, double inv_h, unsigned mask, double *d2_out, int *idx_out, const int *done)
{
    if (done && *done) return;
    const int c = threadIdx.x & 31;
    int i = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = i < n_a;
    i = live ? i : n_a - 1;
    const double x = a[3 * i], y = a[3 * i + 1], z = a[3 * i + 2];
    const double px = T[0] * x + T[1] * y + T[2] * z + T[3];
    const double py = T[4] * x + T[5] * y + T[6] * z + T[7];
    const double pz = T[8] * x + T[9] * y + T[10] * z + T[11];
    double best = INFINITY;
    int bi = -1;
    if (c < 27) {
        const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
        const unsigned h = grid_hash((long long)floor(px * inv_h) + dx, (long long)floor(py * inv_h) + dy, (long long)floor(pz * inv_h) + dz, mask);
        const int e = starts[h + 1];
        for (int j = starts[h]; j < e; j++) {
            const double ex = px - sorted[3 * j], ey = py - sorted[3 * j + 1], ez = pz - sorted[3 * j + 2];
            const double d2 = ex * ex + ey * ey + ez * ez;
            const int oj = sorted_idx[j];
            if (d2 < best || (d2 == best && oj < bi)) { best = d2; bi = oj; }
        }
    }
    for (int m = 16; m; m >>= 1) {  // (xor masks below 32 stay inside the point's half of the wave)
        const double ob = __shfl_xor(best, m);
        const int oi = __shfl_xor(bi, m);
        if (ob < best || (ob == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ob; bi = oi; }
    }
    if (live && c == 0) {
        cur[3 * i] = px; cur[3 * i + 1] = py; cur[3 * i + 2] = pz;
        d2_out[i] = best;
        idx_out[i] = bi;
    }
}

// cyclic Jacobi for a symmetric 4x4 matrix: eigenvalues d, eigenvectors in the columns of v (every loop over matrix indices
// unrolled: the arrays stay in registers -- with run-time indices they lived in scratch memory and the one-lane kernel took 155 us)
__device__ __forceinline__ void jacobi4(double a[4][4], double d[4], double v[4][4])
{
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            diag += a[i][i] * a[i][i];
#pragma unroll
            for (int j = i + 1; j < 4; j++) off += a[i][j] * a[i][j];
        }
        // (relative: the off-diagonal part stalls near eps * |A| and never reaches an absolute 1e-300 -- the loop used to run all
        // 60 sweeps, 150 us on one lane; converged sweeps only rotate by angles below eps)
        if (off <= 1e-32 * diag || off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                for (int k = 0; k < 4; k++) {  // A <- A * J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - sn * akq;
                    a[k][q] = sn * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {  // A <- J^T * A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - sn * aqk;
                    a[q][k] = sn * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - sn * vkq;
                    v[k][q] = sn * vkp + c * vkq;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) d[i] = a[i][i];
}

// One wave closes an evaluation (r03: this was k_icp_finish_sums, k_icp_decide, k_icp_cov and k_icp_update -- the host used to
// read the fitness back after every iteration, 30 synchronisations of ~0.4 ms per frame, and a single lane added the partials):
//   * the per-block partials, one per lane and stride 64, then a butterfly over the wave: stats[0] = count, stats[7] = sum d2,
//     stats[8..10] / [11..13] = mean p / mean q of the correspondences (the next pass's reference points);
//   * the reference loop's bookkeeping -- stats[16..19] = previous / current (fitness, rmse); ctl = (int *)(stats + 24): done,
//     iterations, correspondences.  `first`: the evaluation of the start pose; otherwise one after an update:
//         it++; converged (|d fitness| < relative_fitness and |d rmse| < relative_rmse) or it == max or no correspondences -> done
//   * unless done: the rigid update from the centred covariance (Horn 1987: the rotation is the unit quaternion that maximises
//     q^T N q) and T <- update * T.  With fewer than 3 correspondences the update is the identity (Open3D returns I for an empty set).
__global__ __launch_bounds__(64) void k_icp_finish(const double *part, int blocks, double *stats, double *T, int n_src, double rel_fitness,
                                                   double rel_rmse, int max_iteration, int first)
{
    int *ctl = (int *)(stats + 24);
    if (ctl[0]) return;
    const int lane = threadIdx.x;
    double t[ICP_NSUM];
#pragma unroll
    for (int k = 0; k < ICP_NSUM; k++) {
        double v = 0.0;
        for (int b = lane; b < blocks; b += 64) v += part[b * ICP_NSUM + k];
        for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m);
        t[k] = v;
    }
    if (lane != 0) return;
    const double a[3] = {stats[8], stats[9], stats[10]}, b[3] = {stats[11], stats[12], stats[13]};  // this pass's reference points
    const double n = t[0] > 0 ? t[0] : 1.0;
    const double dp[3] = {t[1] / n, t[2] / n, t[3] / n}, dq[3] = {t[4] / n, t[5] / n, t[6] / n};  // means relative to them
    const double pm[3] = {a[0] + dp[0], a[1] + dp[1], a[2] + dp[2]}, qm[3] = {b[0] + dq[0], b[1] + dq[1], b[2] + dq[2]};
    stats[0] = t[0];
    stats[7] = t[7];
    for (int k = 0; k < 3; k++) { stats[8 + k] = pm[k]; stats[11 + k] = qm[k]; }
    // ---- loop control
    const int ncorr = (int)t[0];
    const double fitness = t[0] / (double)n_src, rmse = ncorr > 0 ? sqrt(t[7] / t[0]) : 0.0;
    bool done;
    if (first) {
        ctl[1] = 0;
        done = !(0 < max_iteration && ncorr > 0);
    } else {
        ctl[1] += 1;
        const bool conv = fabs(stats[16] - fitness) < rel_fitness && fabs(stats[17] - rmse) < rel_rmse;
        done = conv || !(ctl[1] < max_iteration && ncorr > 0);
    }
    stats[16] = fitness; stats[17] = rmse;
    ctl[2] = ncorr;
    ctl[0] = done ? 1 : 0;
    if (done || t[0] < 3.0) return;
    // ---- the update from this evaluation's correspondences
    double S[3][3];
#pragma unroll
    for (int u = 0; u < 3; u++)
#pragma unroll
        for (int v = 0; v < 3; v++) S[u][v] = t[8 + u * 3 + v] - t[0] * dp[u] * dq[v];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double d[4], v[4][4];
    jacobi4(N, d, v);
    double q0 = v[0][0], qx = v[1][0], qy = v[2][0], qz = v[3][0], dm = d[0];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (d[i] > dm) { dm = d[i]; q0 = v[0][i]; qx = v[1][i]; qy = v[2][i]; qz = v[3][i]; }
    const double nq = sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
    q0 /= nq; qx /= nq; qy /= nq; qz /= nq;
    const double R[3][3] = {
        {q0 * q0 + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - q0 * qz), 2 * (qx * qz + q0 * qy)},
        {2 * (qy * qx + q0 * qz), q0 * q0 - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - q0 * qx)},
        {2 * (qz * qx - q0 * qy), 2 * (qz * qy + q0 * qx), q0 * q0 - qx * qx - qy * qy + qz * qz}};
    double U[4][4];
#pragma unroll
    for (int u = 0; u < 3; u++) {
#pragma unroll
        for (int w = 0; w < 3; w++) U[u][w] = R[u][w];
        U[u][3] = qm[u] - (R[u][0] * pm[0] + R[u][1] * pm[1] + R[u][2] * pm[2]);
    }
    U[3][0] = U[3][1] = U[3][2] = 0.0;
    U[3][3] = 1.0;
    double Tn[16];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) acc += U[u][k] * T[k * 4 + w];
            Tn[u * 4 + w] = acc;
        }
#pragma unroll
    for (int k = 0; k < 16; k++) T[k] = Tn[k];
}

struct IcpLayout {
    size_t cur, d2, idx, part, stats, T, nn_d2, nn_idx, total;
};
IcpLayout icp_layout(int n_src)
{
    IcpLayout L;
    size_t o = 0;
    L.cur = o;   o += align_up(sizeof(double) * 3 * (size_t)n_src, 256);
    L.d2 = o;    o += align_up(sizeof(double) * (size_t)n_src, 256);
    L.idx = o;   o += align_up(sizeof(int) * (size_t)n_src, 256);
    L.part = o;  o += align_up(sizeof(double) * ICP_NSUM * RED_BLOCKS, 256);
    L.stats = o; o += 256;
    L.T = o;     o += 256;
    L.nn_d2 = o;  o += align_up(sizeof(double) * NN_SLICES * (size_t)n_src, 256);
    L.nn_idx = o; o += align_up(sizeof(int) * NN_SLICES * (size_t)n_src, 256);
    L.total = o;
    return L;
}

}  // namespace
}  // namespace rcv

using namespace rcv;

extern "C" {

int rcv_nn_distance(const double *a_dev, int n_a, const double *b_dev, int n_b, double *dist_dev, int *idx_dev, void *stream)
{
    if (!a_dev || !b_dev || !dist_dev || n_a < 0 || n_b < 0) return RCV_E_INVALID;
    if (n_a == 0) return RCV_OK;
    if (n_b == 0) return RCV_E_EMPTY;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_nn, dim3((n_a + NN_THREADS - 1) / NN_THREADS), dim3(NN_THREADS), 0, st, a_dev, n_a, b_dev, n_b, dist_dev, idx_dev);
    RCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sqrt, dim3(std::min((n_a + 255) / 256, 1024)), dim3(256), 0, st, dist_dev, n_a, dist_dev);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

size_t rcv_icp_workspace_bytes(int n_src) { return icp_layout(n_src < 1 ? 1 : n_src).total; }

int rcv_icp_point_to_point(const double *src_dev, int n_src, const double *tgt_dev, int n_tgt, double max_corr_dist,
                           const double T0[16], double relative_fitness, double relative_rmse, int max_iteration,
                           rcv_icp_result *out, void *ws, size_t ws_bytes, void *stream)
{
    if (!src_dev || !tgt_dev || !T0 || !out || !ws || n_src <= 0 || n_tgt <= 0 || max_iteration < 0) return RCV_E_INVALID;
    const IcpLayout L = icp_layout(n_src);
    if (ws_bytes < L.total) return RCV_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)ws;
    double *cur = (double *)(w + L.cur), *d2 = (double *)(w + L.d2), *part = (double *)(w + L.part);
    double *stats = (double *)(w + L.stats), *T = (double *)(w + L.T);
    int *idx = (int *)(w + L.idx), *nn_idx = (int *)(w + L.nn_idx);
    double *nn_d2 = (double *)(w + L.nn_d2);
    const double max_d2 = max_corr_dist * max_corr_dist;
    RCV_HIP_CHECK(hipMemcpyAsync(T, T0, 16 * sizeof(double), hipMemcpyHostToDevice, st));
    // One evaluation = three launches: nearest neighbours of the transformed source points (16 target slices), merge + the sums
    // over the correspondences, and a one-wave kernel that closes the evaluation, decides and -- unless done -- updates T.  The
    // loop is enqueued in chunks of up to 32 iterations whose kernels test the device-side `done` flag; the host reads the control
    // words once per chunk (one chunk for Open3D's default of 30 iterations).
    // blocks of the reductions: one point per thread at most
    const int rb = std::max(1, std::min(RED_BLOCKS, (n_src + 255) / 256));
    const int *done = (const int *)(stats + 24);
    RCV_HIP_CHECK(hipMemsetAsync(stats, 0, 256, st));
    // ---- the neighbour search: hashed grid of cell edge max_corr_dist where that bins the target finely, else the sliced scan
    bool use_grid = false;
    struct PoolBlock {  // the grid's buckets: back to the stream's pool on every way out of this function
        char *p = nullptr;
        hipStream_t st;
        ~PoolBlock() { if (p) (void)hipFreeAsync(p, st); }
    } gblock;
    gblock.st = st;
    char *&gmem = gblock.p;
    int *g_starts = nullptr, *g_sorted_idx = nullptr;
    double *g_sorted = nullptr;
    unsigned g_mask = 0;
    // cells a shade wider than the reach (1 + 2^-30): two points within max_corr_dist of each other then provably fall into
    // neighbouring cells whatever the rounding of x * inv_h does (coordinates up to 10^6 cells from the origin), so the 27 scanned
    // cells hold every target the exhaustive search would accept -- identical correspondences, not just almost always
    const double inv_h = max_corr_dist > 0.0 ? 1.0 / (max_corr_dist * (1.0 + 9.313225746154785e-10)) : 0.0;
    const char *no_grid = getenv("RCV_ICP_NO_GRID");  // (tests: the exhaustive search for comparison)
    if (max_corr_dist > 1e-9 && std::isfinite(inv_h) && n_tgt >= 2048 && !(no_grid && no_grid[0] == '1')) {
        int M = 1024;
        while (M < 2 * n_tgt) M <<= 1;
        const size_t o_counts = 0, o_starts = o_counts + align_up(sizeof(int) * (size_t)M, 256);
        const size_t o_cursor = o_starts + align_up(sizeof(int) * ((size_t)M + 1), 256), o_max = o_cursor + align_up(sizeof(int) * (size_t)M, 256);
        const size_t o_sorted = o_max + 256, o_sidx = o_sorted + align_up(sizeof(double) * 3 * (size_t)n_tgt, 256);
        const size_t g_total = o_sidx + align_up(sizeof(int) * (size_t)n_tgt, 256);
        {   // the stream's pool keeps what it has handed out once (default: everything goes back to the OS at each synchronise)
            // (once per process, whatever the number of host threads calling in: std::call_once)
            static std::once_flag pool_once;
            std::call_once(pool_once, [] {
                hipMemPool_t pool;
                int dev = 0;
                if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
                    uint64_t keep = 1ull << 30;
                    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
                }
            });
        }
        if (hipMallocAsync((void **)&gmem, g_total, st) != hipSuccess) {
            (void)hipGetLastError();  // no room next to the caller's allocator: the exhaustive scan needs no extra memory
            gmem = nullptr;
        }
    }
    if (gmem) {
        int M = 1024;
        while (M < 2 * n_tgt) M <<= 1;
        const size_t o_counts = 0, o_starts = o_counts + align_up(sizeof(int) * (size_t)M, 256);
        const size_t o_cursor = o_starts + align_up(sizeof(int) * ((size_t)M + 1), 256), o_max = o_cursor + align_up(sizeof(int) * (size_t)M, 256);
        const size_t o_sorted = o_max + 256, o_sidx = o_sorted + align_up(sizeof(double) * 3 * (size_t)n_tgt, 256);
        int *counts = (int *)(gmem + o_counts), *cursor = (int *)(gmem + o_cursor), *maxc = (int *)(gmem + o_max);
        g_starts = (int *)(gmem + o_starts);
        g_sorted = (double *)(gmem + o_sorted);
        g_sorted_idx = (int *)(gmem + o_sidx);
        g_mask = (unsigned)(M - 1);
        RCV_HIP_CHECK(hipMemsetAsync(gmem, 0, o_sorted, st));
        const int gb = std::min((n_tgt + 255) / 256, 1024);
        hipLaunchKernelGGL(k_grid_count, dim3(gb), dim3(256), 0, st, tgt_dev, n_tgt, inv_h, g_mask, counts, maxc);
        hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, st, (const int *)counts, M, g_starts, (unsigned long long *)(maxc + 2));
        hipLaunchKernelGGL(k_grid_fill, dim3(gb), dim3(256), 0, st, tgt_dev, n_tgt, inv_h, g_mask, (const int *)g_starts, cursor, g_sorted,
                           g_sorted_idx);
        RCV_LAUNCH_CHECK();
        int hgrid[4] = {0, 0, 0, 0};  // largest bucket, -, sum of squared bucket counts (64 bits)
        RCV_HIP_CHECK(hipMemcpyAsync(hgrid, maxc, sizeof(hgrid), hipMemcpyDeviceToHost, st));
        RCV_HIP_CHECK(hipStreamSynchronize(st));
        unsigned long long sumsq;
        memcpy(&sumsq, hgrid + 2, sizeof(sumsq));
        // a fine binning only: the occupancy a point meets in a bucket stays small (a lane scans one bucket from global memory;
        // a reach of the size of the object would scan most of the cloud without the LDS tiles of the exhaustive kernel)
        use_grid = hgrid[0] <= 1024 && (double)sumsq <= 48.0 * (double)n_tgt;
    }
    auto evaluate = [&](int first) {
        if (use_grid) {
            hipLaunchKernelGGL(k_nn_grid, dim3((n_src + 7) / 8), dim3(256), 0, st, src_dev, n_src, (const double *)T, cur,
                               (const double *)g_sorted, (const int *)g_sorted_idx, (const int *)g_starts, inv_h, g_mask, d2, idx, done);
            hipLaunchKernelGGL(k_icp_merge_sums<1>, dim3(rb), dim3(256), 0, st, (const double *)d2, (const int *)idx, (const double *)cur,
                               tgt_dev, n_src, max_d2, (const double *)(stats + 8), d2, idx, part, done);
        } else {
            hipLaunchKernelGGL(k_nn_sliced, dim3((n_src + NN_THREADS - 1) / NN_THREADS, NN_SLICES), dim3(NN_THREADS), 0, st, src_dev, n_src,
                               (const double *)T, cur, tgt_dev, n_tgt, nn_d2, nn_idx, done);
            hipLaunchKernelGGL(k_icp_merge_sums<NN_SLICES>, dim3(rb), dim3(256), 0, st, (const double *)nn_d2, (const int *)nn_idx,
                               (const double *)cur, tgt_dev, n_src, max_d2, (const double *)(stats + 8), d2, idx, part, done);
        }
        hipLaunchKernelGGL(k_icp_finish, dim3(1), dim3(64), 0, st, (const double *)part, rb, stats, T, n_src, relative_fitness,
                           relative_rmse, max_iteration, first);
    };
    evaluate(1);
    RCV_LAUNCH_CHECK();
    double hstat[4];
    int hctl[4] = {0, 0, 0, 0};
    for (int enq = 0; enq < max_iteration;) {
        // (after the first read-back the chunks are short: once `done` is set the remaining kernels of a chunk are empty launches)
        const int chunk = std::min(enq == 0 ? 16 : 8, max_iteration - enq);
        for (int c = 0; c < chunk; c++) evaluate(0);
        RCV_LAUNCH_CHECK();
        enq += chunk;
        RCV_HIP_CHECK(hipMemcpyAsync(hctl, stats + 24, sizeof(hctl), hipMemcpyDeviceToHost, st));
        RCV_HIP_CHECK(hipStreamSynchronize(st));
        if (hctl[0]) break;
    }
    memset(out, 0, sizeof(*out));
    RCV_HIP_CHECK(hipMemcpyAsync(hctl, stats + 24, sizeof(hctl), hipMemcpyDeviceToHost, st));
    RCV_HIP_CHECK(hipMemcpyAsync(hstat, stats + 16, sizeof(hstat), hipMemcpyDeviceToHost, st));
    RCV_HIP_CHECK(hipMemcpyAsync(out->transformation, T, 16 * sizeof(double), hipMemcpyDeviceToHost, st));
    RCV_HIP_CHECK(hipStreamSynchronize(st));
    const double fitness = hstat[0], rmse = hstat[1];
    const int ncorr = hctl[2], it = hctl[1];
    out->fitness = fitness;
    out->inlier_rmse = rmse;
    out->n_correspondences = ncorr;
    out->iterations = it;
    out->status = RCV_OK;
    return RCV_OK;
}

}  // extern "C"
