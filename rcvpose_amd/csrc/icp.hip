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
#include <string.h>

#include <vector>

#include "common.hpp"

namespace rcv {
namespace {

constexpr int NN_THREADS = 256, NN_TILE = 1024, RED_BLOCKS = 256;

// cur[i] = R * src[i] + t  (T row-major 4x4)
// (`done`: the device-side loop control of rcv_icp_point_to_point -- once set, the remaining launches of the iteration chunk
// return at once; null for callers without one)
__global__ __launch_bounds__(256) void k_icp_transform(const double *src, int n, const double *T, double *cur, const int *done)
{
    if (done && *done) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
        cur[3 * i] = T[0] * x + T[1] * y + T[2] * z + T[3];
        cur[3 * i + 1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        cur[3 * i + 2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
}

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
// (squared distance, index) in part_*[s][i]; k_nn_merge folds the slices in order with the same strict `<`, which is the first
// index among equal distances again -- identical results.
constexpr int NN_SLICES = 16;
__global__ __launch_bounds__(NN_THREADS) void k_nn_sliced(const double *a, int n_a, const double *b, int n_b, double *part_d2, int *part_idx,
                                                          const int *done)
{
    if (done && *done) return;
    __shared__ double sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
    const int i = blockIdx.x * NN_THREADS + threadIdx.x;
    const int ii = i < n_a ? i : n_a - 1;
    const int per = (n_b + NN_SLICES - 1) / NN_SLICES;
    const int lo = blockIdx.y * per, hi = min(n_b, lo + per);
    const double px = a[3 * ii], py = a[3 * ii + 1], pz = a[3 * ii + 2];
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
__global__ __launch_bounds__(256) void k_nn_merge(const double *part_d2, const int *part_idx, int n_a, double *d2_out, int *idx_out, const int *done)
{
    if (done && *done) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_a; i += gridDim.x * 256) {
        double best = INFINITY;
        int bi = -1;
        for (int s = 0; s < NN_SLICES; s++) {
            const double d2 = part_d2[(size_t)s * n_a + i];
            if (d2 < best) { best = d2; bi = part_idx[(size_t)s * n_a + i]; }
        }
        d2_out[i] = best;
        idx_out[i] = bi;
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

// pass 1 over the correspondences (source i <-> target idx[i], d2[i] <= max_d2): count, sum p, sum q, sum d2
// per block -> part[block][8]; fixed block / thread assignment makes the result reproducible
__global__ __launch_bounds__(256) void k_icp_sums(const double *cur, const double *tgt, const double *d2, const int *idx, int n,
                                                  double max_d2, double *part, const int *done)
{
    if (done && *done) return;
    __shared__ double s[4];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (idx[i] >= 0 && d2[i] <= max_d2) {
            const int j = idx[i];
            acc[0] += 1.0;
            acc[1] += cur[3 * i]; acc[2] += cur[3 * i + 1]; acc[3] += cur[3 * i + 2];
            acc[4] += tgt[3 * j]; acc[5] += tgt[3 * j + 1]; acc[6] += tgt[3 * j + 2];
            acc[7] += d2[i];
        }
    }
    for (int k = 0; k < 8; k++) {
        const double v = block_sum(acc[k], s);
        if (threadIdx.x == 0) part[blockIdx.x * 8 + k] = v;
    }
}

// reduce the partials: stats[0..7] = totals, stats[8..10] = mean p, stats[11..13] = mean q
__global__ void k_icp_finish_sums(const double *part, int blocks, double *stats, const int *done)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || (done && *done)) return;
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < blocks; b++)
        for (int k = 0; k < 8; k++) t[k] += part[b * 8 + k];
    for (int k = 0; k < 8; k++) stats[k] = t[k];
    const double n = t[0] > 0 ? t[0] : 1.0;
    for (int k = 0; k < 3; k++) { stats[8 + k] = t[1 + k] / n; stats[11 + k] = t[4 + k] / n; }
}

// pass 2: centred cross-covariance S[a][b] = sum (p - pm)_a (q - qm)_b  -> part[block][9]
__global__ __launch_bounds__(256) void k_icp_cov(const double *cur, const double *tgt, const double *d2, const int *idx, int n,
                                                 double max_d2, const double *stats, double *part, const int *done)
{
    if (done && *done) return;
    __shared__ double s[4];
    const double pm[3] = {stats[8], stats[9], stats[10]}, qm[3] = {stats[11], stats[12], stats[13]};
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (idx[i] >= 0 && d2[i] <= max_d2) {
            const int j = idx[i];
            const double p[3] = {cur[3 * i] - pm[0], cur[3 * i + 1] - pm[1], cur[3 * i + 2] - pm[2]};
            const double q[3] = {tgt[3 * j] - qm[0], tgt[3 * j + 1] - qm[1], tgt[3 * j + 2] - qm[2]};
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) acc[a * 3 + b] += p[a] * q[b];
        }
    }
    for (int k = 0; k < 9; k++) {
        const double v = block_sum(acc[k], s);
        if (threadIdx.x == 0) part[blockIdx.x * 9 + k] = v;
    }
}

// cyclic Jacobi for a symmetric 4x4 matrix: eigenvalues d, eigenvectors in the columns of v
__device__ void jacobi4(double a[4][4], double d[4], double v[4][4])
{
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) v[i][j] = i == j ? 1.0 : 0.0;
        d[i] = a[i][i];
    }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++) off += a[i][j] * a[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; p++)
            for (int q = p + 1; q < 4; q++) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) {  // A <- A * J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {  // A <- J^T * A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 4; i++) d[i] = a[i][i];
}

// rigid update from the covariance (Horn 1987: the rotation is the unit quaternion that maximises q^T N q) and
// T <- update * T.  With fewer than 3 correspondences the update is the identity (Open3D returns I for an empty set).
__global__ void k_icp_update(const double *part, int blocks, const double *stats, double *T, const int *done)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || (done && *done)) return;
    if (stats[0] < 3.0) return;
    double S[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double t = 0.0;
            for (int k = 0; k < blocks; k++) t += part[k * 9 + a * 3 + b];
            S[a][b] = t;
        }
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double d[4], v[4][4];
    jacobi4(N, d, v);
    int m = 0;
    for (int i = 1; i < 4; i++)
        if (d[i] > d[m]) m = i;
    double q0 = v[0][m], qx = v[1][m], qy = v[2][m], qz = v[3][m];
    const double nq = sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
    q0 /= nq; qx /= nq; qy /= nq; qz /= nq;
    const double R[3][3] = {
        {q0 * q0 + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - q0 * qz), 2 * (qx * qz + q0 * qy)},
        {2 * (qy * qx + q0 * qz), q0 * q0 - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - q0 * qx)},
        {2 * (qz * qx - q0 * qy), 2 * (qz * qy + q0 * qx), q0 * q0 - qx * qx - qy * qy + qz * qz}};
    const double pm[3] = {stats[8], stats[9], stats[10]}, qm[3] = {stats[11], stats[12], stats[13]};
    double U[4][4] = {{0}};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) U[a][b] = R[a][b];
        U[a][3] = qm[a] - (R[a][0] * pm[0] + R[a][1] * pm[1] + R[a][2] * pm[2]);
    }
    U[3][3] = 1.0;
    double Tn[16];
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
            double t = 0.0;
            for (int k = 0; k < 4; k++) t += U[a][k] * T[k * 4 + b];
            Tn[a * 4 + b] = t;
        }
    for (int k = 0; k < 16; k++) T[k] = Tn[k];
}

// Loop control of the ICP on the device (r03: the host used to read the fitness back after every iteration -- 30 synchronisations
// of ~0.4 ms per frame).  stats[16..19] = previous / current (fitness, rmse); ctl = (int *)(stats + 24): done, iterations, correspondences.
// `first`: after the evaluation of the start pose; otherwise after an update + evaluation -- the reference loop's bookkeeping:
//     it++; converged (|d fitness| < relative_fitness and |d rmse| < relative_rmse) or it == max or no correspondences -> done
__global__ void k_icp_decide(double *stats, int n_src, double rel_fitness, double rel_rmse, int max_iteration, int first)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int *ctl = (int *)(stats + 24);
    if (ctl[0]) return;
    const int ncorr = (int)stats[0];
    const double fitness = stats[0] / (double)n_src, rmse = ncorr > 0 ? sqrt(stats[7] / stats[0]) : 0.0;
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
    L.part = o;  o += align_up(sizeof(double) * 9 * RED_BLOCKS, 256);
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
    const int tb = std::min((n_src + 255) / 256, 1024);
    RCV_HIP_CHECK(hipMemcpyAsync(T, T0, 16 * sizeof(double), hipMemcpyHostToDevice, st));
    // The whole loop is enqueued in chunks of up to 32 iterations whose kernels test the device-side `done` flag; the host reads
    // the control words once per chunk (one chunk for Open3D's default of 30 iterations).
    // blocks of the two-pass reductions: one point per thread at most (the single thread that adds the per-block partials was
    // most of an iteration when 256 blocks shared 3 000 points)
    const int rb = std::max(1, std::min(RED_BLOCKS, (n_src + 255) / 256));
    const int *done = (const int *)(stats + 24);
    RCV_HIP_CHECK(hipMemsetAsync(stats, 0, 256, st));
    auto evaluate = [&]() {
        hipLaunchKernelGGL(k_icp_transform, dim3(tb), dim3(256), 0, st, src_dev, n_src, (const double *)T, cur, done);
        hipLaunchKernelGGL(k_nn_sliced, dim3((n_src + NN_THREADS - 1) / NN_THREADS, NN_SLICES), dim3(NN_THREADS), 0, st, (const double *)cur,
                           n_src, tgt_dev, n_tgt, nn_d2, nn_idx, done);
        hipLaunchKernelGGL(k_nn_merge, dim3(tb), dim3(256), 0, st, (const double *)nn_d2, (const int *)nn_idx, n_src, d2, idx, done);
        hipLaunchKernelGGL(k_icp_sums, dim3(rb), dim3(256), 0, st, (const double *)cur, tgt_dev, (const double *)d2,
                           (const int *)idx, n_src, max_d2, part, done);
        hipLaunchKernelGGL(k_icp_finish_sums, dim3(1), dim3(64), 0, st, (const double *)part, rb, stats, done);
    };
    evaluate();
    hipLaunchKernelGGL(k_icp_decide, dim3(1), dim3(64), 0, st, stats, n_src, relative_fitness, relative_rmse, max_iteration, 1);
    RCV_LAUNCH_CHECK();
    double hstat[4];
    int hctl[4] = {0, 0, 0, 0};
    for (int enq = 0; enq < max_iteration;) {
        const int chunk = std::min(32, max_iteration - enq);
        for (int c = 0; c < chunk; c++) {
            // the update uses the correspondences of the current evaluation (cur, idx, d2, stats are still in place)
            hipLaunchKernelGGL(k_icp_cov, dim3(rb), dim3(256), 0, st, (const double *)cur, tgt_dev, (const double *)d2,
                               (const int *)idx, n_src, max_d2, (const double *)stats, part, done);
            hipLaunchKernelGGL(k_icp_update, dim3(1), dim3(64), 0, st, (const double *)part, rb, (const double *)stats, T, done);
            evaluate();
            hipLaunchKernelGGL(k_icp_decide, dim3(1), dim3(64), 0, st, stats, n_src, relative_fitness, relative_rmse, max_iteration, 0);
        }
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
