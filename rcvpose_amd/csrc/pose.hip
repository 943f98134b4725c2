// pose.hip -- Horn absolute orientation on the device.
//
// Reference: util/horn.py:75-181 (HornPoseFitting.lmshorn) with its Jacobi eigen-solver (:12-72), called on the
// three model / estimated keypoints of a frame at AccumulatorSpace.py:659-661.  One thread per frame: the
// estimated keypoints are read straight out of the result records of rcv_accumulate_3d, so the pose is formed
// on the device before the (single) gather.  float64, the reference's operation order, no FMA contraction
// (results are bit-identical to the numpy implementation on the golden cases).
#pragma clang fp contract(off)

#include <math.h>

#include "common.hpp"

namespace rcv {
namespace {

constexpr int HORN_MAX_POINTS = 16;

__device__ void horn_rotate(double (*a)[5], int i, int j, int k, int l, double s, double tau)
{
    const double g = a[i][j], h = a[k][l];
    a[i][j] = g - s * (h + g * tau);
    a[k][l] = h + s * (g - h * tau);
}

// Jacobi eigen-decomposition of the symmetric 4x4 matrix held 1-based in a 5x5 array (util/horn.py:12-72)
__device__ void horn_jacobi(double (*a)[5], int n, double *d, double (*v)[5])
{
    double b[5] = {0, 0, 0, 0, 0}, z[5] = {0, 0, 0, 0, 0};
    for (int ip = 1; ip <= n; ip++) {
        for (int iq = 1; iq <= n; iq++) v[ip][iq] = 0.0;
        v[ip][ip] = 1.0;
    }
    for (int ip = 1; ip <= n; ip++) { b[ip] = d[ip] = a[ip][ip]; z[ip] = 0.0; }
    for (int i = 1; i <= 50; i++) {
        double sm = 0.0;
        for (int ip = 1; ip < n; ip++)  // rows 1..n-1, all columns, as the reference sums them (:27-29)
            for (int iq = 1; iq <= n; iq++) sm += fabs(a[ip][iq]);
        if (sm == 0.0) return;
        const double tresh = i < 4 ? 0.2 * sm / (double)(n * n) : 0.0;
        for (int ip = 1; ip < n; ip++)
            for (int iq = ip + 1; iq <= n; iq++) {
                const double g = 100.0 * fabs(a[ip][iq]);
                if (i > 4 && fabs(d[ip]) + g == fabs(d[ip]) && fabs(d[iq]) + g == fabs(d[iq])) {
                    a[ip][iq] = 0.0;
                } else if (fabs(a[ip][iq]) > tresh) {
                    double h = d[iq] - d[ip], t;
                    if (fabs(h) + g == fabs(h)) {
                        t = a[ip][iq] / h;
                    } else {
                        const double theta = 0.5 * h / a[ip][iq];
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);
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

__global__ void k_horn(const double *model, const unsigned char *est, long long stride_frame, long long stride_kp, int n,
                       int n_frames, double *rt_out)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    double P1[HORN_MAX_POINTS][3], P2[HORN_MAX_POINTS][3];
    double C1[3] = {0, 0, 0}, C2[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double *e = (const double *)(est + (long long)f * stride_frame + (long long)i * stride_kp);
        for (int j = 0; j < 3; j++) {
            P1[i][j] = model[3 * i + j];
            P2[i][j] = e[j];
            C1[j] += P1[i][j];
            C2[j] += P2[i][j];
        }
    }
    for (int j = 0; j < 3; j++) { C1[j] /= (double)n; C2[j] /= (double)n; }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) { P1[i][j] -= C1[j]; P2[i][j] -= C2[j]; }
    double Sxx = 0, Sxy = 0, Sxz = 0, Syx = 0, Syy = 0, Syz = 0, Szx = 0, Szy = 0, Szz = 0;
    for (int i = 0; i < n; i++) {
        Sxx += P1[i][0] * P2[i][0]; Sxy += P1[i][0] * P2[i][1]; Sxz += P1[i][0] * P2[i][2];
        Syx += P1[i][1] * P2[i][0]; Syy += P1[i][1] * P2[i][1]; Syz += P1[i][1] * P2[i][2];
        Szx += P1[i][2] * P2[i][0]; Szy += P1[i][2] * P2[i][1]; Szz += P1[i][2] * P2[i][2];
    }
    double N[5][5], D[5] = {0, 0, 0, 0, 0}, V[5][5];
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) { N[i][j] = 0.0; V[i][j] = 0.0; }
    N[1][1] = Sxx + Syy + Szz; N[1][2] = Syz - Szy; N[1][3] = Szx - Sxz; N[1][4] = Sxy - Syx;
    N[2][1] = Syz - Szy; N[2][2] = Sxx - Syy - Szz; N[2][3] = Sxy + Syx; N[2][4] = Szx + Sxz;
    N[3][1] = Szx - Sxz; N[3][2] = Sxy + Syx; N[3][3] = -Sxx + Syy - Szz; N[3][4] = Syz + Szy;
    N[4][1] = Sxy - Syx; N[4][2] = Szx + Sxz; N[4][3] = Syz + Szy; N[4][4] = -Sxx - Syy + Szz;
    horn_jacobi(N, 4, D, V);
    int mi = 1;
    double mv = D[1];
    for (int i = 2; i < 5; i++)
        if (D[i] > mv) { mi = i; mv = D[i]; }
    const double q0 = V[1][mi], q1 = V[2][mi], q2 = V[3][mi], q3 = V[4][mi];
    double R[3][3];
    R[0][0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; R[0][1] = 2 * (q1 * q2 - q0 * q3); R[0][2] = 2 * (q1 * q3 + q0 * q2);
    R[1][0] = 2 * (q1 * q2 + q0 * q3); R[1][1] = q0 * q0 + q2 * q2 - q1 * q1 - q3 * q3; R[1][2] = 2 * (q2 * q3 - q0 * q1);
    R[2][0] = 2 * (q1 * q3 - q0 * q2); R[2][1] = 2 * (q2 * q3 + q0 * q1); R[2][2] = q0 * q0 + q3 * q3 - q1 * q1 - q2 * q2;
    double *A = rt_out + 16 * (long long)f;
    for (int i = 0; i < 3; i++) {
        A[4 * i + 3] = C2[i] - (R[i][0] * C1[0] + R[i][1] * C1[1] + R[i][2] * C1[2]);
        for (int j = 0; j < 3; j++) A[4 * i + j] = R[i][j];
        A[12 + i] = 0.0;
    }
    A[15] = 1.0;
}

}  // namespace
}  // namespace rcv

extern "C" int rcv_horn_pose_batch(const double *model_kpts_dev, rcv_strided est_kpts, int n_points, int n_frames,
                                   double *rt_out_dev, void *stream)
{
    if (!model_kpts_dev || !est_kpts.ptr || !rt_out_dev || n_points < 1 || n_points > rcv::HORN_MAX_POINTS || n_frames < 1)
        return RCV_E_INVALID;
    hipLaunchKernelGGL(rcv::k_horn, dim3((n_frames + 63) / 64), dim3(64), 0, (hipStream_t)stream, model_kpts_dev,
                       (const unsigned char *)est_kpts.ptr, (long long)est_kpts.stride_frame, (long long)est_kpts.stride_kp,
                       n_points, n_frames, rt_out_dev);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}
