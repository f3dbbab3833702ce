// Per-clip pose-error sums of the evaluation loop, on the device in float64.
//
// Replaces, for one clip of N frames, the host side of Trainer.evaluate_core after the forward
// (lib/train_val/trainer.py:355-397): D2H copy, cam.normalized2world in NumPy (lib/camera/camera.py:401-410),
// mpjpe / n_mpjpe (lib/loss/loss.py:12-18, :72-82, torch), p_mpjpe (:30-69, NumPy SVD) and mean_velocity_error
// (:95-104).  Everything is per frame and independent, so a thread owns a frame: it moves prediction and ground
// truth to world coordinates in float64 (fp32 inputs promoted, as NumPy does), evaluates the four per-frame
// errors and the first-difference error against the next frame; a fixed-order tree adds a workgroup's frames up
// and a second, one-wavefront launch adds the workgroups' partial sums in index order - no atomics, the same bits
// on every run.
#include <hip/hip_runtime.h>
#include "r3d_internal.hpp"

namespace r3d {

namespace {

constexpr int METRIC_THREADS = 256;
constexpr int MAX_J = 17;

struct MetricArgs {
    const float *pred, *gt;      // (N, J, 3) each, normalised frame
    double *out;                 // R3D_METRIC_COUNT sums, then R3D_METRIC_COUNT partial sums per workgroup
    long long n;
    int J;
    double R[9], T[3];           // world = R @ p + T  (Rn2w, Tn2w: camera.py:258-259, :401-410)
};

__device__ inline void to_world(const MetricArgs &a, const float *src, double (*dst)[3]) {
    for (int j = 0; j < a.J; ++j) {
        const double x = (double)src[3 * j], y = (double)src[3 * j + 1], z = (double)src[3 * j + 2];
        for (int r = 0; r < 3; ++r) dst[j][r] = a.R[3 * r] * x + a.R[3 * r + 1] * y + a.R[3 * r + 2] * z + a.T[r];
    }
}

// Singular value decomposition of a 3x3 matrix by one-sided Jacobi rotations (Hestenes): columns of A are rotated
// until mutually orthogonal, A = U diag(s) V^T with U's columns the normalised columns of the result.  Returns the
// columns ordered by decreasing singular value, as LAPACK (np.linalg.svd in loss.py:50) does.
__device__ inline void svd3(double A[3][3], double U[3][3], double s[3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) {
                    al += A[i][p] * A[i][p];
                    be += A[i][q] * A[i][q];
                    ga += A[i][p] * A[i][q];
                }
                if (fabs(ga) <= 1e-17 * sqrt(al * be) || ga == 0.0) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - sn * aq;
                    A[i][q] = sn * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - sn * vq;
                    V[i][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int order[3] = {0, 1, 2};
    double nrm[3];
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (nrm[order[j]] < nrm[order[j + 1]]) { const int t = order[j]; order[j] = order[j + 1]; order[j + 1] = t; }
    double Vs[3][3];
    for (int k = 0; k < 3; ++k) {
        const int j = order[k];
        s[k] = nrm[j];
        for (int i = 0; i < 3; ++i) {
            U[i][k] = nrm[j] > 0 ? A[i][j] / nrm[j] : 0.0;
            Vs[i][k] = V[i][j];
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) V[i][k] = Vs[i][k];
    // a (nearly) vanishing singular value - planar poses - leaves its left vector to rounding noise: complete the
    // orthonormal basis instead, keeping the column's side (for an exactly zero column either side gives the same
    // aligned pose once the reflection test below has fixed the sign)
    if (s[2] <= 1e-7 * s[0]) {
        const double c0 = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        const double c1 = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        const double c2 = U[0][0] * U[1][1] - U[1][0] * U[0][1];
        const double side = c0 * U[0][2] + c1 * U[1][2] + c2 * U[2][2] < 0 ? -1.0 : 1.0;
        U[0][2] = side * c0;
        U[1][2] = side * c1;
        U[2][2] = side * c2;
    }
}

// mean_j || a * (p_j R) + t - g_j || after the similarity alignment of loss.py:35-66 (target = g, predicted = p)
__device__ inline double procrustes_error(const double (*p)[3], const double (*g)[3], int J) {
    double mu_g[3] = {0, 0, 0}, mu_p[3] = {0, 0, 0};
    for (int j = 0; j < J; ++j)
        for (int r = 0; r < 3; ++r) { mu_g[r] += g[j][r]; mu_p[r] += p[j][r]; }
    for (int r = 0; r < 3; ++r) { mu_g[r] /= J; mu_p[r] /= J; }
    double ng = 0, np_ = 0;
    for (int j = 0; j < J; ++j)
        for (int r = 0; r < 3; ++r) {
            const double a = g[j][r] - mu_g[r], b = p[j][r] - mu_p[r];
            ng += a * a;
            np_ += b * b;
        }
    ng = sqrt(ng);
    np_ = sqrt(np_);
    // H = X0^T Y0 with X0 = (g - mu_g)/|.|, Y0 = (p - mu_p)/|.|   (:47)
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < J; ++j)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[r][c] += ((g[j][r] - mu_g[r]) / ng) * ((p[j][c] - mu_p[c]) / np_);
    double U[3][3], s[3], V[3][3];
    svd3(H, U, s, V);
    // R = V U^T; reflections are undone by negating V's last column and the last singular value (:51-57)
    double Rm[3][3];
    auto vut = [&]() {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rm[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
    };
    vut();
    const double det = Rm[0][0] * (Rm[1][1] * Rm[2][2] - Rm[1][2] * Rm[2][1]) - Rm[0][1] * (Rm[1][0] * Rm[2][2] - Rm[1][2] * Rm[2][0]) +
                       Rm[0][2] * (Rm[1][0] * Rm[2][1] - Rm[1][1] * Rm[2][0]);
    const double sg = det > 0 ? 1.0 : (det < 0 ? -1.0 : 0.0);
    for (int r = 0; r < 3; ++r) V[r][2] *= sg;
    s[2] *= sg;
    vut();
    const double a = (s[0] + s[1] + s[2]) * ng / np_;                       // :61
    double t[3];                                                            // t = mu_g - a * (mu_p R)   (:62)
    for (int c = 0; c < 3; ++c) t[c] = mu_g[c] - a * (mu_p[0] * Rm[0][c] + mu_p[1] * Rm[1][c] + mu_p[2] * Rm[2][c]);
    double e = 0;
    for (int j = 0; j < J; ++j) {
        double d2 = 0;
        for (int c = 0; c < 3; ++c) {
            const double v = a * (p[j][0] * Rm[0][c] + p[j][1] * Rm[1][c] + p[j][2] * Rm[2][c]) + t[c] - g[j][c];
            d2 += v * v;
        }
        e += sqrt(d2);
    }
    return e / J;
}

__global__ __launch_bounds__(METRIC_THREADS) void r3d_clip_metrics_f64(MetricArgs a) {
    __shared__ double red[R3D_METRIC_COUNT][METRIC_THREADS];
    double acc[R3D_METRIC_COUNT] = {0, 0, 0, 0, 0};
    const int J = a.J;
    for (long long f = (long long)blockIdx.x * METRIC_THREADS + threadIdx.x; f < a.n; f += (long long)gridDim.x * METRIC_THREADS) {
        double p[MAX_J][3], g[MAX_J][3];
        to_world(a, a.pred + f * J * 3, p);
        to_world(a, a.gt + f * J * 3, g);
        // mpjpe (loss.py:17-18) and its root-joint restriction (trainer.py:387)
        double e = 0, pp = 0, gp = 0;
        for (int j = 0; j < J; ++j) {
            double d2 = 0;
            for (int r = 0; r < 3; ++r) {
                const double d = p[j][r] - g[j][r];
                d2 += d * d;
                pp += p[j][r] * p[j][r];
                gp += g[j][r] * p[j][r];
            }
            const double d = sqrt(d2);
            e += d;
            if (j == 0) acc[R3D_METRIC_ROOT] += d;
        }
        acc[R3D_METRIC_MPJPE] += e / J;
        // n_mpjpe: scale = mean_j <g,p> / mean_j <p,p>   (loss.py:78-82)
        const double sc = (gp / J) / (pp / J);
        double en = 0;
        for (int j = 0; j < J; ++j) {
            double d2 = 0;
            for (int r = 0; r < 3; ++r) {
                const double d = sc * p[j][r] - g[j][r];
                d2 += d * d;
            }
            en += sqrt(d2);
        }
        acc[R3D_METRIC_NMPJPE] += en / J;
        acc[R3D_METRIC_PMPJPE] += procrustes_error(p, g, J);
        // mean_velocity_error: first differences along the clip (loss.py:101-104)
        if (f + 1 < a.n) {
            double p1[MAX_J][3], g1[MAX_J][3];
            to_world(a, a.pred + (f + 1) * J * 3, p1);
            to_world(a, a.gt + (f + 1) * J * 3, g1);
            double ev = 0;
            for (int j = 0; j < J; ++j) {
                double d2 = 0;
                for (int r = 0; r < 3; ++r) {
                    const double d = (p1[j][r] - p[j][r]) - (g1[j][r] - g[j][r]);
                    d2 += d * d;
                }
                ev += sqrt(d2);
            }
            acc[R3D_METRIC_VELOCITY] += ev / J;
        }
    }
    for (int k = 0; k < R3D_METRIC_COUNT; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = METRIC_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int k = 0; k < R3D_METRIC_COUNT; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < R3D_METRIC_COUNT; ++k) a.out[R3D_METRIC_COUNT * (1 + blockIdx.x) + k] = red[k][0];
}

__global__ __launch_bounds__(64) void r3d_clip_metrics_sum_f64(double *out, int blocks, long long n) {
    const int k = threadIdx.x;
    if (k >= R3D_METRIC_COUNT) return;
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += out[R3D_METRIC_COUNT * (1 + b) + k];
    // n * mean over the n-1 differences (trainer.py:395 weights the clip's mean by its frame count); an empty mean
    // is NaN in NumPy
    if (k == R3D_METRIC_VELOCITY) s = n > 1 ? s * ((double)n / (double)(n - 1)) : nan("");
    out[k] = s;
}

}  // namespace

int launch_clip_metrics(const float *pred, const float *gt, long long n, int J, const double *Rn2w, const double *Tn2w,
                        double *out, hipStream_t stream) {
    MetricArgs a;
    a.pred = pred;
    a.gt = gt;
    a.out = out;
    a.n = n;
    a.J = J;
    for (int i = 0; i < 9; ++i) a.R[i] = Rn2w[i];
    for (int i = 0; i < 3; ++i) a.T[i] = Tn2w[i];
    long long blocks = (n + METRIC_THREADS - 1) / METRIC_THREADS;
    blocks = blocks < 1 ? 1 : (blocks > R3D_METRIC_MAX_BLOCKS ? R3D_METRIC_MAX_BLOCKS : blocks);
    hipLaunchKernelGGL(r3d_clip_metrics_f64, dim3((unsigned)blocks), dim3(METRIC_THREADS), 0, stream, a);
    hipLaunchKernelGGL(r3d_clip_metrics_sum_f64, dim3(1), dim3(64), 0, stream, out, (int)blocks, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace r3d
