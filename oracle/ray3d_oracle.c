/*
 * ray3d_oracle.c - CPU restatement of the Ray3D lifting forward pass (TEST INFRASTRUCTURE ONLY;
 * see ray3d_oracle.h for the rules and the parity status).
 *
 * The code follows the reference op by op (Conv1d -> BatchNorm1d(eval) -> LeakyReLU, Linear,
 * concatenations), WITHOUT BatchNorm folding or any of the product's layout tricks, so that it
 * is an independent statement of the algorithm.  Activations are float32 like the reference;
 * dot products accumulate in double and round once (slightly more accurate than ATen).
 * Internally activations are channels-last (B, T, C); this is a storage choice only.
 */
#include "ray3d_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BN_EPS 1e-5
#define HIDDEN 1024
#define EMB_MID 32

static char g_err[512];

const char *r3o_last_error(void) { return g_err; }

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

/* ------------------------------------------------------------------ tensor lookup */

typedef struct {
    const r3o_tensor *t;
    int n;
    int err;
} store;

static const r3o_tensor *find(store *s, const char *key) {
    for (int i = 0; i < s->n; ++i)
        if (strcmp(s->t[i].key, key) == 0) return &s->t[i];
    if (!s->err) s->err = fail(-1, "missing tensor '%s'", key);
    return NULL;
}

static const float *want(store *s, const char *key, int rank, int64_t d0, int64_t d1, int64_t d2) {
    const r3o_tensor *t = find(s, key);
    if (!t) return NULL;
    int64_t d[3] = {d0, d1, d2};
    int ok = (t->rank == rank);
    for (int i = 0; ok && i < rank; ++i) ok = (t->shape[i] == d[i]);
    if (!ok) {
        if (!s->err)
            s->err = fail(-2, "tensor '%s' has the wrong shape (want rank %d [%lld,%lld,%lld])", key,
                          rank, (long long)d0, (long long)d1, (long long)d2);
        return NULL;
    }
    return t->data;
}

/* ------------------------------------------------------------------ primitive ops */

/* y[m, :] = x[m, :] @ W^T (+ bias);  W is (N, K) row-major like nn.Linear.weight. */
static void linear_rows(const float *x, int64_t M, int64_t K, int64_t ldx, const float *W,
                        const float *bias, int64_t N, float *y, int64_t ldy) {
    /* transpose W once so the inner loop is contiguous in n */
    float *Wt = (float *)malloc(sizeof(float) * (size_t)(K * N));
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) Wt[k * N + n] = W[n * K + k];
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)N);
#pragma omp for schedule(static)
        for (int64_t m = 0; m < M; ++m) {
            for (int64_t n = 0; n < N; ++n) acc[n] = bias ? (double)bias[n] : 0.0;
            const float *xr = x + m * ldx;
            for (int64_t k = 0; k < K; ++k) {
                const double a = (double)xr[k];
                const float *w = Wt + k * N;
                for (int64_t n = 0; n < N; ++n) acc[n] += a * (double)w[n];
            }
            float *yr = y + m * ldy;
            for (int64_t n = 0; n < N; ++n) yr[n] = (float)acc[n];
        }
        free(acc);
    }
    free(Wt);
}

/* nn.Conv1d(Cin, Cout, k, stride=k, bias) on channels-last x (B*T, Cin) -> y (B*T/k, Cout).
 * weight is torch layout (Cout, Cin, k): y[r, o] = sum_{j,c} x[r*k + j, c] * w[o, c, j].
 * (T is a multiple of k, so rows never straddle windows.)  lib/model/rie.py:36-38,55-57 */
static void conv_stride_k(const float *x, int64_t rows_in, int64_t Cin, const float *w, int k,
                          const float *bias, int64_t Cout, float *y) {
    const int64_t K = (int64_t)k * Cin;
    /* repack to (Cout, k*Cin) with index j*Cin + c so a run of k input rows is one GEMM row */
    float *W2 = (float *)malloc(sizeof(float) * (size_t)(Cout * K));
    for (int64_t o = 0; o < Cout; ++o)
        for (int64_t c = 0; c < Cin; ++c)
            for (int j = 0; j < k; ++j) W2[o * K + j * Cin + c] = w[(o * Cin + c) * k + j];
    linear_rows(x, rows_in / k, K, K, W2, bias, Cout, y, Cout);
    free(W2);
}

/* nn.BatchNorm1d in eval mode over the channel (last) axis, then optional LeakyReLU. */
static void bn_eval(float *x, int64_t rows, int64_t C, const float *gamma, const float *beta,
                    const float *mean, const float *var) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float *p = x + r * C;
        for (int64_t c = 0; c < C; ++c) {
            const double inv = 1.0 / sqrt((double)var[c] + BN_EPS);
            p[c] = (float)(((double)p[c] - (double)mean[c]) * inv * (double)gamma[c] + (double)beta[c]);
        }
    }
}

static void leaky(float *x, int64_t n, float slope) {
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : x[i] * slope;
}

static int bn_named(store *s, const char *prefix, float *x, int64_t rows, int64_t C) {
    char key[256];
    const float *g, *b, *m, *v;
    snprintf(key, sizeof key, "%s.weight", prefix);       g = want(s, key, 1, C, 0, 0);
    snprintf(key, sizeof key, "%s.bias", prefix);         b = want(s, key, 1, C, 0, 0);
    snprintf(key, sizeof key, "%s.running_mean", prefix); m = want(s, key, 1, C, 0, 0);
    snprintf(key, sizeof key, "%s.running_var", prefix);  v = want(s, key, 1, C, 0, 0);
    if (!g || !b || !m || !v) return s->err;
    bn_eval(x, rows, C, g, b, m, v);
    return 0;
}

static void emit(r3o_tap_fn tap, void *user, const char *name, const float *d, int64_t a,
                 int64_t b, int64_t c, int rank) {
    if (!tap) return;
    int64_t shape[3] = {a, b, c};
    tap(name, d, shape, rank, user);
}

/* ------------------------------------------------------------------ sub-modules */

/* TemporalBlock.forward, lib/model/rie.py:85-105, on one RF-long window: the strided (Optimize1f) form; the
 * dilated form (:91-92) evaluates the same ternary tree, with the residual at the last tap when causal.
 * x: channels-last (B*T, Cin) with T = 3^L.  out: (B, latent). */
static int temporal_block_dense(store *s, const r3o_config *cfg, const char *prefix, const float *x,
                                int64_t B, int64_t T, int64_t Cin, float *out, r3o_tap_fn tap, void *user);
static int temporal_block(store *s, const r3o_config *cfg, const char *prefix, const float *x,
                          int64_t B, int64_t T, int64_t Cin, float *out, r3o_tap_fn tap, void *user) {
    if (cfg->dense) return temporal_block_dense(s, cfg, prefix, x, B, T, Cin, out, tap, user);
    const int64_t C = cfg->channels;
    char key[256], tapname[256];
    int64_t rows = B * T / 3;
    float *cur = (float *)malloc(sizeof(float) * (size_t)(rows * C));
    snprintf(key, sizeof key, "%s.expand_conv.weight", prefix);
    const float *w = want(s, key, 3, C, Cin, 3);
    if (!w) { free(cur); return s->err; }
    conv_stride_k(x, B * T, Cin, w, 3, NULL, C, cur);                       /* :86 expand_conv */
    snprintf(key, sizeof key, "%s.expand_bn", prefix);
    if (bn_named(s, key, cur, rows, C)) { free(cur); return s->err; }       /* :86 expand_bn   */
    snprintf(tapname, sizeof tapname, "%s.level0.pre", prefix);
    emit(tap, user, tapname, cur, B, T / 3, C, 3);
    leaky(cur, rows * C, 0.2f);                                             /* :86 relu (drop=id) */
    int64_t Tcur = T / 3;
    for (int i = 0; i < cfg->num_levels - 1; ++i) {
        const int64_t rows2 = B * Tcur / 3;
        float *h = (float *)malloc(sizeof(float) * (size_t)(rows2 * C));
        float *g = (float *)malloc(sizeof(float) * (size_t)(rows2 * C));
        snprintf(key, sizeof key, "%s.layers_conv.%d.weight", prefix, 2 * i);
        const float *wa = want(s, key, 3, C, C, 3);
        snprintf(key, sizeof key, "%s.layers_conv.%d.weight", prefix, 2 * i + 1);
        const float *wb = want(s, key, 3, C, C, 1);
        if (!wa || !wb) { free(h); free(g); free(cur); return s->err; }
        conv_stride_k(cur, B * Tcur, C, wa, 3, NULL, C, h);                 /* :96 conv k3 s3 */
        snprintf(key, sizeof key, "%s.layers_bn.%d", prefix, 2 * i);
        if (bn_named(s, key, h, rows2, C)) { free(h); free(g); free(cur); return s->err; }
        snprintf(tapname, sizeof tapname, "%s.level%d.pre", prefix, 2 * i + 1);
        emit(tap, user, tapname, h, B, Tcur / 3, C, 3);
        leaky(h, rows2 * C, 0.2f);
        conv_stride_k(h, rows2, C, wb, 1, NULL, C, g);                      /* :97 conv k1 */
        snprintf(key, sizeof key, "%s.layers_bn.%d", prefix, 2 * i + 1);
        if (bn_named(s, key, g, rows2, C)) { free(h); free(g); free(cur); return s->err; }
        snprintf(tapname, sizeof tapname, "%s.level%d.pre", prefix, 2 * i + 2);
        emit(tap, user, tapname, g, B, Tcur / 3, C, 3);
        leaky(g, rows2 * C, 0.2f);
        /* res = x[:, :, 1::3] (centre tap of each triple), :94 - or, causal and dilated (:92 with shift == pad),
         * the last tap; x = res + ..., :97 */
        const int64_t tapi = cfg->causal ? 2 : 1;
        for (int64_t r = 0; r < rows2; ++r)
            for (int64_t c = 0; c < C; ++c) g[r * C + c] = cur[(3 * r + tapi) * C + c] + g[r * C + c];
        free(h);
        free(cur);
        cur = g;
        Tcur /= 3;
    }
    snprintf(key, sizeof key, "%s.shrink.weight", prefix);
    const float *ws = want(s, key, 3, cfg->latent, C, 1);
    snprintf(key, sizeof key, "%s.shrink.bias", prefix);
    const float *bs = want(s, key, 1, cfg->latent, 0, 0);
    if (!ws || !bs) { free(cur); return s->err; }
    conv_stride_k(cur, B * Tcur, C, ws, 1, bs, cfg->latent, out);           /* :99 shrink */
    free(cur);
    emit(tap, user, prefix, out, B, cfg->latent, 0, 2);
    return 0;
}

/* nn.Conv1d(Cin, Cout, k) (stride 1, no padding) on channels-last x (B, T, Cin) -> y (B, T-k+1, Cout): the k input
 * frames of an output position are k*Cin contiguous floats, so each window is one GEMM whose rows overlap
 * (row stride Cin).  lib/model/rie.py:34,49-53 (the un-optimised constructor branch). */
static void conv_dense_k(const float *x, int64_t B, int64_t T, int64_t Cin, const float *w, int k, int64_t Cout, float *y) {
    const int64_t K = (int64_t)k * Cin, Tout = T - k + 1;
    float *W2 = (float *)malloc(sizeof(float) * (size_t)(Cout * K));
    for (int64_t o = 0; o < Cout; ++o)
        for (int64_t c = 0; c < Cin; ++c)
            for (int j = 0; j < k; ++j) W2[o * K + j * Cin + c] = w[(o * Cin + c) * k + j];
    for (int64_t b = 0; b < B; ++b) linear_rows(x + b * T * Cin, Tout, K, Cin, W2, NULL, Cout, y + b * Tout * Cout, Cout);
    free(W2);
}

/* TemporalBlock.forward (rie.py:85-105) with Optimize1f == False and dense == True: stride-1 convolutions of
 * 3, then 2*3^i + 1 taps; res = x[:, :, pad+shift : T-pad+shift] (:91-92) with pad = 3^i, shift = pad when causal. */
static int temporal_block_dense(store *s, const r3o_config *cfg, const char *prefix, const float *x,
                                int64_t B, int64_t T, int64_t Cin, float *out, r3o_tap_fn tap, void *user) {
    const int64_t C = cfg->channels;
    char key[256];
    int64_t Tcur = T - 2;
    float *cur = (float *)malloc(sizeof(float) * (size_t)(B * Tcur * C));
    snprintf(key, sizeof key, "%s.expand_conv.weight", prefix);
    const float *w = want(s, key, 3, C, Cin, 3);
    if (!w) { free(cur); return s->err; }
    conv_dense_k(x, B, T, Cin, w, 3, C, cur);                                /* :86 */
    snprintf(key, sizeof key, "%s.expand_bn", prefix);
    if (bn_named(s, key, cur, B * Tcur, C)) { free(cur); return s->err; }
    leaky(cur, B * Tcur * C, 0.2f);
    int64_t d = 3;
    for (int i = 0; i < cfg->num_levels - 1; ++i, d *= 3) {
        const int k = (int)(2 * d + 1);
        const int64_t Tn = Tcur - 2 * d, shift = cfg->causal ? d : 0;
        float *h = (float *)malloc(sizeof(float) * (size_t)(B * Tn * C));
        float *g = (float *)malloc(sizeof(float) * (size_t)(B * Tn * C));
        snprintf(key, sizeof key, "%s.layers_conv.%d.weight", prefix, 2 * i);
        const float *wa = want(s, key, 3, C, C, k);
        snprintf(key, sizeof key, "%s.layers_conv.%d.weight", prefix, 2 * i + 1);
        const float *wb = want(s, key, 3, C, C, 1);
        if (!wa || !wb) { free(h); free(g); free(cur); return s->err; }
        conv_dense_k(cur, B, Tcur, C, wa, k, C, h);                           /* :96 */
        snprintf(key, sizeof key, "%s.layers_bn.%d", prefix, 2 * i);
        if (bn_named(s, key, h, B * Tn, C)) { free(h); free(g); free(cur); return s->err; }
        leaky(h, B * Tn * C, 0.2f);
        conv_stride_k(h, B * Tn, C, wb, 1, NULL, C, g);                       /* :97 conv k1 */
        snprintf(key, sizeof key, "%s.layers_bn.%d", prefix, 2 * i + 1);
        if (bn_named(s, key, g, B * Tn, C)) { free(h); free(g); free(cur); return s->err; }
        leaky(g, B * Tn * C, 0.2f);
        for (int64_t b = 0; b < B; ++b)                                        /* :91-92, :97 */
            for (int64_t p = 0; p < Tn; ++p)
                for (int64_t c = 0; c < C; ++c) g[(b * Tn + p) * C + c] += cur[(b * Tcur + p + d + shift) * C + c];
        free(h);
        free(cur);
        cur = g;
        Tcur = Tn;
    }
    if (Tcur != 1) { free(cur); return fail(-1, "dense temporal block: %lld frames left, expected 1", (long long)Tcur); }
    snprintf(key, sizeof key, "%s.shrink.weight", prefix);
    const float *ws = want(s, key, 3, cfg->latent, C, 1);
    snprintf(key, sizeof key, "%s.shrink.bias", prefix);
    const float *bs = want(s, key, 1, cfg->latent, 0, 0);
    if (!ws || !bs) { free(cur); return s->err; }
    conv_stride_k(cur, B, C, ws, 1, bs, cfg->latent, out);                    /* :99 */
    free(cur);
    emit(tap, user, prefix, out, B, cfg->latent, 0, 2);
    return 0;
}

/* nn.Linear named `prefix` on (B, cin) -> (B, cout) */
static int linear_named(store *s, const char *prefix, const float *x, int64_t B, int64_t cin,
                        int64_t cout, float *y) {
    char key[256];
    snprintf(key, sizeof key, "%s.weight", prefix);
    const float *w = want(s, key, 2, cout, cin, 0);
    snprintf(key, sizeof key, "%s.bias", prefix);
    const float *b = want(s, key, 1, cout, 0, 0);
    if (!w || !b) return s->err;
    linear_rows(x, B, cin, cin, w, b, cout, y, cout);
    return 0;
}

/* FCBlock.forward, lib/model/rie.py:159-169, with Linear.forward :122-135 residual units. */
static int fc_block(store *s, const char *prefix, const float *x, int64_t B, int64_t cin,
                    int64_t cout, int nblocks, float *out, r3o_tap_fn tap, void *user) {
    char key[256];
    float *h = (float *)malloc(sizeof(float) * (size_t)(B * HIDDEN));
    float *y = (float *)malloc(sizeof(float) * (size_t)(B * HIDDEN));
    float *z = (float *)malloc(sizeof(float) * (size_t)(B * HIDDEN));
    int rc = 0;
    snprintf(key, sizeof key, "%s.fc_1", prefix);
    if ((rc = linear_named(s, key, x, B, cin, HIDDEN, h))) goto done;
    snprintf(key, sizeof key, "%s.bn_1", prefix);
    if ((rc = bn_named(s, key, h, B, HIDDEN))) goto done;
    leaky(h, B * HIDDEN, 0.2f);
    for (int n = 0; n < nblocks; ++n) {
        snprintf(key, sizeof key, "%s.layers.%d.w1", prefix, n);
        if ((rc = linear_named(s, key, h, B, HIDDEN, HIDDEN, y))) goto done;
        snprintf(key, sizeof key, "%s.layers.%d.batch_norm1", prefix, n);
        if ((rc = bn_named(s, key, y, B, HIDDEN))) goto done;
        leaky(y, B * HIDDEN, 0.2f);
        snprintf(key, sizeof key, "%s.layers.%d.w2", prefix, n);
        if ((rc = linear_named(s, key, y, B, HIDDEN, HIDDEN, z))) goto done;
        snprintf(key, sizeof key, "%s.layers.%d.batch_norm2", prefix, n);
        if ((rc = bn_named(s, key, z, B, HIDDEN))) goto done;
        leaky(z, B * HIDDEN, 0.2f);
        for (int64_t i = 0; i < B * HIDDEN; ++i) h[i] = h[i] + z[i];        /* out = x + y */
    }
    snprintf(key, sizeof key, "%s.fc_2", prefix);
    if ((rc = linear_named(s, key, h, B, HIDDEN, cout, out))) goto done;
    emit(tap, user, prefix, out, B, cout, 0, 2);
done:
    free(h); free(y); free(z);
    return rc;
}

/* Embedding.forward, lib/model/embedding.py:15-18 (LeakyReLU default slope 0.01) */
static int embedding(store *s, const char *prefix, const float *p, int64_t B, int64_t cin,
                     int64_t cout, float *out, r3o_tap_fn tap, void *user) {
    char key[256];
    float *m = (float *)malloc(sizeof(float) * (size_t)(B * EMB_MID));
    int rc = 0;
    snprintf(key, sizeof key, "%s.w1", prefix);
    if ((rc = linear_named(s, key, p, B, cin, EMB_MID, m))) goto done;
    snprintf(key, sizeof key, "%s.b1", prefix);
    if ((rc = bn_named(s, key, m, B, EMB_MID))) goto done;
    leaky(m, B * EMB_MID, 0.01f);
    snprintf(key, sizeof key, "%s.w2", prefix);
    if ((rc = linear_named(s, key, m, B, EMB_MID, cout, out))) goto done;
    snprintf(key, sizeof key, "%s.b2", prefix);
    if ((rc = bn_named(s, key, out, B, cout))) goto done;
    leaky(out, B * cout, 0.01f);
    emit(tap, user, prefix, out, B, cout, 0, 2);
done:
    free(m);
    return rc;
}

/* ------------------------------------------------------------------ grouping tables */

static const char *BRANCH[5] = {"Torso", "LArm", "RArm", "LLeg", "RLeg"};

/* joints of each branch; lib/model/rie.py:308-331 (F=3) and :334-357 (F=2) use the same joints */
static int group_joints(int J, int branch, int *joints) {
    static const int g17[5][5] = {{0, 7, 8, 9, 10}, {14, 15, 16}, {11, 12, 13}, {1, 2, 3}, {4, 5, 6}};
    static const int g15[5][5] = {{0, 1, 14}, {2, 3, 4}, {5, 6, 7}, {8, 9, 10}, {11, 12, 13}};
    static const int g14[5][5] = {{0, 7}, {8, 9, 10}, {11, 12, 13}, {4, 5, 6}, {1, 2, 3}};
    const int n = branch == 0 ? (J == 17 ? 5 : J == 15 ? 3 : 2) : 3;
    const int(*g)[5] = J == 17 ? g17 : J == 15 ? g15 : g14;
    for (int i = 0; i < n; ++i) joints[i] = g[branch][i];
    return n;
}

/* output slot -> (branch, index), lib/model/rie.py:426-431 */
static void output_order(int J, int *br, int *idx) {
    int s = 0;
#define PUT(b, i) do { br[s] = (b); idx[s] = (i); ++s; } while (0)
    if (J == 17) {
        PUT(0, 0);
        for (int i = 0; i < 3; ++i) PUT(3, i);
        for (int i = 0; i < 3; ++i) PUT(4, i);
        for (int i = 1; i < 5; ++i) PUT(0, i);
        for (int i = 0; i < 3; ++i) PUT(2, i);
        for (int i = 0; i < 3; ++i) PUT(1, i);
    } else if (J == 15) {
        PUT(0, 0); PUT(0, 1);
        for (int i = 0; i < 3; ++i) PUT(3, i);
        for (int i = 0; i < 3; ++i) PUT(4, i);
        for (int i = 0; i < 3; ++i) PUT(2, i);
        for (int i = 0; i < 3; ++i) PUT(1, i);
        PUT(0, 2);
    } else {
        PUT(0, 0);
        for (int i = 0; i < 3; ++i) PUT(3, i);
        for (int i = 0; i < 3; ++i) PUT(4, i);
        for (int i = 0; i < 3; ++i) PUT(2, i);
        for (int i = 0; i < 3; ++i) PUT(1, i);
        PUT(0, 1);
    }
#undef PUT
}

/* cat(x_g, diff_g, diff_t_g) for a joint group, channels-last (B*RF, 3*n*F).
 * diff   = x - x[root joint]        (same frame)            rie.py:301
 * diff_t = x - x[frame RF // F]     (same joint, Q1 quirk)  rie.py:304                 */
static float *encode_group(const float *x, int64_t B, int RF, int J, int F, const int *joints,
                           int n) {
    const int cin = 3 * n * F, tcur = RF / F;
    float *e = (float *)malloc(sizeof(float) * (size_t)(B * RF * cin));
    for (int64_t b = 0; b < B; ++b)
        for (int t = 0; t < RF; ++t) {
            const float *fr = x + ((b * RF + t) * J) * F;
            const float *fc = x + ((b * RF + tcur) * J) * F;
            float *o = e + (b * RF + t) * cin;
            for (int i = 0; i < n; ++i)
                for (int f = 0; f < F; ++f) {
                    const float v = fr[joints[i] * F + f];
                    o[i * F + f] = v;
                    o[n * F + i * F + f] = v - fr[f];
                    o[2 * n * F + i * F + f] = v - fc[joints[i] * F + f];
                }
        }
    return e;
}

/* ------------------------------------------------------------------ the two networks */

int r3o_forward(const r3o_config *cfg, const r3o_tensor *tensors, int ntensors, const float *x,
                const float *param, int64_t B, float *out, r3o_tap_fn tap, void *tap_user,
                int threads) {
    g_err[0] = 0;
    if (!cfg || (cfg->kind != 0 && cfg->kind != 1)) return fail(-3, "bad config");
    const int J = cfg->num_joints, F = cfg->in_features, L = cfg->num_levels;
    if ((J != 14 && J != 15 && J != 17) || (F != 2 && F != 3) || L < 1 || L > 8)
        return fail(-3, "unsupported J=%d F=%d levels=%d", J, F, L);
    const int emb = (cfg->extrinsic_dim > 0 && cfg->embed_dim > 0) ? cfg->embed_dim : 0;
    if (emb && !param) return fail(-3, "param is required when the camera embedding is on");
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    int RF = 1;
    for (int i = 0; i < L; ++i) RF *= 3;
    const int lat = cfg->latent, tcur = RF / F;
    store s = {tensors, ntensors, 0};
    int rc = 0;

    /* in_current = x[:, RF // F]  (rie.py:290-292) */
    float *cur = (float *)malloc(sizeof(float) * (size_t)(B * J * F));
    for (int64_t b = 0; b < B; ++b)
        memcpy(cur + b * J * F, x + ((b * RF + tcur) * J) * F, sizeof(float) * (size_t)(J * F));
    float *xglobal = (float *)malloc(sizeof(float) * (size_t)(B * lat));
    float *embv = emb ? (float *)malloc(sizeof(float) * (size_t)(B * emb)) : NULL;
    if ((rc = fc_block(&s, "GlobalInfo", cur, B, J * F, lat, 2, xglobal, tap, tap_user))) goto done0;
    if (emb && (rc = embedding(&s, "embedder", param, B, cfg->extrinsic_dim, emb, embv, tap, tap_user)))
        goto done0;

    if (cfg->kind == 1) {
        /* RIETrajectoryModel.forward, rie.py:518-559 */
        int joints[32];
        for (int j = 0; j < J; ++j) joints[j] = j;
        float *e = encode_group(x, B, RF, J, F, joints, J);
        float *local = (float *)malloc(sizeof(float) * (size_t)(B * lat));
        rc = temporal_block(&s, cfg, "LocalLayer", e, B, RF, 3 * J * F, local, tap, tap_user);
        free(e);
        if (!rc) {
            const int D = 2 * lat + emb;
            float *cat = (float *)malloc(sizeof(float) * (size_t)(B * D));
            for (int64_t b = 0; b < B; ++b) {
                memcpy(cat + b * D, local + b * lat, sizeof(float) * lat);
                memcpy(cat + b * D + lat, xglobal + b * lat, sizeof(float) * lat);
                if (emb) memcpy(cat + b * D + 2 * lat, embv + b * emb, sizeof(float) * emb);
            }
            rc = fc_block(&s, "Integration", cat, B, D, 3, 1, out, tap, tap_user);
            free(cat);
        }
        free(local);
        goto done0;
    }

    /* RIEModel.forward, rie.py:284-434 */
    {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)(B * 5 * lat));   /* (B,5,lat) :371 */
        float *mix = (float *)malloc(sizeof(float) * (size_t)(B * 5 * lat));
        float *br_out = (float *)malloc(sizeof(float) * (size_t)(B * lat));
        char name[64];
        for (int g = 0; g < 5 && !rc; ++g) {
            int joints[8];
            const int n = group_joints(J, g, joints);
            float *e = encode_group(x, B, RF, J, F, joints, n);
            snprintf(name, sizeof name, "LocalLayer_%s", BRANCH[g]);
            rc = temporal_block(&s, cfg, name, e, B, RF, 3 * n * F, br_out, tap, tap_user);
            free(e);
            for (int64_t b = 0; b < B; ++b)
                memcpy(tmp + (b * 5 + g) * lat, br_out + b * lat, sizeof(float) * lat);
        }
        if (!rc && cfg->stage != 1) {
            /* FuseBlocks[i](cat of the other four local features), rie.py:390-394 */
            float *others = (float *)malloc(sizeof(float) * (size_t)(B * 4 * lat));
            for (int i = 0; i < 5 && !rc; ++i) {
                for (int64_t b = 0; b < B; ++b) {
                    int k = 0;
                    for (int g = 0; g < 5; ++g)
                        if (g != i) memcpy(others + (b * 4 + k++) * lat, tmp + (b * 5 + g) * lat,
                                           sizeof(float) * lat);
                }
                snprintf(name, sizeof name, "FuseBlocks.%d", i);
                rc = fc_block(&s, name, others, B, 4 * lat, lat, 1, br_out, tap, tap_user);
                for (int64_t b = 0; b < B; ++b)
                    memcpy(mix + (b * 5 + i) * lat, br_out + b * lat, sizeof(float) * lat);
            }
            free(others);
        }
        if (!rc) {
            const int nfeat = cfg->stage == 1 ? 2 : 3;
            const int D = nfeat * lat + emb;
            float *cat = (float *)malloc(sizeof(float) * (size_t)(B * D));
            float *dec[5];
            int njo[5];
            for (int g = 0; g < 5; ++g) dec[g] = NULL;
            for (int g = 0; g < 5 && !rc; ++g) {
                int joints[8];
                njo[g] = group_joints(J, g, joints);
                for (int64_t b = 0; b < B; ++b) {               /* rie.py:376-407 */
                    float *c = cat + b * D;
                    memcpy(c, tmp + (b * 5 + g) * lat, sizeof(float) * lat);
                    c += lat;
                    if (cfg->stage != 1) { memcpy(c, mix + (b * 5 + g) * lat, sizeof(float) * lat); c += lat; }
                    memcpy(c, xglobal + b * lat, sizeof(float) * lat);
                    c += lat;
                    if (emb) memcpy(c, embv + b * emb, sizeof(float) * emb);
                }
                dec[g] = (float *)malloc(sizeof(float) * (size_t)(B * njo[g] * 3));
                snprintf(name, sizeof name, "Integration_%s", BRANCH[g]);
                rc = fc_block(&s, name, cat, B, D, njo[g] * 3, 1, dec[g], tap, tap_user);
            }
            if (!rc) {
                int br[32], idx[32];
                output_order(J, br, idx);                        /* rie.py:426-432 */
                for (int64_t b = 0; b < B; ++b)
                    for (int sl = 0; sl < J; ++sl)
                        memcpy(out + (b * J + sl) * 3, dec[br[sl]] + (b * njo[br[sl]] + idx[sl]) * 3,
                               sizeof(float) * 3);
            }
            for (int g = 0; g < 5; ++g) free(dec[g]);
            free(cat);
        }
        free(tmp); free(mix); free(br_out);
    }
done0:
    free(cur); free(xglobal); free(embv);
    return rc;
}

/* ------------------------------------------------------------------ camera */

static void mat3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = a;
        }
}
static void mat3_t(const double *A, double *T) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[j * 3 + i];
}
static void mat3_vec(const double *A, const double *v, double *o) {
    for (int i = 0; i < 3; ++i) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}

void r3o_camera_init(const double *K, const double *R, const double *t, r3o_camera *cam) {
    (void)K;
    double Rc2w[9], tmp[3];
    mat3_t(R, Rc2w);                                   /* Rc2w = Rw2c^T, camera.py:245 */
    mat3_vec(Rc2w, t, tmp);
    const double orig_w[3] = {-tmp[0], -tmp[1], -tmp[2]};   /* -R^T t, camera.py:279-285 */
    const double axis[3] = {Rc2w[2], Rc2w[5], Rc2w[8]};     /* Rc2w @ [0,0,1], camera.py:296-298 */
    const double len = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    cam->pitch = acos(axis[2] / (len * 1.0)) - M_PI / 2;    /* angle(axis, z) - pi/2, :308-316 */
    cam->height = orig_w[2];
    const double c = cos(cam->pitch), s = sin(cam->pitch);
    const double Rc2n[9] = {1, 0, 0, 0, c, s, 0, -s, c};    /* camera.py:333-338 */
    memcpy(cam->Rc2n, Rc2n, sizeof Rc2n);
    cam->Tc2n[0] = 0; cam->Tc2n[1] = -orig_w[2]; cam->Tc2n[2] = 0;   /* camera.py:340-343 */
    mat3_mul(cam->Rc2n, R, cam->Rw2n);                      /* Rw2n = Rc2n @ Rw2c, :255 */
    mat3_vec(cam->Rc2n, t, tmp);
    for (int i = 0; i < 3; ++i) cam->Tw2n[i] = tmp[i] + cam->Tc2n[i];   /* :256 */
    double Rn2c[9];
    mat3_t(cam->Rc2n, Rn2c);
    mat3_mul(Rc2w, Rn2c, cam->Rn2w);                        /* Rn2w = Rc2w @ Rn2c, :258 */
    double a[3], b[3];
    mat3_vec(cam->Rn2w, cam->Tc2n, a);
    mat3_vec(Rc2w, t, b);
    for (int i = 0; i < 3; ++i) cam->Tn2w[i] = -a[i] - b[i];   /* :259 */
}

void r3o_rays_from_uv(const double *K, const r3o_camera *cam, const double *uv, int64_t n,
                      double *rays) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int64_t i = 0; i < n; ++i) {
        const double p[3] = {(uv[2 * i] - cx) / fx, (uv[2 * i + 1] - cy) / fy, 1.0};   /* :438-439 */
        mat3_vec(cam->Rc2n, p, rays + 3 * i);               /* pt_cam @ Rc2n.T, :471 */
    }
}

void r3o_uv_from_rays(const double *K, const r3o_camera *cam, const double *rays, int64_t n,
                      double *uv) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double Rn2c[9], p[3];
    mat3_t(cam->Rc2n, Rn2c);
    for (int64_t i = 0; i < n; ++i) {
        mat3_vec(Rn2c, rays + 3 * i, p);                    /* pt @ Rn2c.T, :479 */
        uv[2 * i] = p[0] * fx + cx;                         /* :455-456 */
        uv[2 * i + 1] = p[1] * fy + cy;
    }
}

void r3o_transform(const double *R, const double *T, const double *pts, int64_t n, double *out) {
    for (int64_t i = 0; i < n; ++i) {
        double o[3];
        mat3_vec(R, pts + 3 * i, o);
        for (int k = 0; k < 3; ++k) out[3 * i + k] = o[k] + T[k];
    }
}

void r3o_distort_points(const double *K, const double *d, const double *uv, int64_t n, double *out) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
    for (int64_t i = 0; i < n; ++i) {
        const double x = (uv[2 * i] - cx) / fx, y = (uv[2 * i + 1] - cy) / fy;
        const double r2 = x * x + y * y;
        const double rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
        const double xd = x * rad + (2 * p1 * x * y + p2 * (r2 + 2 * x * x));
        const double yd = y * rad + (p1 * (r2 + 2 * y * y) + 2 * p2 * x * y);
        out[2 * i] = xd * fx + cx;
        out[2 * i + 1] = yd * fy + cy;
    }
}

void r3o_undistort_points(const double *K, const double *d, const double *uv, int64_t n,
                          double *out) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
    for (int64_t i = 0; i < n; ++i) {
        const double x0 = (uv[2 * i] - cx) / fx, y0 = (uv[2 * i + 1] - cy) / fy;
        double x = x0, y = y0;
        for (int it = 0; it < 5; ++it) {
            const double r2 = x * x + y * y;
            const double icd = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - dx) * icd;
            y = (y0 - dy) * icd;
        }
        out[2 * i] = x * fx + cx;          /* re-projection with P = K */
        out[2 * i + 1] = y * fy + cy;
    }
}
