/*
 * ray3d_oracle.h - CPU restatement of the Ray3D lifting forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product path (ray3d_amd + libray3d_hip.so) never
 * links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle.py against the
 * golden fixtures under tests/golden/ that tests/golden/make_golden.py produced by importing the
 * reference (YxZhxn/Ray3D) on CPU.  The one exception is r3o_undistort_points (OpenCV's
 * cv2.undistortPoints is a third-party dependency absent from the reference tree and from this
 * image): that function is "parity unpinned" and only checked by round-trip properties.
 */
#ifndef RAY3D_ORACLE_H
#define RAY3D_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One state_dict tensor as the reference names it. float32 data, row-major;
 * int64 tensors (num_batches_tracked) may be passed but are ignored. */
typedef struct {
    const char *key;
    const float *data;
    int rank;
    int64_t shape[4];
} r3o_tensor;

typedef struct {
    int kind;          /* 0 = RIEModel (pos), 1 = RIETrajectoryModel (trj) */
    int num_joints;    /* 14 | 15 | 17 */
    int in_features;   /* 2 | 3 */
    int num_levels;    /* len(filter_widths); every width is 3 */
    int channels;      /* CHANNELS */
    int latent;        /* LATENT_FEATURES_DIM */
    int stage;         /* STAGE (pos only) */
    int extrinsic_dim; /* 0 when CAMERA_EMBDDING is False */
    int embed_dim;     /* 0 when CAMERA_EMBDDING is False */
    int causal;        /* CAUSAL (with the dilated convolutions, the only pairing the reference runs) */
    int dense;         /* DENSE with DISABLE_OPTIMIZATIONS: (2*pad+1)-tap stride-1 convolutions, rie.py:49-53 */
} r3o_config;

/* Tap sink: called with intermediate tensors (float32, row-major).  Names:
 *   "<Block>"                    output of a sub-module, shape (B, dim)
 *   "<LocalLayer>.level<i>.pre"  pre-activation BatchNorm output of conv i of that TemporalBlock
 *                                (i = 0 expand, then layers_bn order), channels-last (B, T, C)
 */
typedef void (*r3o_tap_fn)(const char *name, const float *data, const int64_t *shape, int rank,
                           void *user);

/* RIEModel.forward / RIETrajectoryModel.forward in eval mode.
 *   x     (B, RF, J, F) float32       param (B, extrinsic_dim) float32 (may be NULL if dim 0)
 *   out   pos: (B, 1, J, 3)           trj: (B, 1, 1, 3)
 * Returns 0, or a negative code (-1 missing tensor, -2 bad shape, -3 bad config); the message is
 * available from r3o_last_error().  threads <= 0 means "all OpenMP threads". */
int r3o_forward(const r3o_config *cfg, const r3o_tensor *tensors, int ntensors,
                const float *x, const float *param, int64_t B, float *out,
                r3o_tap_fn tap, void *tap_user, int threads);

const char *r3o_last_error(void);

/* ---- camera (lib/camera/camera.py), float64 like the reference's NumPy code ---- */

typedef struct {
    double height;      /* (-R^T t)[2]                                   camera.py:279-285 */
    double pitch;       /* acos(axis_w.z/|axis_w|) - pi/2                camera.py:308-316 */
    double Rc2n[9], Tc2n[3];   /* camera.py:325-345 */
    double Rw2n[9], Tw2n[3];   /* camera.py:255-256 */
    double Rn2w[9], Tn2w[3];   /* camera.py:258-259 */
} r3o_camera;

/* K (3x3), R (3x3, world->camera), t (3) row-major doubles. */
void r3o_camera_init(const double *K, const double *R, const double *t, r3o_camera *cam);

/* get_cam_ray_given_uv with undistort=False: uv (n,2) pixels -> rays (n,3).  camera.py:423-471 */
void r3o_rays_from_uv(const double *K, const r3o_camera *cam, const double *uv, int64_t n,
                      double *rays);
/* get_uv_given_cam_ray (camera.py:473-483) */
void r3o_uv_from_rays(const double *K, const r3o_camera *cam, const double *rays, int64_t n,
                      double *uv);
/* pt @ R^T + T^T for (n,3) points; used for world2normalized / normalized2world */
void r3o_transform(const double *R, const double *T, const double *pts, int64_t n, double *out);

/* PARITY UNPINNED: inverse Brown-Conrady distortion as documented for cv2.undistortPoints
 * (opencv-python==4.4.0.42, requirements.txt:40; call site camera.py:420) with P=K:
 * 5 fixed-point iterations, coefficients (k1,k2,p1,p2,k3). */
void r3o_undistort_points(const double *K, const double *dist5, const double *uv, int64_t n,
                          double *out);
/* forward distortion model used for round-trip checks (data/camera_augmentation.py:502-542) */
void r3o_distort_points(const double *K, const double *dist5, const double *uv, int64_t n,
                        double *out);

/* metrics (lib/loss/loss.py) are restated in numpy: oracle/metrics_oracle.py */

#ifdef __cplusplus
}
#endif
#endif
