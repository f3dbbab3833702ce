/*
 * ray3d_hip.h - C ABI of libray3d_hip.so: the MI355X (gfx950) implementation of Ray3D's
 * 2D->3D lifting forward pass.
 *
 * The reference (YxZhxn/Ray3D) has no FFI for this path; its seam is the Python nn.Module
 * contract.  Each entry point below names the reference interface it
 * replaces; INTEGRATION.md shows the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions: plain C, int status codes (0 = ok, <0 = error, text via r3d_last_error()),
 * no exceptions cross the boundary.  All *_dev pointers are device (HBM) pointers owned by the
 * caller; the library owns only its packed weights.  A handle is bound to the HIP device that
 * was current at r3d_finalize().  Threading: a handle - and a (pos, trj) pair used together - must
 * be driven by one thread at a time (its launch plans, per-batch-size tile schedules and profiling
 * records are unguarded caches); different handles are independent and may be used concurrently
 * (the registry that maps handle pairs to plans is mutex-guarded, r3d_last_error() is thread-local; the ordering of whole-device
 * forwards - two of them must never share the chip - and the launch it protects are one critical section across threads).
 * Every call enqueues on the given hipStream_t (passed as void*) and returns without syncing.
 * The first forward of a new batch size builds and uploads a tile schedule (hipMalloc + blocking
 * hipMemcpy): call r3d_prepare() for that size beforehand when the forward is to be captured into
 * a hipGraph or must not stall.
 */
#ifndef RAY3D_HIP_H
#define RAY3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3D_OK 0
#define R3D_ERR_ARG (-1)        /* bad argument / unsupported configuration            */
#define R3D_ERR_KEY (-2)        /* unknown, duplicate or missing state_dict key         */
#define R3D_ERR_SHAPE (-3)      /* tensor shape does not match the configuration         */
#define R3D_ERR_STATE (-4)      /* call order violated (e.g. forward before finalize)    */
#define R3D_ERR_HIP (-5)        /* a HIP runtime call failed                             */
#define R3D_ERR_WORKSPACE (-6)  /* workspace too small                                   */
#define R3D_ERR_ABORTED (-7)    /* r3d_status: a forward gave up waiting for its own tiles: its outputs are NaN */

#define R3D_KIND_POS 0 /* lib/model/rie.py:172  RIEModel            -> (B,1,J,3) */
#define R3D_KIND_TRJ 1 /* lib/model/rie.py:437  RIETrajectoryModel  -> (B,1,1,3) */

/* Mirrors the constructor arguments the reference factory passes
 * (lib/model/__init__.py:23-46 -> lib/model/rie.py:178-181 / :443-446). */
#define R3D_ABI_VERSION 6 /* bumped whenever a struct below changes size or layout: r3d_abi_version() returns the
                           * library's; a binding compares it with the header it was written against       */

typedef struct {
    int32_t struct_size;   /* sizeof(r3d_config) of the CALLER's header: r3d_create rejects any other size, so a
                            * binding built against an older (shorter) struct fails loudly instead of having
                            * r3d_create read past its buffer                                             */
    int32_t kind;          /* R3D_KIND_POS | R3D_KIND_TRJ                               */
    int32_t num_joints;    /* NUM_KPTS: 14, 15 or 17                                    */
    int32_t in_features;   /* INPUT_DIM: 3 (rays) or 2                                  */
    int32_t num_levels;    /* len(ARCHITECTURE); every filter width must be 3           */
    int32_t channels;      /* CHANNELS (multiple of 32)                                 */
    int32_t latent;        /* LATENT_FEATURES_DIM (multiple of 32)                      */
    int32_t stage;         /* STAGE (pos only; 1 = no FuseBlocks)                       */
    int32_t extrinsic_dim; /* EXTRINSIC_DIM, or 0 when CAMERA_EMBDDING is False         */
    int32_t embed_dim;     /* EMBEDD_DIM (multiple of 32), or 0 when CAMERA_EMBDDING off */
    int32_t causal;        /* CAUSAL with DISABLE_OPTIMIZATIONS: each level's residual is the
                            * last of its three input frames instead of the centre one
                            * (rie.py:43-47,88-92).  CAUSAL with the strided convolutions is
                            * not a configuration the reference can run (rie.py:94-97 raises);
                            * DISABLE_OPTIMIZATIONS alone computes the same function for an
                            * RF-long window and needs no flag.                             */
    int32_t dense;         /* DENSE with DISABLE_OPTIMIZATIONS (the dense-convolution ablation,
                            * rie.py:49-53): level i's convolution has 2*3^i + 1 taps, stride 1,
                            * and every level is evaluated at all RF positions of a window
                            * (cost grows with RF^2: meant for the short receptive fields the
                            * ablation is run at).  DENSE without DISABLE_OPTIMIZATIONS is
                            * ignored by the reference constructor (:54-55): pass 0.          */
    int32_t bf16x3;        /* 0: fp32 matrix cores (v_mfma_f32_32x32x2_f32).  1: the large GEMMs on the
                            * bf16 matrix cores with every fp32 operand split exactly into three bf16
                            * terms (six products, fp32 accumulate): fp32-equivalent results - 1.1 to 1.4
                            * times the fp32 path's error against a float64 evaluation, both ~1e-6 of the
                            * output magnitude - at 6/16 of the matrix time.  Not a
                            * reference key.  The environment variable R3D_BF16X3=1 / =0 at r3d_create
                            * overrides the field.                                                     */
} r3d_config;

typedef struct r3d_model r3d_model;

/* ---- construction: replaces RIEModel(...)/RIETrajectoryModel(...) + load_state_dict ---- */

/* lib/model/rie.py:178-253 / :443-494 (module construction). */
int r3d_create(const r3d_config *cfg, r3d_model **out);
int r3d_destroy(r3d_model *m);

/* The float tensors of the reference state_dict this configuration expects (SURVEY.md A.4;
 * int64 num_batches_tracked buffers are not part of the ABI).  Lets a binding enumerate keys. */
int r3d_num_weights(const r3d_model *m);
const char *r3d_weight_key(const r3d_model *m, int index);
int r3d_weight_shape(const r3d_model *m, int index, int64_t shape[4], int *rank);

/* nn.Module.load_state_dict(strict=True) equivalent (lib/train_val/trainer.py:161-164,
 * lib/utils/utils.py:208-218, but loud: unknown key -> R3D_ERR_KEY, wrong shape ->
 * R3D_ERR_SHAPE).  `host` is float32 row-major in torch layout (Conv1d (Cout,Cin,k),
 * Linear (out,in)); it is copied.  A leading "module." (nn.DataParallel checkpoints,
 * lib/model/__init__.py:52) is stripped.  May be called again after finalize to update. */
int r3d_set_weight(r3d_model *m, const char *key, const float *host, const int64_t *shape, int rank);

/* model.eval() + device placement: folds eval-mode BatchNorm (eps 1e-5) into the preceding
 * Conv1d/Linear in float64, repacks every layer into the GEMM layout the kernels read and
 * uploads to the current HIP device.  R3D_ERR_KEY (message lists the first missing key) if any
 * tensor was not set. */
int r3d_finalize(r3d_model *m);

/* ---- forward: replaces pos_model(inputs_2d, inputs_param) [+ trj_model(...)] ---- */

#define R3D_INPUT_RAYS 0 /* x is what the reference feeds the model: ray-encoded keypoints  */
#define R3D_INPUT_UV 1   /* x is pixel keypoints; rays are computed on the fly from `cam`   */

typedef struct {
    int32_t mode;          /* R3D_INPUT_RAYS | R3D_INPUT_UV                                  */
    const float *x_dev;    /* RAYS: float32 (frames, J, F);  UV: float32 (frames, J, 2)      */
    int64_t window_stride; /* frames between the starts of consecutive windows:
                              RF for a (B,RF,J,F) batch (lib/train_val/trainer.py:47-58 output),
                              1 to slide over an edge-padded clip in place (replaces
                              eval_data_prepare: window i = frames [i, i+RF))                */
    const float *param_dev;/* float32 [height, pitch] rows (lib/train_val/trainer.py:297);
                              ignored (may be NULL) when the camera embedding is off         */
    int64_t param_stride;  /* floats between consecutive windows' rows: extrinsic_dim for a
                              (B,E) tensor, 0 to broadcast one row to every window           */
    const double *cam_dev; /* UV mode only: float64 rows {fx, fy, cx, cy, cos(pitch),
                              sin(pitch), 0, 0} (lib/camera/camera.py:423-471)               */
    int64_t cam_stride;    /* doubles between consecutive windows' rows: 8, or 0 = broadcast */
} r3d_input;

/* Bytes of scratch HBM that suffice for every forward of AT MOST B windows (either model may be NULL): the maximum over
 * the launch plans calls of 1..B windows can select (small calls run less fused plans with larger intermediates), so a
 * caller may size its workspace once for its largest batch.
 * The workspace is scratch: nothing in it has to survive between calls, and the caller may use it for something else
 * between them.  (What a forward needs ACROSS calls - the ready counters and the bound problem table of the single-launch
 * forward, and for calls of <= 16 windows two banks of activations - lives in device memory the library owns, per
 * cached batch size; a forward that is being captured into a hipGraph keeps all of it inside the workspace instead, so
 * a graph's workspace must stay allocated, and untouched by other work while the graph runs, as any captured buffer.) */
size_t r3d_workspace_bytes(const r3d_model *pos, const r3d_model *trj, int64_t B);

/* Everything a forward of B windows needs besides its arguments - the launch plan of the pair and the tile schedule
 * of this batch size, uploaded - so that the forward itself only enqueues kernels (hipGraph capture, latency).
 * Either model may be NULL.
 * Lifetime rule: the library caches the tile schedules (device memory) of the 64 most recently used batch sizes per
 * pair and frees the least recently used one beyond that - EXCEPT sizes named in r3d_prepare, which stay resident
 * until r3d_release (or until either model is destroyed): a hipGraph that captured a forward holds pointers into its
 * size's schedule and never calls the library again, so prepare every size you capture and release it only after the
 * graph is destroyed.  Never call r3d_prepare / a first forward of a new size while a stream is capturing.
 * A captured forward also holds the handle's status word (pinned host memory, freed by r3d_destroy): destroy every
 * graph that captured a forward of a handle BEFORE the handle - a replay after r3d_destroy writes to freed memory. */
int r3d_prepare(r3d_model *pos, r3d_model *trj, int64_t B);
int r3d_release(r3d_model *pos, r3d_model *trj, int64_t B);   /* un-pins the size; R3D_ERR_ARG if it was never prepared */

/* One network, exactly the reference module's forward:
 *   pos: out_dev (B,1,J,3)   lib/model/rie.py:284-434
 *   trj: out_dev (B,1,1,3)   lib/model/rie.py:518-559 */
int r3d_forward(r3d_model *m, const r3d_input *in, int64_t B, float *out_dev,
                void *workspace_dev, size_t workspace_bytes, void *hip_stream);

/* Both networks in one pass over the input: out_dev (B,1,J,3) = pos + trj broadcast over
 * joints (lib/train_val/trainer.py:337,346,353); out_trj_dev (B,1,1,3) optional (may be NULL). */
int r3d_forward_pair(r3d_model *pos, r3d_model *trj, const r3d_input *in, int64_t B,
                     float *out_dev, float *out_trj_dev, void *workspace_dev,
                     size_t workspace_bytes, void *hip_stream);

/* ---- errors of a forward that surface on the device; per-handle options ---- */

/* The reference's seam reports errors as Python exceptions.  A forward is asynchronous, so what can only
 * be found out on the device is reported here: r3d_status synchronises `hip_stream` and returns R3D_ERR_ABORTED when a
 * forward of this handle (for a pair: ask the pos handle) since the last call gave up waiting for its own tiles - the
 * single-launch forward needs all its workgroups resident, which another process's persistent kernel on the same GPU or
 * a CU mask can prevent; such a forward ends after the spin timeout with NaN outputs, never hangs.  The flag is cleared
 * by the call.  Remedy: R3D_OPT_STAGED (the Python mirror's checked entry points do exactly that, once, before raising). */
int r3d_status(r3d_model *m, void *hip_stream);

#define R3D_OPT_STAGED 1          /* value != 0: this handle's forwards run as one launch per level of the network (no
                                   * co-residency assumption; a few percent slower) instead of one persistent launch    */
#define R3D_OPT_SPIN_TIMEOUT_MS 2 /* how long a tile of the single-launch forward waits for its producers before the
                                   * forward gives up (default 1000)                                                    */
#define R3D_OPT_CU_LIMIT 3        /* value = n > 0: this handle's forwards are launched on a CU-masked stream that can use n CUs
                                   * (hipExtStreamCreateWithCUMask): the single-launch forward uses at most n workgroups and is
                                   * NOT ordered against masked forwards of OTHER streams (two half-chip forwards side by side:
                                   * DESIGN.md 5.2).  The caller guarantees:
                                   *  - streams used concurrently have DISJOINT masks;
                                   *  - a mask enables at least ceil(n / 8) CUs in EVERY XCD (workgroups are dealt round-robin to
                                   *    the eight XCDs: n enabled CUs anywhere are not enough for n co-resident workgroups);
                                   *  - one handle (pair) per masked stream: a handle has ONE control region; used on a second
                                   *    masked stream its forwards are ordered behind those on the first one.
                                   * The library orders whole-device forwards (n = 0 handles) behind every masked forward issued
                                   * before them and masked forwards behind the last whole-device forward, so the two kinds never
                                   * share the chip.  A violated guarantee ends in the bounded spin (R3D_OPT_SPIN_TIMEOUT_MS), NaN
                                   * outputs and R3D_ERR_ABORTED from r3d_status - never in a hang.  0 (default): the whole device.
                                   * Changing the value waits for the handle's device and drops its cached tile schedules; it
                                   * fails with R3D_ERR_STATE while the handle has prepared (pinned) schedules - r3d_release them
                                   * first.  For a pair set it on both handles.                                              */
#define R3D_OPT_LANES 4           /* value = n in {2, 4} (0 / 1: off): the LIBRARY creates n CU-masked streams on the handle's device -
                                   * lane k: the CUs c of every XCD with c % n == k, so every lane spans all eight XCDs - each with
                                   * its own tile schedules and control regions; the packed weights stay ONE image per handle.  n
                                   * independent forwards then share the chip side by side (a level that holds 192 - 224 tiles
                                   * leaves a quarter of 256 CUs idle and runs as two full rounds on 128): the throughput mode for
                                   * callers with independent batches in flight - the clip evaluation has 240 clips
                                   * (lib/train_val/trainer.py:295-353).  How a forward finds its lane:
                                   *  - `stream` IS a lane's stream (r3d_lane_stream): it runs there, in order with whatever else the
                                   *    caller enqueues on that stream (the metrics of the clip, ...);
                                   *  - any other stream: lanes are served round-robin; the lane's stream waits for everything the
                                   *    caller's stream holds so far, runs the forward, and the caller's stream sees the outputs
                                   *    after r3d_lanes_join(m, stream) - NOT at return as without lanes.
                                   *    Until that join the forward's inputs, workspace and outputs belong to the lane: the
                                   *    caller's stream must not overwrite or free them (it is not ordered behind the lane).
                                   *    (a caller on the LEGACY DEFAULT stream: the lanes' streams are blocking streams - they are
                                   *    behind the default stream's work without an event, and none is recorded there, because an
                                   *    event on the default stream is behind every blocking stream's work: the lanes would take
                                   *    turns.  For the same reason any work issued on the default stream between two lanes'
                                   *    forwards serialises them - drive a lane loop from a stream of your own.)
                                   * One workspace per lane in flight (the caller's, as always).  Set it on both handles of a pair,
                                   * after r3d_finalize; it waits for the device, drops cached schedules and fails with
                                   * R3D_ERR_STATE while prepared (pinned) schedules exist.  r3d_prepare prepares every lane.
                                   * hipGraphs: a forward issued on a lane's OWN stream can be captured there (after r3d_prepare); a
                                   * forward on a capturing stream that is no lane's would have to be relayed inside the capture
                                   * and is refused with R3D_ERR_STATE.
                                   * Abort contract as without lanes: a lane's forward that cannot get its workgroups resident
                                   * ends in NaN outputs and R3D_ERR_ABORTED from r3d_status (which waits for the lanes too).    */
int r3d_set_option(r3d_model *m, int32_t option, int64_t value);

/* R3D_OPT_LANES: the stream of lane `lane` (0 .. n - 1) of the handle (for a pair: the pos handle's lanes are the pair's) - a
 * hipStream_t the library owns; replaces nothing in the reference (its evaluation loop is sequential: trainer.py:295-353). */
int r3d_lane_stream(r3d_model *m, int32_t lane, void **stream);
/* ... and: make `stream` wait (device-side) for every forward that was relayed to a lane and not joined yet by the stream that
 * issued it.  Afterwards the forwards `stream` itself issued count as joined; those of other streams still wait for THEIR join. */
int r3d_lanes_join(r3d_model *m, void *stream);

/* ---- instrumentation (bench.py / tests) ---- */

/* When enabled, the next forward brackets every kernel launch with hipEvents on the launch
 * stream; r3d_profile_read then synchronises and returns per-launch records.  The first record
 * ("r3d_event_pair", stage -1) is an empty bracket: the cost of the two event records themselves,
 * which every other record's `ms` includes. */
typedef struct {
    char kernel[48];  /* kernel family name as rocprofv3 shows it (prefix)                  */
    int32_t stage;    /* position in the launch sequence                                     */
    int32_t blocks;   /* workgroups launched                                                 */
    float ms;         /* elapsed milliseconds between the bracketing events                  */
    double flops;     /* algorithmic FLOPs (2*M*K*N over the reference's layers) in launch   */
    double bytes;     /* algorithmic HBM bytes (operands read once + result written once)    */
} r3d_launch_record;
int r3d_profile_enable(r3d_model *m, int on);
int r3d_profile_read(r3d_model *m, r3d_launch_record *records, int capacity);

/* The shader clock the handle's last single-launch forward ran at, in GHz (for a pair: ask the pos handle): the kernel's
 * first workgroup stamps its cycle counter and the 100 MHz wall clock at both ends.  Synchronises `hip_stream`.  The
 * first forwards after an idle period run below the clock a busy chip settles at, so a roofline wants it
 * next to the rate (north_star: counters against the gfx950 peak - there is no reference counterpart).  *ghz = 0 when the last forward ran level by level (R3D_OPT_STAGED, plans the single launch
 * cannot hold) or none ran yet. */
int r3d_last_clock(r3d_model *m, void *hip_stream, double *ghz);

/* ---- per-clip error sums: the host side of Trainer.evaluate_core after the forward ---- */

/* lib/train_val/trainer.py:355-397 for one clip: prediction and ground truth (float32, (n_frames, J, 3), normalised
 * frame, device memory) go to world coordinates in float64 (camera.py:401-410: p @ Rn2w^T + Tn2w^T; `rn2w` is the
 * row-major 3x3 Rn2w, `tn2w` its translation, host pointers), then
 *   out[R3D_METRIC_MPJPE]    = sum over frames of mean_j |pred - gt|                (loss.py:12-18,  trainer.py:386)
 *   out[R3D_METRIC_PMPJPE]   = ... after the per-frame similarity (Procrustes) fit  (loss.py:30-69,  :393)
 *   out[R3D_METRIC_NMPJPE]   = ... after the per-frame scale fit                    (loss.py:72-82,  :388)
 *   out[R3D_METRIC_VELOCITY] = n_frames * mean |first difference of the error|      (loss.py:95-104, :395; NaN if n < 2)
 *   out[R3D_METRIC_ROOT]     = sum over frames of |pred - gt| of joint 0                             (:387)
 * i.e. the clip's contribution to each epoch_loss_* accumulator, in metres.  `out_dev` is device memory of
 * R3D_METRIC_OUT_DOUBLES doubles: the five sums first, the rest is scratch for the workgroups' partial sums.
 * Deterministic (fixed summation order); enqueued on `stream`, no synchronisation. */
#define R3D_METRIC_MPJPE 0
#define R3D_METRIC_PMPJPE 1
#define R3D_METRIC_NMPJPE 2
#define R3D_METRIC_VELOCITY 3
#define R3D_METRIC_ROOT 4
#define R3D_METRIC_COUNT 5
#define R3D_METRIC_MAX_BLOCKS 128
#define R3D_METRIC_OUT_DOUBLES (R3D_METRIC_COUNT * (1 + R3D_METRIC_MAX_BLOCKS))
int r3d_clip_metrics(const float *pred_dev, const float *gt_dev, int64_t n_frames, int32_t num_joints,
                     const double *rn2w, const double *tn2w, double *out_dev, void *stream);

const char *r3d_last_error(void);
const char *r3d_version(void);
int r3d_abi_version(void);                 /* R3D_ABI_VERSION the library was built with */
/* The arithmetic the handle's large GEMMs actually run in: 0 = fp32 matrix cores, 1 = bf16x3 (r3d_config.bf16x3, or
 * the R3D_BF16X3 environment override read at r3d_create).  bench.py labels its line with this, not with the field. */
int r3d_precision(const r3d_model *m);

/* ---- test hooks: ONLY in libray3d_hip_hooks.so (the same sources built with -DR3D_TEST_HOOKS; tests/test_host.py, host
 * only, no device needed).  That build also reads the development switches (plan / tile-kind A/Bs, schedule dumps, fault
 * injection) from the environment; the product library has neither. ---- */
#ifdef R3D_TEST_HOOKS

/* Builds the static tile schedule of ONE launch for `nprob` GEMM problems (rows M[i], columns N[i], nk[i] K tiles of
 * 32, largest split-K factor max_ks[i], cap on 32-row units per tile max_units[i]) on `nwg` workgroups and verifies
 * that the tiles cover every (32-row unit, 32-column granule) exactly once within the kernel's tile-shape rules.
 * Returns 0, or a negative code naming the first violated rule. */
int r3d_debug_schedule_check(int nprob, const int *M, const int *N, const int *nk, const int *max_ks,
                             const int *max_units, int nwg, int enc, int *out_grid, int *out_tiles,
                             double *out_imbalance);

/* The whole forward's tile lists for `batch` windows on `nwg` CUs: every cell of every problem computed exactly once
 * over all launches, every consumer's launch after all of its producers' tiles.  *spilled = first-level rows that
 * run one launch late (row spill).  Returns 0 or a negative code. */
int r3d_debug_plan_check(r3d_model *pos, r3d_model *trj, int64_t batch, int nwg, int *launches, int *spilled);

/* The single-launch form of the forward (one persistent kernel for all levels, tiles ordered by ready counters), built
 * and executed on the host as a dependency machine: every tile gets to run, every counter ends full, and whenever a tile
 * runs every earlier problem that touches the same buffer columns is complete for the tile's windows.  Returns 0, 1 when
 * the plan of this batch size runs launch by launch (nothing to check), or a negative code. */
int r3d_debug_forward_check(r3d_model *pos, r3d_model *trj, int64_t batch, int nwg, int *tiles, int *counters);
#endif /* R3D_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif
