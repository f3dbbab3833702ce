// Internal declarations of libray3d_hip.so (not part of the ABI; see include/ray3d_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "ray3d_hip.h"

namespace r3d {

constexpr int BK = 32;        // K tile of the GEMM kernels; every packed K is a multiple of it
constexpr int MAX_SEG = 4;    // K segments of an A operand (concatenations are never materialised)
constexpr int MAX_PROB = 12;  // GEMM problems grouped in one launch
constexpr int N_ALIGN = 256;  // packed weight rows are padded to the GEMM column-block width
constexpr int MLP_HIDDEN = 1024;
constexpr int EMBED_MID = 32;
constexpr int CUR_LD = 64;    // padded row length of the "current frame" matrix (J*F <= 51)

// ------------------------------------------------------------------ device-visible PODs

struct GemmProb {
    const float *a[MAX_SEG];  // A operand segments, row-major, row r at a[s] + r*lda[s]
    int lda[MAX_SEG];
    int kend[MAX_SEG];        // cumulative end (in K) of each segment; last used == K
    const float *w;           // packed weights [Npad][K] (BN folded), row n = output channel
    const float *bias;        // [Npad]
    const float *res;         // optional residual, added AFTER the activation
    float *c;
    int ldr, ldc;
    int M, N, K;
    float slope;              // LeakyReLU slope, 1.0f = linear layer
    // --- fused pair (w2 != nullptr): C = res + lrelu2(lrelu(A W^T + b) W2^T + b2), the 1x1 convolution of a
    // TemporalBlock level applied to the 3-tap one's output without leaving the CU (N <= 256, K2 == N) ---
    const float *w2;          // packed weights of the second layer [256][K2]
    const float *bias2;
    int K2;
    float slope2;
    // --- fp32 on the bf16 matrix cores (wb3 != nullptr): the same fp32 weights in bf16-MFMA operand order ---
    const float *wb3;
    const float *w2b3, *w3b3;  // fused first level on the bf16 matrix cores: the b3 copies of w2 / w3 (all three set or none)
    // --- first level of the pyramid in one tile (w3 != nullptr; lut != nullptr): w/bias = expand_conv on the gathered
    // input (three rows per output row), w2/bias2 = the level's 3-tap convolution, w3/bias3 = its 1x1 convolution;
    // the residual is the centre one of the three expand_conv rows.  M counts OUTPUT rows. ---
    const float *w3;
    const float *bias3;
    int K3;
    float slope3;
    // --- fused feature-encoding prologue (first layers only; lut == nullptr otherwise) ---
    // A[row][col] is computed on the fly from the raw input instead of being read from memory:
    // row = window * enc_rows + t3 covers input frames 3*t3 .. 3*t3+2 of that window.
    const int *lut;           // two tables: element byte offsets [K], chunk flags [K/4] (layout below)
    const float *x;           // (frames, J*F) ray-encoded keypoints
    long long enc_ws;         // window stride in elements (frames * J*F)
    int enc_rows;             // GEMM rows per window (RF/3 for a temporal branch, 1 for GlobalInfo)
    int enc_jf;               // J*F
    int enc_cur;              // element offset of the "current" frame inside a window (tcur * J*F)
    int enc_step;             // frames between consecutive operand rows of a window: 3 (stride-3 expand_conv), 1 (dense ablation)
    unsigned enc_bytes;       // size of the raw input in bytes (buffer-descriptor bound)
    int res_tap;              // fused first level: which frame of a triple is the residual (1 centre, 2 causal)
    int pad2_;
    // --- UV input mode (cam != nullptr): x holds pixel keypoints (frames, J, 2), `lut` is the UV variant of the
    // tables (element byte offsets into that layout, the ray component 0/1/2 in the two low bits) and every gathered
    // value is encoded on its way into LDS with the camera row of the window its operand row belongs to:
    // ray = ((u-cx)/fx, c*y+s, -s*y+c), y = (v-cy)/fy, in float64 then cast (lib/camera/camera.py:423-471).
    const double *cam;        // rows {fx, fy, cx, cy, cos(pitch), sin(pitch), 0, 0}
    long long cam_stride;     // doubles between consecutive windows' rows (0: one camera for all)
    // --- the fused first level again as a register-chained tile (wchain != nullptr; r3d_chain.hpp): the three layers' weights as
    // ONE stream of 16 KiB slabs in the order the tile multiplies them (r3d_model.cpp, pack_chain) ---
    const float *wchain;
};

// Kernel argument of one persistent GEMM launch.  `tiles`/`wg_off` live in HBM (built once per
// (plan, batch size) by r3d_schedule.cpp); the problem table travels in the kernarg segment.
struct LaunchArgs {
    const int4 *tiles;    // {problem | MI << 8, first row, first column, 0}
    const int *wg_off;    // [grid + 1]: chunk c executes tiles [wg_off[c], wg_off[c+1])
    int nprob;
    int ks;               // (unused: the split-K factor travels with each tile)
    long long *dbg;       // optional phase timestamps (R3D_TIMING builds only)
    GemmProb p[MAX_PROB];
};

// ---- the whole forward as ONE persistent launch (r3d_forward_f32): tiles of every DAG level in one list per workgroup,
// ordered by tile-level dependencies instead of kernel boundaries.  The network is row-local - a tile of rows R of layer L
// reads rows of layer L-1 that belong to the same windows - so a tile waits for exactly the producer tiles of its
// windows: one ready counter per (problem, 32-row unit) in the caller's workspace counts finished 32-column granules.
constexpr int FWD_TILE_INT4 = 6;        // a tile descriptor: 24 ints (below)
constexpr int FWD_MAX_DEP = 8;
// ints of a descriptor: [0] problem | units << 8   [1] first row   [2] first column   [3] split-K factor
//                       [4] number of dependency ranges   [5] index of the first unit's ready counter
//                       [6] granules (32 columns) this tile adds to each of its units' counters   [7] unused
//                       [8 + 2d] first counter of range d   [9 + 2d] counters in the range | granules required << 16
struct FwdArgs {
    const int4 *tiles;        // FWD_TILE_INT4 int4 per tile
    const int *wg_off;        // [grid + 1]: workgroup b executes tiles [wg_off[b], wg_off[b+1])
    const GemmProb *probs;    // the call's problem table (absolute pointers; written by r3d_bind_f32 ahead of the launch)
    unsigned *cnt;            // ready counters, zeroed by r3d_bind_f32; cnt[ncnt] is the abort flag (a spin gave up)
    unsigned *cnt_next;       // the other bank of counters (library-owned control region), zeroed by THIS launch for the
                              // next call, which then needs no r3d_bind_f32 of its own; nullptr: one bank, bound per call
    int ncnt;
    int poll;                 // GEMV / latency tiles read their operands until no ACT_SENTINEL is left instead of waiting for counters
    void *arm;                // the other bank of activations: filled with sentinels by this launch, for the next call (or nullptr)
    long long arm_vec4;       // ... its size in 16-byte units
    long long spin_ticks;     // bound of a dependency wait in ticks of the 100 MHz wall clock (r3d_set_option, R3D_OPT_SPIN_TIMEOUT_MS)
    int fault_tile1;          // test hook (R3D_FAULT_TILE=<n>): workgroup 0's n-th tile behind the first level never raises its counters, and
                              // its n-th GEMV tile neither stores nor reports (0: none; n + 1 stored)
    long long *dbg;
};
enum { FWD_KERNEL_F32 = 0, FWD_KERNEL_B3 = 1, FWD_KERNEL_LAT = 2, FWD_KERNEL_CLIP = 3, FWD_KERNEL_CHAIN = 4, FWD_KERNEL_COUNT = 5 };   // specialisations of the single-launch forward
constexpr int BIND_NPTR = 19;  // pointer fields of a GemmProb, in the order r3d_bind_f32 walks them
enum { BIND_NULL = 0, BIND_WS, BIND_ARENA0, BIND_ARENA1, BIND_IARENA0, BIND_IARENA1, BIND_X, BIND_PARAM, BIND_CAM, BIND_NBASE };
struct BindArgs {
    const GemmProb *rel;           // problems with byte OFFSETS in their pointer fields
    const unsigned char *tags;     // [nprob][BIND_NPTR]: which base each pointer field is relative to (BIND_*)
    GemmProb *out;
    unsigned *cnt;
    int nprob, ncnt;
    const void *base[BIND_NBASE];
    long long enc_ws, cam_stride;  // per call: window stride in elements, doubles between camera rows
    unsigned enc_bytes;
    int param_stride;
    void *arm;                     // activation bank(s) to fill with sentinels (poll mode; nullptr otherwise)
    long long arm_vec4;
};

constexpr int MAX_DEC = 6;     // 5 body-part decoders + the trajectory decoder
// Fused decoder tail: the last Linear (1024 -> 3*n_g) of every Integration block, the joint
// reassembly (rie.py:415-432) and the trajectory add (trainer.py:353) in one pass.
struct DecodeArgs {
    const float *h[MAX_DEC];   // (B, 1024) hidden activations
    const float *w[MAX_DEC];   // packed rows [n_out][1024]
    const float *bias[MAX_DEC];
    int n_out[MAX_DEC];
    int first[MAX_DEC];        // index of the source's first output in the flat output list
    int nsrc;                  // sources; the trajectory source (if any) is the last one
    int has_pos, has_trj;
    int J;
    long long B;
    float *out;                // pos: (B, J, 3);  trj-only: (B, 3)
    float *out_trj;            // optional (B, 3)
    const unsigned *abort_flag;   // single-launch forward: nonzero when a dependency spin gave up - the outputs become NaN
    unsigned *status;             // ... and this word (pinned host memory of the handle: Model::status_host) becomes 1 - r3d_status
    int slot[5 * 16];          // flat pos output index -> element of (J,3) it lands in
};

// ------------------------------------------------------------------ host side

struct TensorSpec {
    std::string key;
    int rank;
    int64_t shape[4];
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < rank; ++i) n *= shape[i];
        return n;
    }
};

// One GEMM layer = Conv1d(k3,s3) / Conv1d(k1) / Linear, optionally followed by eval BatchNorm.
// bf16x3 mode below this many windows per call runs the fp32 tiles (both weight copies are resident): a bf16x3 tile's
// fixed cost is the larger one, and with a handful of tiles per launch nothing else counts - 0.325 against 0.288 ms at
// 64 windows, 0.370 against 0.386 at 128 (bench.py --batch).
inline int64_t b3_min_batch() {
    return 96;
}

inline bool env_on(const char *name) {      // set and not "0"
    const char *e = getenv(name);
    return e && atoi(e) != 0;
}
// Development and test switches (plan / tile-kind A/Bs, schedule dumps, fault injection) exist only in the hooks build of the
// library - libray3d_hip_hooks.so, the same sources with -DR3D_TEST_HOOKS, which tests/ and tools/ load when they need one
// (ray3d_amd/_capi.py, use_hooks).  In the product library they are compile-time constants: no getenv on any call path.
// What the product reads from the environment: R3D_BF16X3 (r3d_create) and R3D_STAGED (first use), both documented in
// include/ray3d_hip.h, and the CU-mask variables of the runtime (schedule_get).
#ifdef R3D_TEST_HOOKS
inline bool hook_on(const char *name) { return env_on(name); }
inline const char *hook_env(const char *name) { return getenv(name); }
#else
inline bool hook_on(const char *) { return false; }
inline const char *hook_env(const char *) { return nullptr; }
#endif

struct Layer {
    std::string weight_key;   // "<prefix>.weight"
    std::string bias_key;     // "" when the layer has no bias
    std::string bn_prefix;    // "" when no BatchNorm follows
    int taps;                 // 3 for the strided convs, 1 otherwise
    int cin;                  // input channels per tap
    int N, K, Npad, Kpad;     // K = taps*cin
    float slope;
    std::vector<int> colmap;  // optional: GEMM column of torch column (tap*cin + c); empty = identity
    std::vector<int> colmap_neg;   // first layers: column the same weight is SUBTRACTED from (-1: none)
    bool frag;                // packed in MFMA fragment order (GEMM layers) or row-major [N][Kpad] (decoder tail)
    size_t w_off, b_off;      // offsets (floats) into the packed arena
    bool bf3 = false;         // also packed in bf16-MFMA operand order for gemm_tile_b3 (the FCBlocks' 1024-wide Linears)
    bool bf3_conv = false;    // ... and the three layers of a fused first level (first_level_taps_b3)
    size_t wb3_off = 0;       // offset (floats) of that copy in the arena: Npad * Kpad floats
    // Linear layers that read a TemporalBlock's `shrink` output take the block's last activations instead: shrink is a
    // 1x1 convolution with bias and NO activation (rie.py:105), so W (S h + s) = (W S) h + W s is folded into the
    // consumer's weights at r3d_finalize (in double) and the shrink launch disappears.  One entry per folded input range:
    // reference columns [ref_col0, ref_col0 + ref_width) become new_width columns of layer `shrink_layer`'s input.
    struct Pre { int ref_col0, ref_width, new_width, shrink_layer; };
    std::vector<Pre> pre;
    int cin_ref = 0;          // input width of the reference layer (== cin unless `pre` is used)
    // Synthetic per-FRAME form of a first layer (shared_of >= 0; no state_dict tensor of its own): expand_conv is linear in
    // its operand, whose columns are either relative to the operand row's first frame or to the window's current frame
    // (tables above ENC_INVALID), so a row's pre-activation is E[first frame] + V[current frame] with
    //   E[p] = W[:, row-relative columns] . gather(p) + b      V[q] = W[:, current-frame columns] . x(q)
    // - two per-frame vectors that every window of a clip which contains the frame shares (trainer.py:47-58 makes N
    // windows of a clip).  The layer has N = 2 C rows: [E | V], both gathered at the SAME frame
    // (shared_split = first current-frame column of the source layer), linear (slope 1), bias in E.
    int shared_of = -1;
    int shared_split = 0;
    // expand_conv of a TemporalBlock whose fused first level can run as the register-chained tile (256 channels, K0 = 64):
    // offset (floats) of the three layers' slab stream in the arena, (K0 / 16 + 64) slabs of 4096 floats; 0: none
    size_t chain_off = 0;
    int chain_l1 = -1, chain_l2 = -1;   // the level's 3-tap and 1x1 layers
};

struct Model {
    r3d_config cfg;
    int RF;
    std::vector<TensorSpec> specs;                 // the state_dict grammar
    std::map<std::string, int> spec_index;
    std::vector<std::vector<float>> host_weights;  // by spec index
    std::vector<bool> have;
    std::vector<Layer> layers;
    std::map<std::string, int> layer_index;        // by weight prefix, e.g. "GlobalInfo.fc_1"
    // per branch: joints and first-layer K
    struct Branch {
        std::string prefix;           // "LocalLayer_Torso" ...
        std::vector<int> joints;
        int cin, k0, k0pad;
        size_t lut_off;               // offset (ints) into the int arena
        size_t lut_uv_off;            // the same tables for the UV input mode (in_features == 3 only)
        size_t lut_frame_off = 0, lut_frame_uv_off = 0;   // tables of the per-frame form (every column relative to the row's own frame)
        int frame_layer = -1;         // the synthetic [E | V] layer (Layer::shared_of), -1: none
    };
    std::vector<Branch> branches;
    size_t global_lut_off = 0;        // LUT of GlobalInfo.fc_1's input (the current frame)
    size_t global_lut_uv_off = 0;
    bool use_b3 = false;              // opt-in (R3D_BF16X3=1 at r3d_create): M = B layers on the bf16 matrix cores
    bool fold_shrink = false;         // the TemporalBlocks' shrink folded into its consumers (Layer::pre)
    std::vector<float> arena;         // packed floats (host mirror)
    std::vector<int> iarena;          // LUTs
    float *d_arena = nullptr;
    int *d_iarena = nullptr;
    int device = -1;
    bool finalized = false;
    bool dirty = true;
    // r3d_set_option
    bool opt_staged = false;          // this handle's forwards run one launch per level (no co-residency assumption)
    int spin_timeout_ms = 1000;       // bound of a dependency wait of the single-launch forward
    int cu_limit = 0;                 // R3D_OPT_CU_LIMIT: CUs of the (masked) stream this handle's forwards run on; 0 = the whole device
    void *last_fwd_stream = nullptr;  // ... the stream of its last masked forward (a second masked stream waits for it: one control region per handle)
    hipEvent_t order_ev = nullptr;
    // R3D_OPT_LANES = n (2 | 4): n CU-masked streams the LIBRARY owns - lane k: CUs c of every XCD with c % n == k - each with its own
    // schedules / control regions (Plan::schedules by (B, lane)); the packed weights are this handle's, shared by all lanes.  A forward
    // whose stream IS a lane's stream runs there; any other stream is served round-robin: the lane waits for the caller's stream,
    // and the caller's stream waits for the lanes at r3d_lanes_join (or when the same lane comes round again).
    int lanes = 0;
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr, in = nullptr;
        // streams that had a forward relayed to this lane and have not joined it since (r3d_lanes_join): `done` is re-recorded behind
        // every relayed forward and the lane is in order, so one wait on it covers all of them; a join by ONE stream must not
        // make the lane look joined to the others
        std::vector<void *> waiters;
    };
    Lane lane[4];
    int next_lane = 0;
    const unsigned *last_clk_dev = nullptr;   // the clock stamp of the last single-launch forward (two words of its counter bank: r3d_last_clock)
    unsigned *status_host = nullptr;  // pinned host word the decoder kernel raises when a wait gave up (r3d_status reads and clears it)
    // profiling
    bool profiling = false;
    struct Rec {
        hipEvent_t e0, e1;
        r3d_launch_record r;
    };
    std::vector<Rec> recs;
    int nrec = 0;
    // Launch plans live in a registry keyed by the (never reused) ids of the models they join - r3d_plan.cpp;
    // a model's destructor drops every plan that names it, so no plan outlives a partner.
    uint64_t id = 0;
    ~Model();
};

struct BufferSpec {
    std::string name;
    int64_t floats_per_window;   // rows_per_window * ld
    int external;                // 0 workspace, 3 the caller's camera-parameter rows (input)
    int64_t offset_per_window;   // workspace offset / B (floats)
};

struct ProbSpec {
    int model;                   // 0 or 1: which model's arena the layer lives in
    int layer;
    int rows_per_window;
    int nseg;
    struct Seg { int buf, col, ld, width; } seg[MAX_SEG];
    int res_buf, res_col, res_ld;
    int c_buf, c_col, c_ld;
    int layer2;                  // >= 0: second layer of a fused pair (applied to the first one's output tile)
    int layer3;                  // >= 0 (with enc_lut >= 0): third layer of the fused first level
    int enc_lut;                 // >= 0: fused-encode problem, offset of its LUT in the model's int arena
    int enc_lut_uv;              // the UV-mode tables of the same problem (-1: in_features != 3)
    bool enc_kernel;             // runs in r3d_gemm_enc_f32 (the model's first level is not fused), not in r3d_gemm_f32
    int enc_rows;
    int enc_step;                // frames between the operand rows of a window (3; 1 for the dense ablation's stride-1 expand_conv)
    int frame_col = -1;          // fused first level: first column of this branch's [E | V] block in the plan's per-frame buffer (Plan::frame_buf), -1: none
    std::vector<int> deps;
    int depth;
    double flops_per_window;     // 2 * rows * K_true * N_true
};

// Per-(plan, batch) work distribution of the persistent GEMM launches.
struct StageSchedule {
    int nwg;               // grid size
    int ks;                // largest split-K factor among this launch's tiles (1, 2 or 4)
    int kind;              // STAGE_BIG / STAGE_ENC: which kernel runs the launch
    int ntiles;
    size_t tiles_off;      // offsets (in int4 / int) into Schedule::d_tiles / d_wgoff
    size_t wgoff_off;
    double flops, bytes;   // algorithmic, for the launch records
    double imbalance;      // max chunk cost / mean chunk cost
    double makespan;       // modelled cycles of the longest chunk
};

struct Schedule {
    int64_t B = 0;
    bool pinned = false;   // named in r3d_prepare: never evicted (a captured hipGraph holds its device pointers) until r3d_release
    int spill_row0 = -1;   // rows [spill_row0, M) of Plan::spill_prob run in the following launch; -1: the plain
                           // level assignment (Plan::stages / stages_alt) is in use, else Plan::stages_spill / _alt
    const std::vector<std::vector<int>> *levels = nullptr;   // the assignment this schedule was built for
    std::vector<StageSchedule> stages;
    int4 *d_tiles = nullptr;
    int *d_wgoff = nullptr;
    // the launch of per-frame first layers a clip call runs ahead of its forward (Plan::frame_probs; B + RF - 3 rows each)
    StageSchedule frame_stage{};
    int4 *d_frame_tiles = nullptr;
    int *d_frame_wgoff = nullptr;
    // single-launch form (empty / null when the plan has launches it cannot hold: r3d_gemm_enc_f32 stages)
    struct Fwd {
        int grid = 0, ntiles = 0, ncnt = 0, nprob = 0;
        std::vector<int> cnt_base;            // per table index: first ready counter
        int4 *d_tiles = nullptr;              // FWD_TILE_INT4 int4 per tile
        int *d_wgoff = nullptr;
        GemmProb *d_rel[4] = {nullptr, nullptr, nullptr, nullptr};   // relative problem tables: rays / UV input, + 2: first levels on the per-frame buffer (CallShape::shared)
        unsigned char *d_tags[4] = {nullptr, nullptr, nullptr, nullptr};
        std::vector<int> h_tiles, h_wgoff;    // host copies of the lists (diagnostics)
        double flops = 0, bytes = 0;
        bool uses_gather = false;             // some problem gathers from the input (UV mode selects the _uv kernel)
        int kernel = FWD_KERNEL_F32;          // which specialisation runs these lists: FWD_KERNEL_* (by the tile kinds they hold)
        // Library-owned control region of the calls that are not being captured into a graph: two banks of ready
        // counters (+ abort flag) and the bound problem table.  A call whose buffers are the ones the table was bound
        // to skips r3d_bind_f32: it runs on the bank the previous launch zeroed and zeroes the other one itself.
        // (Captured calls keep their control region in the caller's workspace and bind inside the graph: a replay
        // must not depend on, or disturb, what eager calls left here.)
        char *d_ctrl = nullptr;
        size_t bank_bytes = 0;
        // Calls of a few windows (tiles of the GEMV / latency kinds in the lists): two library-owned banks of ACTIVATIONS as
        // well, each filled with sentinels by the launch that runs on the other one - the tiles then take data as its
        // own ready flag (r3d_kernels.hip, ACT_SENTINEL).  Not while capturing: a captured call uses the caller's
        // workspace and the ready counters.
        char *d_act = nullptr;
        size_t act_bytes = 0;
        struct Bound {
            bool valid = false;
            int bank = 0;
            const void *base[BIND_NBASE] = {nullptr};
            long long enc_ws = 0, cam_stride = 0;
            unsigned enc_bytes = 0;
            int param_stride = 0, uv = 0;
        } bound;
    } fwd;
    ~Schedule();
};

struct Plan {
    const Model *m[2] = {nullptr, nullptr};   // m[0] may be pos or trj (single), m[1] partner
    std::vector<BufferSpec> buffers;
    std::vector<ProbSpec> probs;
    std::vector<std::vector<int>> stages;      // problem ids per launch
    // Row spill: a second level assignment (r3d_plan.cpp) in which the last rows of problem `spill_prob` may run one
    // launch later than the rest (entry | STAGE_SPILL_IN there), so that the launch it belongs to need not open a
    // nearly empty extra round of tiles.  Whether it is used, and how many rows move, is decided per batch size in
    // schedule_get (Schedule::spill_row0).  Empty when the plan has no such problem.
    std::vector<std::vector<int>> stages_spill;
    // the spill assignment with the problem that reads the spilled one kept right behind it (instead of as late as its
    // users allow); empty when that gives the same assignment.  schedule_get models all of them per batch size and
    // keeps the shortest.
    std::vector<std::vector<int>> stages_spill_alt;
    // every problem as early as its inputs allow (no problem held back to level the launches): a call of up to four windows
    // is a latency chain - 0.133 against 0.138 ms at one window; from 16 windows on the levelled assignment wins
    std::vector<std::vector<int>> stages_asap;
    int spill_prob = -1;
    int kind = 0;                // PLAN_FUSED, or one of the less fused plans of small calls (plan_kind)
    int64_t floats_per_window = 0;
    int64_t tail_floats = 0;     // slack behind the last buffer (the dense ablation's overlapping operand rows read past a window's end)
    // Clip calls (window stride one frame): the first layers evaluated once per input FRAME instead of once per window
    // row (Layer::shared_of) by a launch of gathered GEMMs ahead of the forward, into one buffer of frame_ld floats per
    // frame - the blocks [E | V] of every fused first level side by side - that the first-level tiles then read instead of
    // gathering and multiplying (r3d_kernels.hip, first_level_shared).  The buffer holds B + RF - 1 rows: B of them from
    // its per-window share, the rest from Plan::tail_floats.
    int frame_buf = -1, frame_ld = 0;
    struct FrameProb { int model, layer, col, lut, lut_uv; };
    std::vector<FrameProb> frame_probs;
    int emb_buf[2] = {-1, -1};
    int param_buf = -1;          // pseudo-buffer standing for r3d_input::param_dev
    // fused decoder tail: (model, layer, hidden buffer) per Integration block
    struct Dec { int model, layer, hbuf; };
    std::vector<Dec> decs;
    int pos_model = -1, trj_model = -1;
    std::map<int64_t, Schedule *> schedules;   // by batch size (small LRU, see schedule_get)
    std::vector<int64_t> schedule_lru;
    ~Plan();
};

// Where a problem's pointers point: real addresses (a launch's kernarg table) or all-null bases, which leaves byte
// OFFSETS in the pointer fields plus one BIND_* tag per field - the schedule's relative table that r3d_bind_f32 turns
// into a call's absolute one on the device (r3d_kernels.hip).
struct Bases {
    const char *ws = nullptr, *arena[2] = {nullptr, nullptr}, *iarena[2] = {nullptr, nullptr};
    const char *x = nullptr, *param = nullptr, *cam = nullptr;
};
struct CallShape {
    bool uv = false;
    bool shared = false;    // the fused first levels read the per-frame buffer (Plan::frame_buf) instead of gathering
    int64_t window_stride = 0, param_stride = 0, cam_stride = 0;
    long long frames = 0;
};

int fill_prob(const Plan *pl, const ProbSpec &q, int64_t B, const Model *a, const Bases &bs, const CallShape &cs, GemmProb &g,
              unsigned char *tags);   // r3d_api.cpp

void set_error(const char *fmt, ...);
const char *last_error();
int hip_fail(hipError_t e, const char *what);

Model *model_create(const r3d_config &cfg);
int model_set_weight(Model *m, const char *key, const float *host, const int64_t *shape, int rank);
int model_finalize(Model *m);
enum { PLAN_FUSED = 0, PLAN_SMALL = 1, PLAN_MEDIUM = 2, PLAN_LARGE = 3 };   // everything but the top level fused / nothing / first level only / the top level too (r3d_plan.cpp)
int plan_kind(int64_t B);                       // the plan a call of B windows runs
std::vector<int64_t> plan_kind_edges();         // the largest window count of every plan kind that has one (r3d_workspace_bytes)
Plan *plan_get(Model *a, Model *b, int kind);
void plans_drop(const Model *m);   // delete every cached plan (and its schedules) that names `m`
void lanes_destroy(Model *m);       // R3D_OPT_LANES: the lanes' streams and events (r3d_api.cpp)
bool plans_pinned(const Model *m); // some schedule of a plan that names `m` is pinned (r3d_prepare: a captured graph may point into it)
constexpr int STAGE_SPILL_IN = 1 << 30;   // flag on a Plan::stages entry: the spilled rows of that problem
constexpr int GEMM_SCHED_MAX_UNITS = 6;   // widest tile of r3d_gemm_f32: 6 x 32 rows
struct SchedProb {
    int M, N, nk;
    int max_ks;      // largest split-K factor the operand allows (1 = none: fused-prologue operands, or a
                     // concatenated operand with a boundary that is not a multiple of 32*KS)
    int max_units;   // per-problem cap on 32-row units per tile (0 = the launch default)
    int nk2 = 0;     // K-loop iterations of the fused further layers, in 32-row units (cost only)
    int row0 = 0;    // first row this launch computes (a multiple of 32): rows [row0, M)
    bool gemv = false;   // M <= GEMV_ROWS rows of a plain layer: 32-column GEMV tiles (tile code ks == 8), r3d_kernels.hip gemv_tile
    bool lat = false;    // ... of up to 32 rows: 32-column latency tiles on the matrix cores (tile code 16), lat_tile
    bool nb_ok = false;  // a plain fp32 layer of whole 32-column blocks whose single-unit tiles may be 4 - 7 blocks wide (gemm_tile_nb)
};
constexpr int GEMV_ROWS = 4;          // == GEMV_MAX_M of the kernels (at eight rows the MFMA split-K tiles are the faster ones: 0.207 against 0.222 ms)
constexpr int COL_GRANULE = 32;       // ready counters and cover checks count columns in granules of this many
constexpr int NB_CODE = 64;           // tile code NB_CODE + nb: one 32-row unit x nb column blocks of 32, nb = 4 .. 7 (r3d_tiles.hpp, gemm_tile_nb)
constexpr bool tile_is_nb(int ks) { return ks > NB_CODE; }
constexpr bool tile_is_narrow(int ks) { return ks >= 8 && ks < NB_CODE; }    // GEMV (8) / latency (16) tiles
constexpr int tile_width(int ks) { return ks > NB_CODE ? (ks - NB_CODE) * 32 : ks >= 8 ? 32 : 256 / ks; }   // columns of a tile by its code
void schedule_stage(const std::vector<SchedProb> &probs, int nwg, int max_units, std::vector<int4> &tiles,
                    std::vector<int> &wgoff, StageSchedule &out, bool enc = false);
// index of weight element (output channel o, GEMM column k) in the fragment-ordered packing
inline size_t frag_index(int o, int k, int nk) {
    const int nb = o >> 5, li = o & 31, kt = k >> 5, kin = k & 31, lh = kin >> 4, q = (kin & 15) >> 2, e = kin & 3;
    return ((((size_t)nb * nk + kt) * 4 + q) * 64 + (lh * 32 + li)) * 4 + e;
}
const std::vector<std::vector<int>> *schedule_build_host(const Plan *pl, int64_t B, int nwg, int &spill_row0, std::vector<int4> &tiles,
                                                        std::vector<int> &wgoff, std::vector<StageSchedule> &stages);
bool schedule_build_fwd(const Plan *pl, int64_t B, int nwg, const std::vector<std::vector<int>> &levels, const std::vector<StageSchedule> &stages,
                        const std::vector<int4> &tiles, const std::vector<int> &wgoff, Schedule::Fwd &fw, std::vector<int> &out_tiles,
                        std::vector<int> &out_wgoff);
Schedule *schedule_get(Plan *pl, int64_t B, int nwg, bool pin = false, int lane = 0);   // nullptr + set_error on failure
inline int64_t schedule_key(int64_t B, int lane) { return B | ((int64_t)lane << 48); }    // key of Plan::schedules (lane 0: handles without lanes)
int device_cu_count();

// kernel launchers (r3d_kernels.hip)
enum { STAGE_BIG = 0, STAGE_ENC = 1 };   // r3d_gemm_f32 / r3d_gemm_enc_f32
// the kernels live in one translation unit per family (r3d_k_*.hip, compiled side by side); each exports its entry points
typedef void (*GemmKernel)(const LaunchArgs);
typedef void (*FwdKernel)(const FwdArgs);
GemmKernel gemm_kernel_f32(bool uv);      // r3d_gemm_f32 / r3d_gemm_uv_f32             (r3d_k_gemm.hip)
GemmKernel gemm_kernel_enc(bool uv);      // r3d_gemm_enc_f32 / r3d_gemm_enc_uv_f32     (r3d_k_gemm_enc.hip)
GemmKernel gemm_kernel_b3(bool uv);       // r3d_gemm_b3 / r3d_gemm_uv_b3               (r3d_k_gemm_b3.hip)
FwdKernel fwd_kernel_f32(bool uv);        // r3d_forward_f32 / r3d_forward_uv_f32       (r3d_k_fwd_f32.hip)
FwdKernel fwd_kernel_b3(bool uv);         // r3d_forward_b3 / r3d_forward_uv_b3         (r3d_k_fwd_b3.hip)
FwdKernel fwd_kernel_lat(bool uv);        // r3d_forward_lat / r3d_forward_uv_lat       (r3d_k_fwd_lat.hip)
FwdKernel fwd_kernel_clip(bool uv);       // r3d_forward_clip_f32 / _clip_uv_f32        (r3d_k_fwd_clip.hip)
FwdKernel fwd_kernel_chain(bool uv);      // r3d_forward_chain_f32 (experiment, rays only) (r3d_k_fwd_chain.hip)
hipError_t launch_gemm_stage(const LaunchArgs &args, int nwg, int kind, bool uv, hipStream_t stream);   // uv: the launch gathers pixel keypoints
hipError_t launch_decode(const DecodeArgs &args, hipStream_t stream);
hipError_t launch_forward(const FwdArgs &args, int nwg, int kind, bool uv, hipStream_t stream);
const char *forward_kernel_name(int kind, bool uv);
int forward_resident_capacity(int kind, bool uv);      // workgroups of that kernel the current device holds at once (0: unknown)
hipError_t launch_bind(const BindArgs &args, hipStream_t stream);
bool forward_single_launch();   // the single-launch form is in use (R3D_STAGED=1 turns it off)
size_t fwd_ctrl_bytes(const Plan *pl, int64_t B);   // workspace bytes behind the activations: counters + problem table

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Fused feature-encoding prologue tables (built in r3d_model.cpp, consumed by r3d_gemm_enc_f32).
// The reference feeds a branch cat(x, x - root, x - x_current) per tap (rie.py:301-315); the layer being linear, the
// kernel multiplies the raw values instead - (W1+W2+W3) x - W2 root - W3 x_current, folded into the packed
// weights in float64 - so an operand column is ONE gathered input element.  Columns are grouped: the x values of
// the three taps, the root joint's values (when the group does not contain joint 0 itself), the window's
// current frame; each group padded to a multiple of four columns so that the four columns one staging thread
// handles share their base.  Per column k:
//   lut1[k]   = byte offset of the element from the row's first frame - or from the window's current frame
//               (quirk Q1) when lutk[k/4] is set; ENC_INVALID for padding columns;
//   lutk[k/4] = 1 when the chunk is current-frame relative (GlobalInfo's input: every chunk).
// ENC_INVALID pushes the address past the buffer descriptor's bound: the load returns 0.
// UV variant (Model::Branch::lut_uv_off): the element (frame, joint, f) of the ray layout is computed from the pixel
// coordinate at ((frame * J + joint) * 2 + (f > 0)) * 4 bytes of the (frames, J, 2) input; lut1[k] holds that offset
// with f in its two low bits (offsets are multiples of four).
constexpr int ENC_INVALID = (int)0x80000000u;

int launch_clip_metrics(const float *pred, const float *gt, long long n, int J, const double *Rn2w, const double *Tn2w,
                        double *out, hipStream_t stream);

}  // namespace r3d
