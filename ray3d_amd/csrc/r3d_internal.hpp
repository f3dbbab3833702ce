// Internal declarations of libray3d_hip.so (not part of the ABI; see include/ray3d_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
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
constexpr int DEC_SLOT = 16;  // column slot per body part in the decoder output matrix
constexpr int MAX_BRANCH = 6; // 5 body parts (pos) + 1 (trj) encoded by one prologue launch

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
};

// Kernel argument of one persistent GEMM launch.  `tiles`/`wg_off` live in HBM (built once per
// (plan, batch size) by r3d_schedule.cpp); the problem table travels in the kernarg segment.
struct LaunchArgs {
    const int4 *tiles;    // {problem | MI << 8, first row, first column, 0}
    const int *wg_off;    // [grid + 1]: chunk c executes tiles [wg_off[c], wg_off[c+1])
    int nprob;
    int pad_;
    long long *dbg;       // optional phase timestamps (R3D_TIMING builds only)
    GemmProb p[MAX_PROB];
};

struct EncodeBranch {
    float *a0;          // (B * RF/3, k0pad)
    const int *lut;     // k0pad entries, see encode_lut_entry()
    int k0pad;
    int pad_;
};

struct EncodeArgs {
    const float *x;            // rays or uv
    const double *cam;
    const float *param;
    long long window_stride;   // in frames
    long long param_stride, cam_stride;
    int mode, J, F, RF, tcur, nbranch;
    long long B;
    EncodeBranch br[MAX_BRANCH];
    float *cur;                // (B, CUR_LD) current-frame matrix, zero padded
    // camera embedding MLPs (BN folded) of up to two models
    int nembed, E;
    const float *emb_w[2];     // packed [w1 (32,E) | b1 (32) | w2 (D,32) | b2 (D)]
    float *emb_out[2];         // (B, D)
    int emb_dim[2];
};

struct AssembleArgs {
    const float *dec;   // (B, 5*DEC_SLOT) decoder outputs, part g at column g*DEC_SLOT
    const float *trj;   // optional (B, ldt) root trajectory, broadcast over joints
    float *out;         // (B, J, 3)
    long long B;
    int J, ldt;
    int src[17 * 3];    // out[b, e] = dec[b, src[e]] (+ trj[b, e % 3])
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
struct Layer {
    std::string weight_key;   // "<prefix>.weight"
    std::string bias_key;     // "" when the layer has no bias
    std::string bn_prefix;    // "" when no BatchNorm follows
    int taps;                 // 3 for the strided convs, 1 otherwise
    int cin;                  // input channels per tap
    int N, K, Npad, Kpad;     // K = taps*cin
    float slope;
    size_t w_off, b_off;      // offsets (floats) into the packed arena
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
    };
    std::vector<Branch> branches;
    size_t embed_off = 0;             // packed embedding MLP in the float arena
    std::vector<float> arena;         // packed floats (host mirror)
    std::vector<int> iarena;          // LUTs
    float *d_arena = nullptr;
    int *d_iarena = nullptr;
    int device = -1;
    bool finalized = false;
    bool dirty = true;
    // profiling
    bool profiling = false;
    struct Rec {
        hipEvent_t e0, e1;
        r3d_launch_record r;
    };
    std::vector<Rec> recs;
    int nrec = 0;
    // cached plans keyed by the partner model (nullptr for single)
    std::map<const Model *, struct Plan *> plans;
    ~Model();
};

struct BufferSpec {
    std::string name;
    int64_t floats_per_window;   // rows_per_window * ld
    int external;                // 0 workspace, 1 out_dev, 2 out_trj_dev
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
    std::vector<int> deps;
    int depth;
    double flops_per_window;     // 2 * rows * K_true * N_true
};

// Per-(plan, batch) work distribution of the persistent GEMM launches.
struct StageSchedule {
    int nwg;               // grid size
    int ntiles;
    size_t tiles_off;      // offsets (in int4 / int) into Schedule::d_tiles / d_wgoff
    size_t wgoff_off;
    double flops, bytes;   // algorithmic, for the launch records
    double imbalance;      // max chunk cost / mean chunk cost
};

struct Schedule {
    int64_t B = 0;
    std::vector<StageSchedule> stages;
    int4 *d_tiles = nullptr;
    int *d_wgoff = nullptr;
    ~Schedule();
};

struct Plan {
    const Model *m[2] = {nullptr, nullptr};   // m[0] may be pos or trj (single), m[1] partner
    std::vector<BufferSpec> buffers;
    std::vector<ProbSpec> probs;
    std::vector<std::vector<int>> stages;      // problem ids per launch
    int64_t floats_per_window = 0;
    // prologue
    struct Enc { int model, branch, buf; };
    std::vector<Enc> enc;
    int cur_buf = -1;
    int emb_buf[2] = {-1, -1};
    // epilogue
    int dec_buf = -1;            // pos decoder matrix (assemble input), -1 when no pos model
    int trj_buf = -1;            // trj output matrix
    int pos_model = -1, trj_model = -1;
    std::map<int64_t, Schedule *> schedules;   // by batch size (small LRU, see schedule_get)
    std::vector<int64_t> schedule_lru;
    ~Plan();
};

void set_error(const char *fmt, ...);
const char *last_error();
int hip_fail(hipError_t e, const char *what);

Model *model_create(const r3d_config &cfg);
int model_set_weight(Model *m, const char *key, const float *host, const int64_t *shape, int rank);
int model_finalize(Model *m);
Plan *plan_get(Model *a, Model *b);
struct SchedProb { int M, N, nk; };
void schedule_stage(const std::vector<SchedProb> &probs, int nwg, std::vector<int4> &tiles, std::vector<int> &wgoff,
                    StageSchedule &out);
Schedule *schedule_get(Plan *pl, int64_t B, int nwg);   // nullptr + set_error on failure
int device_cu_count();

// kernel launchers (r3d_kernels.hip)
hipError_t launch_encode(const EncodeArgs &args, hipStream_t stream, int *blocks);
hipError_t launch_gemm_stage(const LaunchArgs &args, int nwg, hipStream_t stream);
hipError_t launch_assemble(const AssembleArgs &args, hipStream_t stream, int *blocks);

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// LUT entry for the feature-encoding prologue: bits [1:0] tap, [3:2] kind (0 x, 1 x - root,
// 2 x - current frame, 3 zero padding), [11:4] source element joint*F+f, [13:12] f.
inline int encode_lut_entry(int tap, int kind, int src, int f) {
    return (tap & 3) | ((kind & 3) << 2) | ((src & 255) << 4) | ((f & 3) << 12);
}

}  // namespace r3d
