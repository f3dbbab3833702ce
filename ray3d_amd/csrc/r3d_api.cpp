// extern "C" entry points of libray3d_hip.so (contract: include/ray3d_hip.h).
#include <algorithm>
#include <cstring>
#include <mutex>

#include "r3d_internal.hpp"

namespace r3d {
const char *last_error();

// output joint slot -> (body part, index inside the part), lib/model/rie.py:426-431 (quirk Q2:
// for J = 14 / 15 this is not the inverse of the input grouping).  Fills slot[] such that the
// o-th output of part g (flat index first[g] + o) lands in element slot[first[g] + o] of (J,3).
static void output_slots(int J, const int *first, int *slot) {
    int s = 0;
    auto put = [&](int part, int idx) {
        for (int f = 0; f < 3; ++f) slot[first[part] + idx * 3 + f] = s * 3 + f;
        ++s;
    };
    enum { T = 0, LA = 1, RA = 2, LL = 3, RL = 4 };
    if (J == 17) {
        put(T, 0);
        for (int i = 0; i < 3; ++i) put(LL, i);
        for (int i = 0; i < 3; ++i) put(RL, i);
        for (int i = 1; i < 5; ++i) put(T, i);
        for (int i = 0; i < 3; ++i) put(RA, i);
        for (int i = 0; i < 3; ++i) put(LA, i);
    } else if (J == 15) {
        put(T, 0); put(T, 1);
        for (int i = 0; i < 3; ++i) put(LL, i);
        for (int i = 0; i < 3; ++i) put(RL, i);
        for (int i = 0; i < 3; ++i) put(RA, i);
        for (int i = 0; i < 3; ++i) put(LA, i);
        put(T, 2);
    } else {
        put(T, 0);
        for (int i = 0; i < 3; ++i) put(LL, i);
        for (int i = 0; i < 3; ++i) put(RL, i);
        for (int i = 0; i < 3; ++i) put(RA, i);
        for (int i = 0; i < 3; ++i) put(LA, i);
        put(T, 1);
    }
}

static bool same_input_shape(const Model *a, const Model *b) {
    return a->cfg.num_joints == b->cfg.num_joints && a->cfg.in_features == b->cfg.in_features &&
           a->cfg.num_levels == b->cfg.num_levels && a->cfg.extrinsic_dim == b->cfg.extrinsic_dim;
}

// activations (the input is read in place in both modes: UV mode encodes the rays inside the gather), then - 256-byte
// aligned - the single-launch forward's control region: ready counters + abort flag, the call's problem table
static size_t workspace_act_bytes(const Plan *pl, int64_t B) {
    const size_t act = ((size_t)pl->floats_per_window * (size_t)B + (size_t)pl->tail_floats + 64) * sizeof(float);
    return (act + 255) / 256 * 256;
}
static size_t workspace_need(const Plan *pl, int64_t B) { return workspace_act_bytes(pl, B) + fwd_ctrl_bytes(pl, B); }

struct Recorder {
    Model *m;
    hipStream_t stream;
    size_t n = 0;
    bool on() const { return m->profiling; }
    hipError_t begin(const char *kernel, int stage, int blocks, double flops, double bytes) {
        if (!on()) return hipSuccess;
        if (n == m->recs.size()) {
            Model::Rec r;
            hipError_t e = hipEventCreate(&r.e0);
            if (e != hipSuccess) return e;
            if ((e = hipEventCreate(&r.e1)) != hipSuccess) return e;
            m->recs.push_back(r);
        }
        Model::Rec &r = m->recs[n];
        memset(&r.r, 0, sizeof r.r);
        strncpy(r.r.kernel, kernel, sizeof r.r.kernel - 1);
        r.r.stage = stage;
        r.r.blocks = blocks;
        r.r.flops = flops;
        r.r.bytes = bytes;
        return hipEventRecord(r.e0, stream);
    }
    hipError_t end() {
        if (!on()) return hipSuccess;
        return hipEventRecord(m->recs[n++].e1, stream);
    }
};

// One GEMM problem of the plan for a call of B windows.  `tags` (optional, BIND_NPTR entries): the base of every pointer field.
int fill_prob(const Plan *pl, const ProbSpec &q, int64_t B, const Model *a, const Bases &bs, const CallShape &cs, GemmProb &g,
              unsigned char *tags) {
    memset(&g, 0, sizeof g);
    unsigned char tg[BIND_NPTR] = {0};
    const Model *m = pl->m[q.model];
    const Layer &L = m->layers[q.layer];
    const int JF = a->cfg.num_joints * (cs.uv ? 2 : a->cfg.in_features);   // floats per input frame
    auto ws_ptr = [&](int buf, int col) {
        return reinterpret_cast<float *>(const_cast<char *>(bs.ws) + ((size_t)pl->buffers[buf].offset_per_window * (size_t)B + (size_t)col) * sizeof(float));
    };
    auto arena_ptr = [&](size_t off) { return reinterpret_cast<const float *>(bs.arena[q.model] + off * sizeof(float)); };
    const unsigned char TA = (unsigned char)(BIND_ARENA0 + q.model), TI = (unsigned char)(BIND_IARENA0 + q.model);
    int kend = 0;
    for (int s = 0; s < MAX_SEG; ++s) {
        if (s < q.nseg) {
            if (pl->buffers[q.seg[s].buf].external == 3) {      // the caller's camera-parameter rows
                g.a[s] = reinterpret_cast<const float *>(bs.param);
                g.lda[s] = (int)cs.param_stride;
                tg[s] = BIND_PARAM;
            } else {
                g.a[s] = ws_ptr(q.seg[s].buf, q.seg[s].col);
                g.lda[s] = q.seg[s].ld;
                tg[s] = BIND_WS;
            }
            kend += q.seg[s].width;
        } else {
            g.a[s] = g.a[0];
            g.lda[s] = g.lda[0];
            tg[s] = tg[0];
        }
        g.kend[s] = s < q.nseg ? kend : 0x7fffffff;
    }
    // the last real segment absorbs the rest - unless it is narrower than the padded K (embedder.w1 on
    // the 2-wide parameter rows): its true width bounds the buffer descriptor, the rest reads as zeros
    if (q.nseg > 0 && !(q.nseg == 1 && kend < L.Kpad)) g.kend[q.nseg - 1] = 0x7fffffff;
    if (q.enc_lut >= 0) {
        if (cs.uv && q.enc_lut_uv < 0) { set_error("internal: no UV tables for an encoded operand"); return R3D_ERR_STATE; }
        g.lut = reinterpret_cast<const int *>(bs.iarena[q.model] + (size_t)(cs.uv ? q.enc_lut_uv : q.enc_lut) * sizeof(int));
        tg[15] = TI;
        g.x = reinterpret_cast<const float *>(bs.x);
        tg[16] = BIND_X;
        g.cam = cs.uv ? reinterpret_cast<const double *>(bs.cam) : nullptr;
        tg[17] = cs.uv ? BIND_CAM : BIND_NULL;
        g.cam_stride = cs.cam_stride;
        g.enc_ws = cs.window_stride * JF;
        g.enc_rows = q.enc_rows;
        g.enc_step = q.enc_step;
        g.enc_jf = JF;
        g.enc_cur = (a->RF / a->cfg.in_features) * JF;   // quirk Q1: "current" frame is RF // in_features
        g.enc_bytes = (unsigned)((size_t)cs.frames * JF * sizeof(float));
        g.res_tap = 1 + m->cfg.causal;
        if (cs.shared && q.layer3 >= 0 && q.frame_col >= 0) {
            // clip call: the tile reads its expand_conv pre-activations from the per-frame buffer (Plan::frame_buf, written by
            // the launch ahead of the forward) - `x` is this branch's [E | V] block, a row per input frame of enc_jf floats,
            // enc_ws / enc_cur the window stride and the current frame's offset in those rows; no tables, no camera
            g.lut = nullptr;
            tg[15] = BIND_NULL;
            g.x = ws_ptr(pl->frame_buf, q.frame_col);
            tg[16] = BIND_WS;
            g.cam = nullptr;
            tg[17] = BIND_NULL;
            g.cam_stride = 0;
            g.enc_jf = pl->frame_ld;
            g.enc_ws = cs.window_stride * pl->frame_ld;
            g.enc_cur = (a->RF / a->cfg.in_features) * pl->frame_ld;
            g.enc_bytes = (unsigned)(((size_t)cs.frames * pl->frame_ld - (size_t)q.frame_col) * sizeof(float));
        }
    }
    g.w = arena_ptr(L.w_off);
    tg[4] = TA;
    const bool b3 = B >= b3_min_batch();
    if (b3 && L.bf3 && q.layer2 < 0 && q.enc_lut < 0) { g.wb3 = arena_ptr(L.wb3_off); tg[10] = TA; }
    g.bias = arena_ptr(L.b_off);
    tg[5] = TA;
    if (q.res_buf >= 0) { g.res = ws_ptr(q.res_buf, q.res_col); tg[6] = BIND_WS; }
    g.ldr = q.res_ld;
    g.c = ws_ptr(q.c_buf, q.c_col);
    tg[7] = BIND_WS;
    g.ldc = q.c_ld;
    g.M = (int)(B * q.rows_per_window);
    g.N = L.N;
    g.K = L.Kpad;
    g.slope = L.slope;
    if (q.layer2 >= 0) {
        const Layer &L2 = m->layers[q.layer2];
        if (b3 && q.layer3 < 0 && L.bf3_conv && L2.bf3_conv && q.nseg == 1) {   // gemm_tile_b3t
            g.wb3 = arena_ptr(L.wb3_off);
            g.w2b3 = arena_ptr(L2.wb3_off);
            tg[10] = tg[11] = TA;
        }
        g.w2 = arena_ptr(L2.w_off);
        g.bias2 = arena_ptr(L2.b_off);
        tg[8] = tg[9] = TA;
        g.K2 = L2.Kpad;
        g.slope2 = L2.slope;
    }
    if (q.layer3 >= 0) {
        const Layer &L3 = m->layers[q.layer3];
        const Layer &L2b = m->layers[q.layer2];
        if (b3 && L.bf3_conv && L2b.bf3_conv && L3.bf3_conv) {   // first_level_taps_b3
            g.wb3 = arena_ptr(L.wb3_off);
            g.w2b3 = arena_ptr(L2b.wb3_off);
            g.w3b3 = arena_ptr(L3.wb3_off);
            tg[10] = tg[11] = tg[12] = TA;
        }
        g.w3 = arena_ptr(L3.w_off);
        g.bias3 = arena_ptr(L3.b_off);
        tg[13] = tg[14] = TA;
        g.K3 = L3.Kpad;
        g.slope3 = L3.slope;
        // the register-chained form of the same three layers (r3d_chain.hpp): gathered rays, fp32 tiles - an experiment that only the
        // hooks build switches on (R3D_CHAIN=1, at r3d_finalize AND here): faster stand-alone, slower inside the forward
        if (L.chain_off != 0 && L.chain_l1 == q.layer2 && L.chain_l2 == q.layer3 && !cs.uv && !cs.shared && g.lut != nullptr &&
            hook_on("R3D_CHAIN")) {
            g.wchain = arena_ptr(L.chain_off);
            tg[18] = TA;
        }
    }
    if (tags) memcpy(tags, tg, sizeof tg);
    return R3D_OK;
}

// Two single-launch forwards must never be on the GPU at the same time: each needs ALL its workgroups resident (a waiting
// workgroup spins for tiles of workgroups that may not have been dispatched yet), and two such kernels from two streams
// could each hold part of the chip and wait for the rest forever (until the bounded spins give up).  Within a process the
// library therefore orders them: per device it remembers the stream and an event of the last single-launch forward, and a
// forward on ANOTHER stream first records an event behind the work of the previous forward's stream and waits for it (a
// device-side dependency, no host synchronisation).  The common case - one stream - costs a mutex and a compare.  Streams being captured are left alone (a capture
// must not wait on events from outside it): capture one forward stream per graph, replay graphs one at a time.
struct FwdOrder {
    std::mutex mu;
    hipEvent_t ev[64] = {nullptr};
    hipStream_t last[64] = {nullptr};
    bool have[64] = {false};
    // R3D_OPT_CU_LIMIT: forwards on CU-masked streams are not ordered against EACH OTHER (disjoint masks: that is their point), but
    // a whole-device forward and a masked one must never share the chip either: the streams that have run a masked forward since
    // the last whole-device forward waited for them
    std::vector<hipStream_t> masked[64];
    // ... and a masked stream waits behind the last whole-device forward ONCE, not with every call: `gen` counts the device's
    // whole-device forwards, `seen` which one each masked stream has waited for.  (An event per call would be recorded on the
    // whole-device forward's stream - usually the legacy default stream, where an event is behind the work of EVERY blocking
    // stream, the other lanes' forwards in flight included: the masked streams would run one after the other.)
    struct Seen { hipStream_t s; unsigned long long gen; };
    std::vector<Seen> seen[64];
    unsigned long long gen[64] = {0};
};
static FwdOrder g_fwd_order;
static std::mutex g_fwd_launch_mu;      // order_single_launch(before) .. launch .. order_single_launch(after) of one forward

// A stream of the library's own is about to be destroyed: nothing may record events on it any more.
static void order_forget(hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_fwd_order.mu);
    FwdOrder &o = g_fwd_order;
    for (int d = 0; d < 64; ++d) {
        o.masked[d].erase(std::remove(o.masked[d].begin(), o.masked[d].end(), stream), o.masked[d].end());
        o.seen[d].erase(std::remove_if(o.seen[d].begin(), o.seen[d].end(), [&](const FwdOrder::Seen &x) { return x.s == stream; }), o.seen[d].end());
        if (o.have[d] && o.last[d] == stream) o.have[d] = false;
    }
}

// `behind`: make `stream` wait (device-side) for everything `other` has been given so far.  A stream that is gone or capturing is skipped.
static hipError_t wait_behind(hipStream_t stream, hipStream_t other, hipEvent_t &ev) {
    if (other == stream) return hipSuccess;
    if (other == nullptr) {          // the legacy default stream: a blocking stream is behind its work already (and an event on it would be
        unsigned flags = 0;          // behind every other blocking stream's work too - see FwdOrder::seen)
        if (hipStreamGetFlags(stream, &flags) == hipSuccess && !(flags & hipStreamNonBlocking)) return hipSuccess;
        (void)hipGetLastError();
    }
    hipStreamCaptureStatus ocs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(other, &ocs) != hipSuccess || ocs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return hipSuccess; }
    if (!ev) {
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipEventRecord(ev, other);
    if (e != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }      // (the other stream is gone: nothing of it can still run)
    return hipStreamWaitEvent(stream, ev, 0);
}

// masked: the forward runs on a CU-masked stream with R3D_OPT_CU_LIMIT workgroups.  Rules: whole-device forwards are ordered among
// themselves and behind every masked forward issued before them; a masked forward is ordered behind the last whole-device forward;
// masked forwards of different streams are not ordered against each other.
static hipError_t order_single_launch(hipStream_t stream, bool before, bool masked = false) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipSuccess;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }
    if (cs != hipStreamCaptureStatusNone) return hipSuccess;
    std::lock_guard<std::mutex> lock(g_fwd_order.mu);
    FwdOrder &o = g_fwd_order;
    if (!before) {                                   // (after the launch: just remember whose it was - no event on the one-stream path)
        if (masked) {
            if (std::find(o.masked[dev].begin(), o.masked[dev].end(), stream) == o.masked[dev].end()) o.masked[dev].push_back(stream);
        } else {
            o.last[dev] = stream;
            o.have[dev] = true;
            ++o.gen[dev];
        }
        return hipSuccess;
    }
    // another stream ran the previous whole-device forward: an event behind everything that stream has been given so far, and wait for it
    if (o.have[dev] && o.last[dev] != stream) {
        bool wait = true;
        if (masked) {                                // (once per whole-device forward and masked stream)
            auto it = std::find_if(o.seen[dev].begin(), o.seen[dev].end(), [&](const FwdOrder::Seen &x) { return x.s == stream; });
            if (it == o.seen[dev].end()) o.seen[dev].push_back({stream, o.gen[dev]});
            else if (it->gen == o.gen[dev]) wait = false;
            else it->gen = o.gen[dev];
        }
        if (wait)
            if (hipError_t e = wait_behind(stream, o.last[dev], o.ev[dev]); e != hipSuccess) return e;
    }
    if (!masked) {
        for (hipStream_t ms : o.masked[dev])
            if (hipError_t e = wait_behind(stream, ms, o.ev[dev]); e != hipSuccess) return e;
        o.masked[dev].clear();                       // (they re-enter the list with their next masked forward)
    }
    return hipSuccess;
}

static int run(Model *pos, Model *trj, const r3d_input *in, int64_t B, float *out, float *out_trj, void *ws,
               size_t ws_bytes, void *stream_v) {
    Model *a = pos ? pos : trj, *b = pos ? trj : nullptr;
    if (!a) { set_error("forward: no model given"); return R3D_ERR_ARG; }
    if (!in || !in->x_dev || !out || B <= 0) { set_error("forward: null input/output or B <= 0"); return R3D_ERR_ARG; }
    for (Model *m : {a, b})
        if (m && (!m->finalized || m->dirty)) {
            set_error("forward called before r3d_finalize (or weights changed since)");
            return R3D_ERR_STATE;
        }
    if (b && !same_input_shape(a, b)) { set_error("pos and trj models disagree on J / F / levels / extrinsic_dim"); return R3D_ERR_ARG; }
    if (in->mode != R3D_INPUT_RAYS && in->mode != R3D_INPUT_UV) { set_error("bad input mode %d", in->mode); return R3D_ERR_ARG; }
    if (in->mode == R3D_INPUT_UV && (a->cfg.in_features != 3 || !in->cam_dev)) {
        set_error("R3D_INPUT_UV needs in_features == 3 and cam_dev");
        return R3D_ERR_ARG;
    }
    const bool needs_param = a->cfg.embed_dim > 0 || (b && b->cfg.embed_dim > 0);
    if (needs_param && !in->param_dev) { set_error("param_dev is required when the camera embedding is on"); return R3D_ERR_ARG; }
    if (in->window_stride <= 0) { set_error("window_stride must be positive"); return R3D_ERR_ARG; }
    if (((B - 1) * in->window_stride + a->RF) * (int64_t)(a->cfg.num_joints * 3) * 4 >= 0x7fffffffLL || B * (int64_t)(a->RF / 3) >= 0x7fffffffLL) {
        set_error("B too large for one call (the raw input must stay below 2 GiB)");
        return R3D_ERR_ARG;
    }

    Plan *pl = plan_get(a, b, plan_kind(B));
    const long long frames = (B - 1) * in->window_stride + a->RF;
    const size_t need = workspace_need(pl, B);
    if (!ws || ws_bytes < need) {
        set_error("workspace too small: need %zu bytes, got %zu", need, ws_bytes);
        return R3D_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_v;
    // R3D_OPT_LANES: which lane runs this forward - the one whose stream the caller passed, or the next one round-robin (then the
    // lane's stream waits for the caller's, runs the forward, and the caller's stream joins later: r3d_lanes_join)
    const int lanes = a->lanes;
    if (b && b->lanes != lanes) { set_error("pos and trj handles disagree on R3D_OPT_LANES (%d / %d): set it on both", a->lanes, b->lanes); return R3D_ERR_STATE; }
    int lane = 0;
    Model::Lane *relay = nullptr;          // round-robin: the lane this call is relayed to
    if (lanes > 1) {
        lane = -1;
        for (int k = 0; k < lanes; ++k)
            if (a->lane[k].stream == stream) lane = k;
        if (lane < 0) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
                (void)hipGetLastError();
                set_error("a handle with R3D_OPT_LANES cannot be captured from a caller's stream: capture on a lane's own stream (r3d_lane_stream)");
                return R3D_ERR_STATE;
            }
            lane = a->next_lane;
            a->next_lane = (a->next_lane + 1) % lanes;
            relay = &a->lane[lane];
            hipError_t e0;
            // (the lane is in order: a forward still pending on it for another stream simply runs first)
            // (a caller on the legacy default stream: the lanes' streams are blocking ones and behind its work as they are - an
            //  event recorded there would also be behind the other lanes' forwards, and the lanes would take turns)
            if (stream != nullptr &&
                ((e0 = hipEventRecord(relay->in, stream)) != hipSuccess || (e0 = hipStreamWaitEvent(relay->stream, relay->in, 0)) != hipSuccess))
                return hip_fail(e0, "hipStreamWaitEvent(lane)");
            stream = relay->stream;
        }
        lane += 1;                         // schedules of lane k live under key k + 1 (0: the handle without lanes)
    }
    float *act_base = (float *)ws;         // (poll mode: the schedule's own activation bank of this call)
    auto buf_ptr = [&](int id) -> float * { return act_base + (size_t)pl->buffers[id].offset_per_window * (size_t)B; };
    Recorder rec{a, stream};
    hipError_t e;
    int stage_no = 0;
    // profiling: an empty bracket first - what two event records cost by themselves on this stream (stage -1)
    if ((e = rec.begin("r3d_event_pair", -1, 0, 0.0, 0.0)) != hipSuccess) return hip_fail(e, "hipEventRecord");
    if ((e = rec.end()) != hipSuccess) return hip_fail(e, "hipEventRecord");

    // UV mode: the kernels read pixel keypoints (frames, J, 2) and encode the rays while gathering them
    const bool uv = in->mode == R3D_INPUT_UV;
    Bases bases;
    bases.ws = reinterpret_cast<const char *>(ws);
    for (int mi = 0; mi < 2; ++mi)
        if (pl->m[mi]) {
            bases.arena[mi] = reinterpret_cast<const char *>(pl->m[mi]->d_arena);
            bases.iarena[mi] = reinterpret_cast<const char *>(pl->m[mi]->d_iarena);
        }
    bases.x = reinterpret_cast<const char *>(in->x_dev);
    bases.param = reinterpret_cast<const char *>(in->param_dev);
    bases.cam = reinterpret_cast<const char *>(in->cam_dev);
    CallShape shape;
    shape.uv = uv;
    shape.window_stride = in->window_stride;
    shape.param_stride = in->param_stride;
    shape.cam_stride = in->cam_stride;
    shape.frames = frames;

    // (R3D_OPT_CU_LIMIT: a CU-masked stream - fewer workgroups, and no ordering against other streams' forwards below)
    const int cu_limit = lanes > 1 ? device_cu_count() / lanes : std::max(a->cu_limit, b ? b->cu_limit : 0);
    Schedule *sched = schedule_get(pl, B, cu_limit > 0 ? std::min(cu_limit, device_cu_count()) : device_cu_count(), false, lane);
    if (!sched) return R3D_ERR_HIP;
    const unsigned *abort_flag = nullptr;
    // Clip calls (window stride one frame, lib/train_val/trainer.py:47-58): consecutive windows share all but one of their
    // frames, and expand_conv is linear - its pre-activations are evaluated once per FRAME by a launch of gathered GEMMs
    // ahead of the forward (Plan::frame_probs) and the first-level tiles read them instead of gathering and multiplying
    // (r3d_kernels.hip, first_level_shared).  Where it pays (four times fewer rows), one camera for the
    // clip, fp32 tiles.
    bool b3_call = false;
    for (const Model *mm : pl->m) b3_call = b3_call || (mm && mm->use_b3 && B >= b3_min_batch());
    // (the per-frame buffer is addressed with 32-bit byte offsets in first_level_shared - row tables, descriptor bound: a
    //  clip whose buffer would reach 4 GiB, ~349 k windows for a pos + trj pair, keeps the gathered path, which has 64-bit tile bases)
    const bool shared = pl->frame_buf >= 0 && sched->d_frame_tiles != nullptr && in->window_stride == 1 && !b3_call &&
                        (frames - 2) * 4 <= B * (int64_t)(a->RF / 3) && !(uv && in->cam_stride != 0) && !hook_on("R3D_NO_SHARED_L0") &&
                        (unsigned long long)frames * (unsigned long long)pl->frame_ld * 4ull < 0xffffffffull;
    shape.shared = shared;
    const int variant = (uv ? 1 : 0) + (shared ? 2 : 0);
    const bool single = forward_single_launch() && !a->opt_staged && !(b && b->opt_staged) && sched->fwd.grid > 0 && sched->fwd.d_rel[variant] != nullptr &&
                        !(shared && sched->fwd.kernel != FWD_KERNEL_F32);
    if (shared) {
        const StageSchedule &fs = sched->frame_stage;
        LaunchArgs la;
        memset(&la, 0, sizeof la);
        la.tiles = sched->d_frame_tiles;
        la.wg_off = sched->d_frame_wgoff;
        la.nprob = (int)pl->frame_probs.size();
        la.ks = fs.ks;
        const int JF = a->cfg.num_joints * (uv ? 2 : a->cfg.in_features);
        float *fbase = reinterpret_cast<float *>(ws) + (size_t)pl->buffers[pl->frame_buf].offset_per_window * (size_t)B;
        for (int i = 0; i < la.nprob; ++i) {
            const Plan::FrameProb &f = pl->frame_probs[i];
            const Model *mm = pl->m[f.model];
            const Layer &L = mm->layers[f.layer];
            GemmProb &g = la.p[i];
            for (int sg = 0; sg < MAX_SEG; ++sg) g.kend[sg] = 0x7fffffff;
            g.w = mm->d_arena + L.w_off;
            g.bias = mm->d_arena + L.b_off;
            g.c = fbase + f.col;
            g.ldc = pl->frame_ld;
            g.M = (int)(frames - 2);
            g.N = L.N;
            g.K = L.Kpad;
            g.slope = 1.0f;
            g.lut = mm->d_iarena + (uv ? f.lut_uv : f.lut);
            g.x = reinterpret_cast<const float *>(in->x_dev);
            g.cam = uv ? reinterpret_cast<const double *>(in->cam_dev) : nullptr;
            g.cam_stride = 0;
            g.enc_ws = 0;
            g.enc_rows = g.M;                    // (one "window": operand row r starts at frame r)
            g.enc_step = 1;
            g.enc_jf = JF;
            g.enc_cur = 0;
            g.enc_bytes = (unsigned)((size_t)frames * JF * sizeof(float));
            g.res_tap = 1;
        }
        if ((e = rec.begin(uv ? "r3d_gemm_uv_f32" : "r3d_gemm_f32", stage_no, fs.nwg, 0.0, 0.0)) != hipSuccess) return hip_fail(e, "hipEventRecord");
        if ((e = launch_gemm_stage(la, fs.nwg, STAGE_BIG, uv, stream)) != hipSuccess) return hip_fail(e, "launch r3d_gemm_f32 (per-frame first layers)");
        if ((e = rec.end()) != hipSuccess) return hip_fail(e, "hipEventRecord");
        ++stage_no;
    }
    if (single) {
        // ---- the whole forward as ONE persistent launch: bind (zero the ready counters, resolve the problem table), run
        Schedule::Fwd &fw = sched->fwd;
        // Control region: the caller's workspace while the stream is being captured (the graph binds for itself), the
        // schedule's own otherwise - there a call on the buffers of the previous one finds the table bound and a zeroed
        // bank of counters, and skips r3d_bind_f32 (4-5 us per call; R3D_BIND_ALWAYS=1: never)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
        const bool own = cap == hipStreamCaptureStatusNone && fw.d_ctrl != nullptr;
        char *ctrl = own ? fw.d_ctrl : reinterpret_cast<char *>(ws) + workspace_act_bytes(pl, B);
        const size_t bank_bytes = ((size_t)(fw.ncnt + 4) * sizeof(unsigned) + 255) / 256 * 256;
        GemmProb *tables = reinterpret_cast<GemmProb *>(ctrl + (own ? 2 : 1) * bank_bytes);   // (own: one table per activation bank)
        // calls of a few windows, not captured: activations in the schedule's own two banks, data as its own ready flag
        const bool poll = own && fw.d_act != nullptr;
        BindArgs ba;
        memset(&ba, 0, sizeof ba);
        ba.rel = fw.d_rel[variant];
        ba.tags = fw.d_tags[variant];
        ba.out = tables;
        ba.nprob = fw.nprob;
        ba.base[BIND_WS] = ws;
        ba.base[BIND_ARENA0] = bases.arena[0];
        ba.base[BIND_ARENA1] = bases.arena[1];
        ba.base[BIND_IARENA0] = bases.iarena[0];
        ba.base[BIND_IARENA1] = bases.iarena[1];
        ba.base[BIND_X] = in->x_dev;
        ba.base[BIND_PARAM] = in->param_dev;
        ba.base[BIND_CAM] = in->cam_dev;
        const int JF = a->cfg.num_joints * (uv ? 2 : a->cfg.in_features);
        ba.enc_ws = in->window_stride * JF;
        ba.cam_stride = in->cam_stride;
        ba.enc_bytes = (unsigned)((size_t)frames * JF * sizeof(float));
        ba.param_stride = (int)in->param_stride;
        Schedule::Fwd::Bound &bd = fw.bound;
        bool bound = own && bd.valid && bd.uv == variant && bd.enc_ws == ba.enc_ws && bd.cam_stride == ba.cam_stride &&
                     bd.enc_bytes == ba.enc_bytes && bd.param_stride == ba.param_stride;
        for (int k = 0; bound && k < BIND_NBASE; ++k) bound = bd.base[k] == ba.base[k];
        const int bank = bound ? bd.bank ^ 1 : 0;
        unsigned *cnt = reinterpret_cast<unsigned *>(ctrl + (own ? bank : 0) * bank_bytes);
        // (handles of different threads: the waits below, the launch and the note of whose forward was last are one critical section -
        //  two threads that both passed the waits before either had launched would put two whole-device forwards on the chip together)
        std::unique_lock<std::mutex> launch_lock(g_fwd_launch_mu);
        if ((e = order_single_launch(stream, true, cu_limit > 0)) != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");
        // one handle on two masked streams: its counter banks and control region are one per handle - the second stream waits for the first
        if (cu_limit > 0 && lanes <= 1 && a->last_fwd_stream && a->last_fwd_stream != stream) {
            if ((e = wait_behind(stream, (hipStream_t)a->last_fwd_stream, a->order_ev)) != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");
        }
        a->last_fwd_stream = stream;
        if (!bound) {
            ba.cnt = reinterpret_cast<unsigned *>(ctrl);
            ba.ncnt = own ? (int)(2 * bank_bytes / sizeof(unsigned)) - 4 : fw.ncnt;      // (the kernel zeroes ncnt + 4 words: both banks)
            if (poll) {                 // both activation banks armed, bank 0's table
                ba.base[BIND_WS] = fw.d_act;
                ba.arm = fw.d_act;
                ba.arm_vec4 = (long long)(2 * fw.act_bytes / 16);
            }
            if ((e = rec.begin("r3d_bind_f32", stage_no, 1, 0.0, 0.0)) != hipSuccess) return hip_fail(e, "hipEventRecord");
            if ((e = launch_bind(ba, stream)) != hipSuccess) return hip_fail(e, "launch r3d_bind_f32");
            if (poll) {                 // ... and bank 1's
                BindArgs b1 = ba;
                b1.base[BIND_WS] = fw.d_act + fw.act_bytes;
                b1.out = tables + fw.nprob;
                b1.ncnt = -4;
                b1.arm = nullptr;
                if ((e = launch_bind(b1, stream)) != hipSuccess) return hip_fail(e, "launch r3d_bind_f32");
            }
            if ((e = rec.end()) != hipSuccess) return hip_fail(e, "hipEventRecord");
            ++stage_no;
            ba.base[BIND_WS] = ws;      // (the key of the bound state: what the caller passed)
        }
        const GemmProb *table = tables + (poll ? bank * fw.nprob : 0);
        if (poll) act_base = reinterpret_cast<float *>(fw.d_act + (size_t)bank * fw.act_bytes);
        bd.valid = false;          // (until this call's launch is on the stream: it is what zeroes the bank the next call runs on)
        FwdArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.tiles = fw.d_tiles;
        fa.wg_off = fw.d_wgoff;
        fa.probs = table;
        fa.cnt = cnt;
        fa.cnt_next = own ? reinterpret_cast<unsigned *>(ctrl + (bank ^ 1) * bank_bytes) : nullptr;
        fa.ncnt = fw.ncnt;
        if (poll) {
            fa.poll = 1;
            fa.arm = fw.d_act + (size_t)(bank ^ 1) * fw.act_bytes;
            fa.arm_vec4 = (long long)(fw.act_bytes / 16);
        }
        fa.spin_ticks = (long long)std::max(a->spin_timeout_ms, 1) * 100000LL;          // 100 MHz wall clock
        if (const char *ft = hook_env("R3D_FAULT_TILE")) fa.fault_tile1 = atoi(ft) + 1;   // (hooks build only: see FwdArgs)
        const bool uv_launch = uv && fw.uses_gather;
        int fwd_kernel = shared ? FWD_KERNEL_CLIP : fw.kernel;           // (shared: fw.kernel is FWD_KERNEL_F32 - `single` above)
        if (fwd_kernel == FWD_KERNEL_F32 && !uv_launch && hook_on("R3D_CHAIN")) fwd_kernel = FWD_KERNEL_CHAIN;   // (experiment: fill_prob has set wchain)
        if ((e = rec.begin(forward_kernel_name(fwd_kernel, uv_launch), stage_no, fw.grid, fw.flops, fw.bytes)) != hipSuccess)
            return hip_fail(e, "hipEventRecord");
#ifdef R3D_TIMING
        static long long *timing_buf1 = nullptr;
        if (getenv("R3D_TIMING_STAGE")) {
            const size_t tbytes = (16384 + 4 * 65536) * 8;
            if (!timing_buf1) (void)hipMalloc((void **)&timing_buf1, tbytes);
            (void)hipMemsetAsync(timing_buf1, 0, tbytes, stream);
            if (fw.ntiles <= 65536) fa.dbg = timing_buf1;
        }
#endif
        if ((e = launch_forward(fa, fw.grid, fwd_kernel, uv_launch, stream)) != hipSuccess) return hip_fail(e, "launch r3d_forward_f32");
        a->last_clk_dev = cap == hipStreamCaptureStatusNone ? cnt + fw.ncnt + 2 : nullptr;   // (a captured call runs later, maybe never)
        if ((e = order_single_launch(stream, false, cu_limit > 0)) != hipSuccess) return hip_fail(e, "hipEventRecord");
        launch_lock.unlock();
        if (own) {                     // the next call on these buffers needs no bind
            bd.valid = true;
            bd.bank = bank;
            bd.uv = variant;
            bd.enc_ws = ba.enc_ws;
            bd.cam_stride = ba.cam_stride;
            bd.enc_bytes = ba.enc_bytes;
            bd.param_stride = ba.param_stride;
            for (int k = 0; k < BIND_NBASE; ++k) bd.base[k] = ba.base[k];
        }
#ifdef R3D_TIMING
        if (fa.dbg) {
            (void)hipStreamSynchronize(stream);
            std::vector<long long> hw(4 * 1024);
            (void)hipMemcpy(hw.data(), timing_buf1 + 1024, hw.size() * 8, hipMemcpyDeviceToHost);
            long long w0 = 1LL << 62, w1 = 0, e0 = 1LL << 62;
            std::vector<double> d;
            for (int w = 0; w < fw.grid && w < 1024; ++w) {
                w0 = std::min(w0, hw[w * 4 + 2]); w1 = std::max(w1, hw[w * 4 + 3]); e0 = std::min(e0, hw[w * 4 + 3]);
                d.push_back((hw[w * 4 + 3] - hw[w * 4 + 2]) / 100.0);
            }
            std::sort(d.begin(), d.end());
            fprintf(stderr, "[timing] forward: first workgroup start -> last end %.2f us; ends spread over %.2f us; busy min %.1f median %.1f max %.1f us\n",
                    (w1 - w0) / 100.0, (w1 - e0) / 100.0, d.front(), d[d.size() / 2], d.back());
            if (getenv("R3D_TIMING_ALL"))
                for (int w = 0; w < fw.grid && w < 1024; ++w)
                    fprintf(stderr, "[timing-wg] %d start %.2f end %.2f\n", w, (hw[w * 4 + 2] - w0) / 100.0, (hw[w * 4 + 3] - w0) / 100.0);
            if (const char *dump = getenv("R3D_TIMING_DUMP")) {      // every tile: who ran it, what it is, fetched / ready / finished
                std::vector<long long> tt((size_t)fw.ntiles * 4);
                (void)hipMemcpy(tt.data(), timing_buf1 + 16384, tt.size() * 8, hipMemcpyDeviceToHost);
                if (FILE *f = fopen(dump, "w")) {
                    for (int i = 0; i < fw.nprob; ++i) {
                        const ProbSpec &q = pl->probs[i];
                        fprintf(f, "P %d %s rows_per_window %d M %lld N %d K %d fused %d\n", i, pl->m[q.model]->layers[q.layer].weight_key.c_str(),
                                q.rows_per_window, (long long)(B * q.rows_per_window), pl->m[q.model]->layers[q.layer].N,
                                pl->m[q.model]->layers[q.layer].Kpad, q.layer3 >= 0 ? 3 : q.layer2 >= 0 ? 2 : 1);
                    }
                    for (int w = 0; w < fw.grid; ++w)
                        for (int t = fw.h_wgoff[w]; t < fw.h_wgoff[w + 1]; ++t) {
                            const int *d = &fw.h_tiles[(size_t)t * FWD_TILE_INT4 * 4];
                            fprintf(f, "T %d %d %d %d %d %d %d %d %.2f %.2f %.2f %lld\n", w, t, d[0] & 0xff, d[0] >> 8, d[1], d[2], d[3], d[4],
                                    tt[(size_t)t * 4] ? (tt[(size_t)t * 4] - w0) / 100.0 : -1.0, tt[(size_t)t * 4 + 1] ? (tt[(size_t)t * 4 + 1] - w0) / 100.0 : -1.0,
                                    tt[(size_t)t * 4 + 2] ? (tt[(size_t)t * 4 + 2] - w0) / 100.0 : -1.0, tt[(size_t)t * 4 + 3]);
                        }
                    fclose(f);
                }
            }
        }
#endif
        if ((e = rec.end()) != hipSuccess) return hip_fail(e, "hipEventRecord");
        ++stage_no;
        abort_flag = cnt + fw.ncnt;
    }
    if (!single) a->last_clk_dev = nullptr;
    // ---- (staged form) persistent GEMM launches, one per DAG level
    for (size_t si = 0; !single && si < sched->levels->size(); ++si) {
        const auto &st = (*sched->levels)[si];
        const StageSchedule &ss = sched->stages[si];
        LaunchArgs la;
        memset(&la, 0, sizeof la);
        la.tiles = sched->d_tiles + ss.tiles_off;
        la.wg_off = sched->d_wgoff + ss.wgoff_off;
        la.nprob = (int)st.size();
        la.ks = ss.ks;
        int n_enc = 0;
        for (int i = 0; i < la.nprob; ++i) {
            const ProbSpec &q = pl->probs[st[i] & ~STAGE_SPILL_IN];      // (a spilled tail uses the problem as it is: tiles carry absolute rows)
            if (q.enc_lut >= 0 && q.enc_kernel) ++n_enc;                 // (with a fused first level these run in the GEMM kernel)
            const int rc = fill_prob(pl, q, B, a, bases, shape, la.p[i], nullptr);
            if (rc != R3D_OK) return rc;
        }
        if (n_enc != 0 && n_enc != la.nprob) { set_error("internal: launch mixes encoded and plain operands"); return R3D_ERR_STATE; }
        if ((n_enc != 0) != (ss.kind == STAGE_ENC)) { set_error("internal: schedule and plan disagree on the launch kind"); return R3D_ERR_STATE; }
        bool uv_launch = false;                             // UV mode: only the launches that gather from the input
        for (int i = 0; i < la.nprob; ++i) uv_launch = uv_launch || la.p[i].cam != nullptr;
        bool b3_launch = false;                             // (launch_gemm_stage picks the kernel by the same test)
        for (int i = 0; i < la.nprob; ++i) b3_launch = b3_launch || la.p[i].wb3 != nullptr;
        const char *kname = ss.kind == STAGE_ENC ? (uv_launch ? "r3d_gemm_enc_uv_f32" : "r3d_gemm_enc_f32")
                          : b3_launch ? (uv_launch ? "r3d_gemm_uv_b3" : "r3d_gemm_b3") : (uv_launch ? "r3d_gemm_uv_f32" : "r3d_gemm_f32");
        if ((e = rec.begin(kname, stage_no, ss.nwg, ss.flops, ss.bytes)) != hipSuccess) return hip_fail(e, "hipEventRecord");
#ifdef R3D_TIMING
        // development build only (tools/build_probe.sh): phase stamps of the first tiles of launch $R3D_TIMING_STAGE
        static long long *timing_buf = nullptr;
        const char *tstage = getenv("R3D_TIMING_STAGE");
        const bool timed = tstage && (!strcmp(tstage, "all") || atoi(tstage) == (int)si);
        if (timed) {
            if (!timing_buf) (void)hipMalloc((void **)&timing_buf, (1024 + 4 * 1024) * 8 + 65536);
            (void)hipMemsetAsync(timing_buf, 0, (1024 + 4 * 1024) * 8 + 65536, stream);
            la.dbg = timing_buf;
        }
#endif
        if ((e = launch_gemm_stage(la, ss.nwg, ss.kind, uv_launch, stream)) != hipSuccess) return hip_fail(e, "launch r3d_gemm_f32");
#ifdef R3D_TIMING
        if (timed) {
            (void)hipStreamSynchronize(stream);
            std::vector<long long> ht(16 * 64);
            (void)hipMemcpy(ht.data(), timing_buf + 6144, ht.size() * 8, hipMemcpyDeviceToHost);
            std::vector<long long> hw(4 * 1024);
            (void)hipMemcpy(hw.data(), timing_buf + 1024, hw.size() * 8, hipMemcpyDeviceToHost);
            long long w0 = 1LL << 62, w1 = 0, s1 = 0, e0 = 1LL << 62;
            for (int w = 0; w < ss.nwg && w < 1024; ++w) {
                w0 = std::min(w0, hw[w * 4 + 2]); s1 = std::max(s1, hw[w * 4 + 2]);
                w1 = std::max(w1, hw[w * 4 + 3]); e0 = std::min(e0, hw[w * 4 + 3]);
            }
            fprintf(stderr, "[timing] launch %zu: first workgroup start -> last end %.2f us; starts spread over %.2f us, ends over %.2f us; "
                    "wg 0: start -> first tile entry %.2f us\n", si, (w1 - w0) / 100.0, (s1 - w0) / 100.0, (w1 - e0) / 100.0,
                    (ht[0] - hw[2]) / 100.0);
            {   // distribution of the workgroups' busy times (start -> end of the persistent loop)
                std::vector<double> d;
                for (int w = 0; w < ss.nwg && w < 1024; ++w) d.push_back((hw[w * 4 + 3] - hw[w * 4 + 2]) / 100.0);
                std::sort(d.begin(), d.end());
                if (!d.empty())
                    fprintf(stderr, "[timing] launch %zu: workgroup busy time min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us (%zu workgroups)\n", si,
                            d.front(), d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10], d.back(), d.size());
                {   // shader clock during the launch: cycle counter against the 100 MHz wall clock, median over the workgroups
                    std::vector<double> g;
                    for (int w = 0; w < ss.nwg && w < 1024; ++w)
                        if (hw[w * 4 + 3] > hw[w * 4 + 2]) g.push_back((double)(hw[w * 4 + 1] - hw[w * 4 + 0]) / ((hw[w * 4 + 3] - hw[w * 4 + 2]) * 10.0));
                    std::sort(g.begin(), g.end());
                    if (!g.empty()) fprintf(stderr, "[timing] launch %zu: shader clock %.2f GHz (median), %.2f .. %.2f\n", si, g[g.size() / 2], g.front(), g.back());
                }
                if (getenv("R3D_TIMING_ALL"))
                    for (int w = 0; w < ss.nwg && w < 1024; ++w)
                        fprintf(stderr, "[timing-wg] %d start %.2f end %.2f\n", w, (hw[w * 4 + 2] - w0) / 100.0, (hw[w * 4 + 3] - w0) / 100.0);
            }
            fprintf(stderr, "[timing] launch %zu: wg tile | phase lengths in us (100 MHz wall clock)\n", si);
            for (int w = 0; w < 16 && w < ss.nwg; ++w)
                for (int t = 0; t < 8; ++t) {
                    const long long *q = &ht[w * 64 + t * 8];
                    if (!q[0]) continue;
                    fprintf(stderr, "  wg %2d tile %d: %6.2f %6.2f %6.2f %6.2f | total %6.2f", w, t, (q[1] - q[0]) / 100.0,
                            (q[2] - q[1]) / 100.0, (q[3] - q[2]) / 100.0, (q[4] - q[3]) / 100.0, (q[4] - q[0]) / 100.0);
                    if (q[5]) fprintf(stderr, " | first tap: expand %6.2f  H write %6.2f  3-tap third %6.2f", (q[5] - q[0]) / 100.0,
                                      (q[6] - q[5]) / 100.0, (q[7] - q[6]) / 100.0);
                    fprintf(stderr, "\n");
                }
        }
#endif
        if ((e = rec.end()) != hipSuccess) return hip_fail(e, "hipEventRecord");
        ++stage_no;
    }

    // ---- fused decoder tail
    DecodeArgs da;
    memset(&da, 0, sizeof da);
    da.B = B;
    da.J = a->cfg.num_joints;
    da.has_pos = pl->pos_model >= 0;
    da.has_trj = pl->trj_model >= 0;
    da.out = out;
    da.out_trj = da.has_pos ? out_trj : nullptr;
    da.abort_flag = abort_flag;
    da.status = a->status_host;
    double dec_flops = 0;
    int first = 0;
    // plan.decs lists the pos parts (Torso, LArm, RArm, LLeg, RLeg) then the trajectory decoder
    std::vector<Plan::Dec> order;
    for (const auto &d : pl->decs) if (pl->m[d.model]->cfg.kind == R3D_KIND_POS) order.push_back(d);
    for (const auto &d : pl->decs) if (pl->m[d.model]->cfg.kind == R3D_KIND_TRJ) order.push_back(d);
    int firsts[MAX_DEC] = {0};
    for (const auto &d : order) {
        const Model *m = pl->m[d.model];
        const Layer &L = m->layers[d.layer];
        const int sidx = da.nsrc++;
        da.h[sidx] = buf_ptr(d.hbuf);
        da.w[sidx] = m->d_arena + L.w_off;
        da.bias[sidx] = m->d_arena + L.b_off;
        da.n_out[sidx] = L.N;
        da.first[sidx] = first;
        firsts[sidx] = first;
        if (m->cfg.kind == R3D_KIND_POS) first += L.N;
        dec_flops += 2.0 * (double)B * L.K * L.N;
    }
    if (da.has_pos) output_slots(da.J, firsts, da.slot);
    if ((e = rec.begin(B >= 128 ? "r3d_decode_w4_f32" : "r3d_decode_f32", stage_no, 0, dec_flops, (double)B * da.nsrc * MLP_HIDDEN * 4.0)) != hipSuccess) return hip_fail(e, "hipEventRecord");
    if ((e = launch_decode(da, stream)) != hipSuccess) return hip_fail(e, "launch r3d_decode_f32");
    if ((e = rec.end()) != hipSuccess) return hip_fail(e, "hipEventRecord");
    if (rec.on()) a->nrec = (int)rec.n;
    if (relay) {                           // the caller's stream sees the outputs once it has joined this lane (r3d_lanes_join)
        if ((e = hipEventRecord(relay->done, relay->stream)) != hipSuccess) return hip_fail(e, "hipEventRecord(lane)");
        if (std::find(relay->waiters.begin(), relay->waiters.end(), stream_v) == relay->waiters.end()) relay->waiters.push_back(stream_v);
    }
    return R3D_OK;
}

// CU mask of lane k of n: CUs c of every XCD with c % n == k (mask bit i is CU i / 8 of XCD i % 8: consecutive bits go to consecutive
// XCDs) - every lane spans all eight XCDs and their L2s with device CUs / n CUs
static int lanes_create(Model *m, int n) {
    const int cus = device_cu_count();
    const int words = (cus + 31) / 32;
    for (int k = 0; k < n; ++k) {
        std::vector<uint32_t> mask((size_t)std::max(words, 1), 0u);
        for (int i = 0; i < cus; ++i)
            if ((i / 8) % n == k) mask[(size_t)i / 32] |= 1u << (i % 32);
        hipError_t e = hipExtStreamCreateWithCUMask(&m->lane[k].stream, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) return hip_fail(e, "hipExtStreamCreateWithCUMask");
        if ((e = hipEventCreateWithFlags(&m->lane[k].done, hipEventDisableTiming)) != hipSuccess) return hip_fail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&m->lane[k].in, hipEventDisableTiming)) != hipSuccess) return hip_fail(e, "hipEventCreate");
    }
    return R3D_OK;
}
void lanes_destroy(Model *m) {
    for (auto &ln : m->lane) {
        if (ln.done) (void)hipEventDestroy(ln.done);
        if (ln.in) (void)hipEventDestroy(ln.in);
        if (ln.stream) {
            order_forget(ln.stream);
            (void)hipStreamSynchronize(ln.stream);
            (void)hipStreamDestroy(ln.stream);
        }
        ln = Model::Lane();
    }
    m->next_lane = 0;
}

}  // namespace r3d

using namespace r3d;

extern "C" {

int r3d_create(const r3d_config *cfg, r3d_model **out) {
    if (!cfg || !out) { set_error("r3d_create: null argument"); return R3D_ERR_ARG; }
    if (cfg->struct_size != (int32_t)sizeof(r3d_config)) {
        set_error("r3d_create: r3d_config.struct_size is %d, this library's r3d_config has %d bytes (ABI version %d): the "
                  "binding was written against another include/ray3d_hip.h", cfg->struct_size, (int)sizeof(r3d_config), R3D_ABI_VERSION);
        return R3D_ERR_ARG;
    }
    Model *m = model_create(*cfg);
    if (!m) return R3D_ERR_ARG;
    *out = reinterpret_cast<r3d_model *>(m);
    return R3D_OK;
}

int r3d_destroy(r3d_model *m) {
    delete reinterpret_cast<Model *>(m);
    return R3D_OK;
}

int r3d_num_weights(const r3d_model *m) {
    return m ? (int)reinterpret_cast<const Model *>(m)->specs.size() : R3D_ERR_ARG;
}

const char *r3d_weight_key(const r3d_model *m, int index) {
    const Model *mm = reinterpret_cast<const Model *>(m);
    if (!mm || index < 0 || index >= (int)mm->specs.size()) return nullptr;
    return mm->specs[index].key.c_str();
}

int r3d_weight_shape(const r3d_model *m, int index, int64_t shape[4], int *rank) {
    const Model *mm = reinterpret_cast<const Model *>(m);
    if (!mm || !shape || !rank || index < 0 || index >= (int)mm->specs.size()) { set_error("r3d_weight_shape: bad argument"); return R3D_ERR_ARG; }
    *rank = mm->specs[index].rank;
    for (int i = 0; i < 4; ++i) shape[i] = mm->specs[index].shape[i];
    return R3D_OK;
}

int r3d_set_weight(r3d_model *m, const char *key, const float *host, const int64_t *shape, int rank) {
    return model_set_weight(reinterpret_cast<Model *>(m), key, host, shape, rank);
}

int r3d_finalize(r3d_model *m) { return model_finalize(reinterpret_cast<Model *>(m)); }

size_t r3d_workspace_bytes(const r3d_model *pos, const r3d_model *trj, int64_t B) {
    Model *p = const_cast<Model *>(reinterpret_cast<const Model *>(pos));
    Model *t = const_cast<Model *>(reinterpret_cast<const Model *>(trj));
    Model *a = p ? p : t, *b = p ? t : nullptr;
    if (!a || B <= 0) return 0;
    // monotonic in B: the plan kind switches with the window count and the less fused plans of small calls keep larger
    // intermediates, so a call of fewer windows may need MORE bytes than one of B - the answer covers every size <= B
    size_t need = workspace_need(plan_get(a, b, plan_kind(B)), B);
    for (int64_t edge : plan_kind_edges())
        if (edge < B) need = std::max(need, workspace_need(plan_get(a, b, plan_kind(edge)), edge));
    return need;
}

int r3d_prepare(r3d_model *pos, r3d_model *trj, int64_t B) {
    Model *p = reinterpret_cast<Model *>(pos), *t = reinterpret_cast<Model *>(trj);
    Model *a = p ? p : t, *b = p ? t : nullptr;
    if (!a || B <= 0) { set_error("r3d_prepare: no model given or B <= 0"); return R3D_ERR_ARG; }
    for (Model *m : {a, b})
        if (m && (!m->finalized || m->dirty)) { set_error("r3d_prepare called before r3d_finalize (or weights changed since)"); return R3D_ERR_STATE; }
    if (b && !same_input_shape(a, b)) { set_error("pos and trj models disagree on J / F / levels / extrinsic_dim"); return R3D_ERR_ARG; }
    if (a->lanes > 1) {                    // every lane's schedule of this size
        for (int k = 1; k <= a->lanes; ++k)
            if (!schedule_get(plan_get(a, b, plan_kind(B)), B, device_cu_count() / a->lanes, /*pin=*/true, k)) return R3D_ERR_HIP;
        return R3D_OK;
    }
    const int cu_limit = std::max(a->cu_limit, b ? b->cu_limit : 0);
    return schedule_get(plan_get(a, b, plan_kind(B)), B, cu_limit > 0 ? std::min(cu_limit, device_cu_count()) : device_cu_count(), /*pin=*/true) ? R3D_OK : R3D_ERR_HIP;
}

int r3d_release(r3d_model *pos, r3d_model *trj, int64_t B) {
    Model *p = reinterpret_cast<Model *>(pos), *t = reinterpret_cast<Model *>(trj);
    Model *a = p ? p : t, *b = p ? t : nullptr;
    if (!a || B <= 0) { set_error("r3d_release: no model given or B <= 0"); return R3D_ERR_ARG; }
    Plan *pl = plan_get(a, b, plan_kind(B));
    bool any = false;
    for (int k = 0; k <= 4; ++k) {         // (the handle without lanes: key 0; lane k: key k + 1)
        auto it = pl->schedules.find(schedule_key(B, k));
        if (it != pl->schedules.end() && it->second->pinned) { it->second->pinned = false; any = true; }
    }
    if (!any) { set_error("r3d_release: %lld windows were never prepared for this pair", (long long)B); return R3D_ERR_ARG; }
    return R3D_OK;
}

int r3d_forward(r3d_model *m, const r3d_input *in, int64_t B, float *out_dev, void *ws, size_t ws_bytes, void *stream) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm) { set_error("r3d_forward: null model"); return R3D_ERR_ARG; }
    return mm->cfg.kind == R3D_KIND_POS ? run(mm, nullptr, in, B, out_dev, nullptr, ws, ws_bytes, stream)
                                        : run(nullptr, mm, in, B, out_dev, nullptr, ws, ws_bytes, stream);
}

int r3d_forward_pair(r3d_model *pos, r3d_model *trj, const r3d_input *in, int64_t B, float *out_dev, float *out_trj_dev,
                     void *ws, size_t ws_bytes, void *stream) {
    Model *p = reinterpret_cast<Model *>(pos), *t = reinterpret_cast<Model *>(trj);
    if (!p || !t) { set_error("r3d_forward_pair: both models are required"); return R3D_ERR_ARG; }
    if (p->cfg.kind != R3D_KIND_POS || t->cfg.kind != R3D_KIND_TRJ) { set_error("r3d_forward_pair: (pos, trj) expected in that order"); return R3D_ERR_ARG; }
    return run(p, t, in, B, out_dev, out_trj_dev, ws, ws_bytes, stream);
}

int r3d_profile_enable(r3d_model *m, int on) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm) { set_error("r3d_profile_enable: null model"); return R3D_ERR_ARG; }
    mm->profiling = on != 0;
    mm->nrec = 0;
    return R3D_OK;
}

int r3d_profile_read(r3d_model *m, r3d_launch_record *records, int capacity) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm) { set_error("r3d_profile_read: null model"); return R3D_ERR_ARG; }
    int n = 0;
    for (int i = 0; i < mm->nrec; ++i) {
        Model::Rec &r = mm->recs[i];
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
        if ((e = hipEventElapsedTime(&r.r.ms, r.e0, r.e1)) != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
        if (records && n < capacity) records[n] = r.r;
        ++n;
    }
    return n;
}

// Test hook: build the static schedule of one launch on the host and verify
// that its tiles cover every (32-row unit, 32-column granule) of every problem exactly once within the
// kernel's tile-shape limits.  Returns 0 or a negative code naming the first violated rule.
#ifdef R3D_TEST_HOOKS      // (libray3d_hip_hooks.so only)
int r3d_debug_schedule_check(int nprob, const int *M, const int *N, const int *nk, const int *max_ks, const int *max_units,
                             int nwg, int enc, int *out_grid, int *out_tiles, double *out_imbalance) {
    std::vector<SchedProb> probs;
    for (int i = 0; i < nprob; ++i) {
        probs.push_back({M[i], N[i], nk[i], max_ks[i], max_units[i]});
        // (as sched_prob_of marks the plan's wide plain layers: their single-unit tiles may be 4 .. 7 column blocks wide)
        probs.back().nb_ok = !enc && N[i] % 32 == 0 && N[i] >= 512 && nk[i] >= 8 && max_units[i] == 0 && !hook_on("R3D_NO_NB");
    }
    std::vector<int4> tiles;
    std::vector<int> wgoff;
    StageSchedule ss{};
    schedule_stage(probs, nwg, GEMM_SCHED_MAX_UNITS, tiles, wgoff, ss, enc != 0);
    if (out_grid) *out_grid = ss.nwg;
    if (out_tiles) *out_tiles = ss.ntiles;
    if (out_imbalance) *out_imbalance = ss.imbalance;
    if (ss.nwg < 1 || ss.nwg > nwg) return -1;
    if ((int)wgoff.size() != ss.nwg + 1 || wgoff.front() != 0 || wgoff.back() != ss.ntiles || (int)tiles.size() != ss.ntiles) return -2;
    for (size_t i = 1; i < wgoff.size(); ++i)
        if (wgoff[i] <= wgoff[i - 1] && ss.ntiles > 0) return -3;          // empty or unordered chunk
    std::vector<std::vector<int>> cover(nprob);
    for (int i = 0; i < nprob; ++i) cover[i].assign((size_t)((M[i] + 31) / 32) * ((N[i] + COL_GRANULE - 1) / COL_GRANULE), 0);
    for (const int4 &t : tiles) {
        const int pi = t.x & 0xff, mi = t.x >> 8, ks = t.w;
        if (pi >= nprob || mi < 1 || (ks != 1 && ks != 2 && ks != 4 && ks != 8 && ks != 16 && !(ks >= NB_CODE + 4 && ks <= NB_CODE + 7))) return -4;
        if (ks < 8 && ks > max_ks[pi]) return -5;
        if ((ks == 1 && mi > (max_units[pi] > 0 ? std::min(max_units[pi], GEMM_SCHED_MAX_UNITS) : GEMM_SCHED_MAX_UNITS)) || (ks == 2 && mi > 2) || (ks >= 4 && mi != 1)) return -6;
        if (t.y % 32 || t.y < 0 || t.y >= M[pi] || t.z % (tile_is_nb(ks) ? 32 : tile_width(ks)) || t.z < 0 || t.z >= N[pi]) return -7;
        if (tile_is_nb(ks) && t.z + tile_width(ks) > N[pi]) return -7;        // (narrow tiles cover whole blocks of existing columns)
        if (ks > 1 && ks < 8 && (nk[pi] + ks - 1) / ks < 2) return -8;
        const int gcols = (N[pi] + COL_GRANULE - 1) / COL_GRANULE;
        for (int u = t.y / 32; u < t.y / 32 + mi; ++u) {
            if (u * 32 >= M[pi]) return -9;
            for (int g = t.z / COL_GRANULE; g < (t.z + tile_width(ks)) / COL_GRANULE && g < gcols; ++g) ++cover[pi][(size_t)u * gcols + g];
        }
    }
    for (int i = 0; i < nprob; ++i)
        for (int c : cover[i])
            if (c != 1) return -10;
    return 0;
}

// Test hook: the whole forward's tile lists for `batch` windows on `nwg` CUs, built on the host (no device needed):
// every 32-row x 32-column cell of every problem must be computed exactly once over all launches, a problem's
// tiles must sit in launches that list it, and a consumer's launch must come after all of its producers' tiles.
// Returns 0, or a negative code; *spilled = rows of the first level that run one launch late.
int r3d_debug_plan_check(r3d_model *pos, r3d_model *trj, int64_t batch, int nwg, int *launches, int *spilled) {
    Model *a = reinterpret_cast<Model *>(pos ? pos : trj), *b = reinterpret_cast<Model *>(pos && trj ? trj : nullptr);
    if (!a) return -1;
    Plan *pl = plan_get(a, b, plan_kind(batch));
    if (!pl) return -2;
    int spill_row0 = -1;
    std::vector<int4> tiles;
    std::vector<int> wgoff;
    std::vector<StageSchedule> stages;
    const std::vector<std::vector<int>> &levels = *schedule_build_host(pl, batch, nwg, spill_row0, tiles, wgoff, stages);
    if (launches) *launches = (int)stages.size();
    if (spilled) *spilled = 0;
    const int np = (int)pl->probs.size();
    std::vector<std::vector<int>> cover(np);
    std::vector<int> last_launch(np, -1), first_launch(np, 1 << 30);
    for (int i = 0; i < np; ++i) {
        const ProbSpec &q = pl->probs[i];
        const int M = (int)(batch * q.rows_per_window), N = pl->m[q.model]->layers[q.layer].N;
        cover[i].assign((size_t)((M + 31) / 32) * ((N + COL_GRANULE - 1) / COL_GRANULE), 0);
    }
    for (size_t si = 0; si < stages.size(); ++si) {
        const StageSchedule &ss = stages[si];
        const auto &st = levels[si];
        if (ss.nwg < 1 || ss.nwg > 2 * nwg) return -3;
        {   // a launch runs ONE kernel: its problems are all r3d_gemm_enc_f32's or none is
            int n_enc = 0;
            for (int e : st) n_enc += pl->probs[e & ~STAGE_SPILL_IN].enc_kernel ? 1 : 0;
            if (n_enc != 0 && n_enc != (int)st.size()) return -11;
            if ((n_enc != 0) != (ss.kind == STAGE_ENC)) return -12;
        }
        for (int t = 0; t < ss.ntiles; ++t) {
            const int4 &tl = tiles[ss.tiles_off + t];
            const int slot = tl.x & 0xff, mi = tl.x >> 8, ks = tl.w;
            if (slot >= (int)st.size() || mi < 1) return -4;
            const int id = st[slot] & ~STAGE_SPILL_IN;
            const ProbSpec &q = pl->probs[id];
            const int M = (int)(batch * q.rows_per_window), N = pl->m[q.model]->layers[q.layer].N;
            if (tl.y % 32 || tl.y < 0 || tl.y >= M || tl.z < 0 || tl.z >= N) return -5;
            if (id == pl->spill_prob) {
                const bool late = (st[slot] & STAGE_SPILL_IN) != 0;
                if (spill_row0 < 0 ? late : (late != (tl.y >= spill_row0))) return -6;
                if (late && spilled) *spilled += std::min(mi * 32, M - tl.y);
            }
            const int gcols = (N + COL_GRANULE - 1) / COL_GRANULE;
            for (int u = tl.y / 32; u < tl.y / 32 + mi; ++u) {
                if (u * 32 >= M) return -7;
                for (int g = tl.z / COL_GRANULE; g < (tl.z + tile_width(ks)) / COL_GRANULE && g < gcols; ++g) ++cover[id][(size_t)u * gcols + g];
            }
            last_launch[id] = std::max(last_launch[id], (int)si);
            first_launch[id] = std::min(first_launch[id], (int)si);
        }
    }
    for (int i = 0; i < np; ++i) {
        for (int c : cover[i])
            if (c != 1) return -8;
        for (int d : pl->probs[i].deps)
            if (last_launch[d] >= first_launch[i]) return -9;
    }
    return 0;
}

// Test hook: the single-launch form of the forward for `batch` windows on `nwg` CUs, built and EXECUTED on the host as a
// dependency machine: a workgroup's next tile runs when the ready counters it waits for are full; every tile must get to
// run (no waiting cycle), every counter must end full, and - independently of the dependency ranges the scheduler wrote -
// at the moment a tile runs, every earlier problem that writes what the tile reads, or reads / writes what the tile
// writes (same buffer, overlapping columns), must be complete for the tile's windows.
// Returns 0 (or 1: this plan runs launch by launch, nothing to check), or a negative code.  For the plan of calls of a few
// windows also: every workspace element is written exactly once per call (what poll mode relies on).
int r3d_debug_forward_check(r3d_model *pos, r3d_model *trj, int64_t batch, int nwg, int *out_tiles, int *out_counters) {
    Model *a = reinterpret_cast<Model *>(pos ? pos : trj), *b = reinterpret_cast<Model *>(pos && trj ? trj : nullptr);
    if (!a) return -1;
    Plan *pl = plan_get(a, b, plan_kind(batch));
    int spill_row0 = -1;
    std::vector<int4> tiles;
    std::vector<int> wgoff;
    std::vector<StageSchedule> stages;
    const std::vector<std::vector<int>> &levels = *schedule_build_host(pl, batch, nwg, spill_row0, tiles, wgoff, stages);
    Schedule::Fwd fw;
    std::vector<int> ft, fo;
    if (!schedule_build_fwd(pl, batch, nwg, levels, stages, tiles, wgoff, fw, ft, fo)) return 1;
    if (out_tiles) *out_tiles = fw.ntiles;
    if (out_counters) *out_counters = fw.ncnt;
    const int np = (int)pl->probs.size(), TI = FWD_TILE_INT4 * 4;
    std::vector<int> gcols(np);
    for (int i = 0; i < np; ++i) gcols[i] = (pl->m[pl->probs[i].model]->layers[pl->probs[i].layer].N + COL_GRANULE - 1) / COL_GRANULE;
    std::vector<unsigned> cnt(fw.ncnt, 0);
    // column range a problem reads / writes in a workspace buffer
    struct Acc { int buf, c0, c1; };
    auto reads = [&](const ProbSpec &q) {
        std::vector<Acc> v;
        for (int sgi = 0; sgi < q.nseg; ++sgi)
            if (pl->buffers[q.seg[sgi].buf].external == 0) v.push_back({q.seg[sgi].buf, q.seg[sgi].col, q.seg[sgi].col + q.seg[sgi].width});
        if (q.res_buf >= 0) v.push_back({q.res_buf, q.res_col, q.res_col + pl->m[q.model]->layers[q.layer2 >= 0 && q.layer3 < 0 ? q.layer2 : q.layer].N});
        return v;
    };
    auto writes = [&](const ProbSpec &q) {
        const int N = pl->m[q.model]->layers[q.layer3 >= 0 ? q.layer3 : q.layer2 >= 0 ? q.layer2 : q.layer].N;
        return Acc{q.c_buf, q.c_col, q.c_col + N};
    };
    auto overlap = [](const Acc &x, const Acc &y) { return x.buf == y.buf && x.c0 < y.c1 && y.c0 < x.c1; };
    auto complete = [&](int prob, int w0, int w1) {           // every unit of `prob` that holds rows of windows [w0, w1)
        const ProbSpec &q = pl->probs[prob];
        const int M = (int)(batch * q.rows_per_window);
        const int a0 = w0 * q.rows_per_window, a1 = std::min(w1 * q.rows_per_window, M);
        for (int u = a0 / 32; u < (a1 + 31) / 32; ++u)
            if (cnt[fw.cnt_base[prob] + u] != (unsigned)gcols[prob]) return false;
        return true;
    };
    if (pl->kind == PLAN_SMALL) {
        // Calls of a few windows may take data as its own ready flag (poll mode): no element of a workspace buffer may then
        // be written twice in a call (a stale value would pass for data), and everything a problem reads from the workspace
        // must be written by some problem (a sentinel nobody replaces would be waited for until the spins give up).
        for (int i = 0; i < np; ++i)
            for (int o = 0; o < i; ++o)
                if (overlap(writes(pl->probs[i]), writes(pl->probs[o]))) return -25;
        // (column-exact where reader and writer see the buffer with the same row geometry - the MLPs' concatenations; a
        // pyramid level reads three of its producer's rows as one, there only "somebody writes this buffer" is checked)
        for (int i = 0; i < np; ++i)
            for (const Acc &rd : reads(pl->probs[i])) {
                std::vector<char> covered(rd.c1 - rd.c0, 0);
                bool any = false, same_rows = true;
                for (int o = 0; o < np; ++o) {
                    const Acc w = writes(pl->probs[o]);
                    if (w.buf != rd.buf) continue;
                    any = true;
                    same_rows = same_rows && pl->probs[o].rows_per_window == pl->probs[i].rows_per_window;
                    for (int c = std::max(w.c0, rd.c0); c < std::min(w.c1, rd.c1); ++c) covered[c - rd.c0] = 1;
                }
                if (!any) return -26;
                if (same_rows)
                    for (char c : covered)
                        if (!c) return -26;
            }
    }
    std::vector<int> next(fw.grid);
    for (int w = 0; w < fw.grid; ++w) next[w] = fo[w];
    int done = 0;
    for (bool progress = true; progress;) {
        progress = false;
        for (int w = 0; w < fw.grid; ++w) {
            while (next[w] < fo[w + 1]) {
                const int *d = &ft[(size_t)next[w] * TI];
                bool ready = true;
                for (int k = 0; k < d[4] && ready; ++k) {
                    const int base = d[8 + 2 * k], n = d[9 + 2 * k] & 0xffff;
                    const unsigned need = (unsigned)d[9 + 2 * k] >> 16;
                    for (int u = 0; u < n && ready; ++u) ready = cnt[base + u] >= need;
                }
                if (!ready) break;
                const int id = d[0] & 0xff, mi = d[0] >> 8;
                const ProbSpec &q = pl->probs[id];
                const int M = (int)(batch * q.rows_per_window);
                const int r1 = std::min(d[1] + mi * 32, M);
                const int w0 = d[1] / q.rows_per_window, w1 = (r1 - 1) / q.rows_per_window + 1;
                const Acc wr = writes(q);
                for (int o = 0; o < id; ++o) {                 // (problems are created in the reference's execution order)
                    const ProbSpec &oq = pl->probs[o];
                    bool hazard = false;
                    for (const Acc &rd : reads(q)) hazard = hazard || overlap(rd, writes(oq));        // read after write
                    for (const Acc &rd : reads(oq)) hazard = hazard || overlap(rd, wr);               // write after read
                    hazard = hazard || overlap(writes(oq), wr);                                      // write after write
                    if (hazard && !complete(o, w0, w1)) return -20;
                }
                if (d[5] != fw.cnt_base[id] + d[1] / 32 || d[5] + mi > fw.ncnt) return -21;
                for (int u = 0; u < mi; ++u) {
                    cnt[d[5] + u] += (unsigned)d[6];
                    if (cnt[d[5] + u] > (unsigned)gcols[id]) return -22;
                }
                ++next[w];
                ++done;
                progress = true;
            }
        }
    }
    if (done != fw.ntiles) return -23;                          // a waiting cycle
    for (int i = 0; i < np; ++i)
        for (int u = 0; u < (int)((batch * pl->probs[i].rows_per_window + 31) / 32); ++u)
            if (cnt[fw.cnt_base[i] + u] != (unsigned)gcols[i]) return -24;
    return 0;
}

#endif  // R3D_TEST_HOOKS

int r3d_clip_metrics(const float *pred_dev, const float *gt_dev, int64_t n_frames, int32_t num_joints,
                     const double *rn2w, const double *tn2w, double *out_dev, void *stream) {
    if (!pred_dev || !gt_dev || !rn2w || !tn2w || !out_dev) { r3d::set_error("r3d_clip_metrics: null pointer"); return R3D_ERR_ARG; }
    if (n_frames < 1) { r3d::set_error("r3d_clip_metrics: n_frames must be >= 1 (got %lld)", (long long)n_frames); return R3D_ERR_ARG; }
    if (num_joints < 1 || num_joints > 17) { r3d::set_error("r3d_clip_metrics: num_joints must be in 1..17 (got %d)", num_joints); return R3D_ERR_ARG; }
    if (r3d::launch_clip_metrics(pred_dev, gt_dev, n_frames, num_joints, rn2w, tn2w, out_dev, (hipStream_t)stream)) {
        r3d::set_error("r3d_clip_metrics: launch failed: %s", hipGetErrorString(hipGetLastError()));
        return R3D_ERR_HIP;
    }
    return 0;
}

int r3d_set_option(r3d_model *m, int32_t option, int64_t value) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm) { r3d::set_error("r3d_set_option: null model"); return R3D_ERR_ARG; }
    switch (option) {
        case R3D_OPT_STAGED: mm->opt_staged = value != 0; return R3D_OK;
        case R3D_OPT_SPIN_TIMEOUT_MS:
            if (value < 1 || value > 600000) { r3d::set_error("r3d_set_option: spin timeout must be 1 .. 600000 ms (got %lld)", (long long)value); return R3D_ERR_ARG; }
            mm->spin_timeout_ms = (int)value;
            return R3D_OK;
        case R3D_OPT_CU_LIMIT:
            if (value < 0 || value > 4096) { r3d::set_error("r3d_set_option: CU limit must be 0 .. 4096 (got %lld)", (long long)value); return R3D_ERR_ARG; }
            if (mm->cu_limit != (int)value) {
                // The cached schedules were packed for another workgroup count: they are dropped - which must not happen under a
                // captured graph (r3d_prepare pins the schedules a graph's kernels point into) or while a capture is being recorded.
                if (r3d::plans_pinned(mm)) {
                    r3d::set_error("r3d_set_option(R3D_OPT_CU_LIMIT): the handle has prepared (pinned) schedules - r3d_release them first; "
                                   "a captured hipGraph would be left with dangling tile lists");
                    return R3D_ERR_STATE;
                }
                // ... and their launches may still be in flight on the handle's device
                int cur = 0;
                const bool have_dev = mm->device >= 0 && hipGetDevice(&cur) == hipSuccess;
                if (have_dev && cur != mm->device) (void)hipSetDevice(mm->device);
                if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
                if (have_dev && cur != mm->device) (void)hipSetDevice(cur);
                r3d::plans_drop(mm);
                mm->cu_limit = (int)value;
                mm->last_fwd_stream = nullptr;
            }
            return R3D_OK;
        case R3D_OPT_LANES: {
            if (value != 0 && value != 1 && value != 2 && value != 4) { r3d::set_error("r3d_set_option: R3D_OPT_LANES is 0, 1, 2 or 4 (got %lld)", (long long)value); return R3D_ERR_ARG; }
            const int n = value <= 1 ? 0 : (int)value;
            if (n == mm->lanes) return R3D_OK;
            if (!mm->finalized) { r3d::set_error("r3d_set_option(R3D_OPT_LANES): r3d_finalize the handle first (the lanes live on its device)"); return R3D_ERR_STATE; }
            if (r3d::plans_pinned(mm)) {
                r3d::set_error("r3d_set_option(R3D_OPT_LANES): the handle has prepared (pinned) schedules - r3d_release them first");
                return R3D_ERR_STATE;
            }
            int cur = 0;
            const bool have_dev = mm->device >= 0 && hipGetDevice(&cur) == hipSuccess;
            if (have_dev && cur != mm->device) (void)hipSetDevice(mm->device);
            if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
            r3d::plans_drop(mm);
            r3d::lanes_destroy(mm);
            mm->lanes = 0;
            int rc = R3D_OK;
            if (n > 1 && (rc = r3d::lanes_create(mm, n)) == R3D_OK) mm->lanes = n;
            if (rc != R3D_OK) r3d::lanes_destroy(mm);
            if (have_dev && cur != mm->device) (void)hipSetDevice(cur);
            return rc;
        }
        default: r3d::set_error("r3d_set_option: unknown option %d", option); return R3D_ERR_ARG;
    }
}

int r3d_lane_stream(r3d_model *m, int32_t lane, void **stream) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm || !stream) { r3d::set_error("r3d_lane_stream: null argument"); return R3D_ERR_ARG; }
    if (lane < 0 || lane >= mm->lanes) { r3d::set_error("r3d_lane_stream: lane %d of %d", lane, mm->lanes); return R3D_ERR_ARG; }
    *stream = mm->lane[lane].stream;
    return R3D_OK;
}

int r3d_lanes_join(r3d_model *m, void *stream) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm) { r3d::set_error("r3d_lanes_join: null model"); return R3D_ERR_ARG; }
    for (int k = 0; k < mm->lanes; ++k) {
        r3d::Model::Lane &ln = mm->lane[k];
        if (ln.waiters.empty()) continue;
        // (every lane somebody still has to join: `stream` waits for more than its own forwards at most - never for less)
        hipError_t e = hipStreamWaitEvent((hipStream_t)stream, ln.done, 0);
        if (e != hipSuccess) return r3d::hip_fail(e, "hipStreamWaitEvent(lane)");
        ln.waiters.erase(std::remove(ln.waiters.begin(), ln.waiters.end(), stream), ln.waiters.end());
    }
    return R3D_OK;
}

int r3d_last_clock(r3d_model *m, void *stream, double *ghz) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm || !ghz) { r3d::set_error("r3d_last_clock: null argument"); return R3D_ERR_ARG; }
    *ghz = 0.0;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return r3d::hip_fail(e, "hipStreamSynchronize");
    if (!mm->last_clk_dev) return R3D_OK;
    unsigned w[2] = {0u, 0u};
    if ((e = hipMemcpy(w, mm->last_clk_dev, sizeof w, hipMemcpyDeviceToHost)) != hipSuccess) return r3d::hip_fail(e, "hipMemcpy(clock stamp)");
    if (w[1] > 0u) *ghz = (double)w[0] / ((double)w[1] * 10.0);      // cycles per 10 ns tick
    return R3D_OK;
}

int r3d_status(r3d_model *m, void *stream) {
    Model *mm = reinterpret_cast<Model *>(m);
    if (!mm) { r3d::set_error("r3d_status: null model"); return R3D_ERR_ARG; }
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return r3d::hip_fail(e, "hipStreamSynchronize");
    for (int k = 0; k < mm->lanes; ++k)            // (R3D_OPT_LANES: the forwards ran on the lanes' streams)
        if ((e = hipStreamSynchronize(mm->lane[k].stream)) != hipSuccess) return r3d::hip_fail(e, "hipStreamSynchronize(lane)");
    if (mm->status_host && *reinterpret_cast<volatile unsigned *>(mm->status_host) != 0u) {
        *reinterpret_cast<volatile unsigned *>(mm->status_host) = 0u;
        r3d::set_error("a dependency wait of the single-launch forward gave up after %d ms (workgroups not co-resident: the GPU is shared "
                       "with another persistent kernel, or CUs are masked): the outputs of that forward are NaN.  Run the handle "
                       "level by level: r3d_set_option(m, R3D_OPT_STAGED, 1)", mm->spin_timeout_ms);
        return R3D_ERR_ABORTED;
    }
    return R3D_OK;
}

const char *r3d_last_error(void) { return r3d::last_error(); }
const char *r3d_version(void) { return "ray3d_hip 0.6 (gfx950, ABI 6)"; }
int r3d_abi_version(void) { return R3D_ABI_VERSION; }
int r3d_precision(const r3d_model *m) {
    if (!m) { r3d::set_error("r3d_precision: null model"); return R3D_ERR_ARG; }
    return reinterpret_cast<const Model *>(m)->use_b3 ? 1 : 0;
}

}  // extern "C"
