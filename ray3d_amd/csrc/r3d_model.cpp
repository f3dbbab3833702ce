// Weight ABI of the lifting networks: the reference state_dict grammar, strict loading, and the
// BatchNorm-folding repack into the GEMM layout the gfx950 kernels read.
//
// Reference constructors this mirrors: lib/model/rie.py:13-63 (TemporalBlock), :108-120 (Linear),
// :138-157 (FCBlock), :178-253 (RIEModel), :443-494 (RIETrajectoryModel),
// lib/model/embedding.py:4-13 (Embedding).  Eval-mode BatchNorm1d: y = (x-mean)/sqrt(var+1e-5)*g+b.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "r3d_internal.hpp"

namespace r3d {

static thread_local char g_error[1024];

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_error; }

int hip_fail(hipError_t e, const char *what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return R3D_ERR_HIP;
}

static const char *kBranchNames[5] = {"Torso", "LArm", "RArm", "LLeg", "RLeg"};

// joints per body part (lib/model/rie.py:308-331 for F=3, :334-357 for F=2: same joints)
static std::vector<int> group_joints(int J, int g) {
    static const std::vector<int> g17[5] = {{0, 7, 8, 9, 10}, {14, 15, 16}, {11, 12, 13}, {1, 2, 3}, {4, 5, 6}};
    static const std::vector<int> g15[5] = {{0, 1, 14}, {2, 3, 4}, {5, 6, 7}, {8, 9, 10}, {11, 12, 13}};
    static const std::vector<int> g14[5] = {{0, 7}, {8, 9, 10}, {11, 12, 13}, {4, 5, 6}, {1, 2, 3}};
    return J == 17 ? g17[g] : J == 15 ? g15[g] : g14[g];
}

namespace {

struct Grammar {
    Model *m;
    void tensor(const std::string &key, std::initializer_list<int64_t> shape) {
        TensorSpec t;
        t.key = key;
        t.rank = (int)shape.size();
        int i = 0;
        for (auto d : shape) t.shape[i++] = d;
        for (; i < 4; ++i) t.shape[i] = 1;
        m->spec_index[key] = (int)m->specs.size();
        m->specs.push_back(t);
    }
    void bn(const std::string &p, int c) {
        tensor(p + ".weight", {c});
        tensor(p + ".bias", {c});
        tensor(p + ".running_mean", {c});
        tensor(p + ".running_var", {c});
    }
    int layer(const std::string &prefix, int taps, int cin, int n, bool bias, const std::string &bn_prefix,
              float slope, bool conv, const std::vector<Layer::Pre> &pre = {}) {
        Layer L;
        L.pre = pre;
        L.cin_ref = cin;
        for (const auto &r : pre) cin += r.new_width - r.ref_width;
        L.weight_key = prefix + ".weight";
        L.bias_key = bias ? prefix + ".bias" : "";
        L.bn_prefix = bn_prefix;
        L.taps = taps;
        L.cin = cin;
        L.N = n;
        L.K = taps * cin;
        L.Npad = round_up(n, N_ALIGN);
        L.Kpad = round_up(L.K, BK);
        L.slope = slope;
        L.frag = true;
        L.w_off = L.b_off = 0;
        if (conv)
            tensor(L.weight_key, {n, cin, taps});
        else
            tensor(L.weight_key, {n, L.cin_ref});
        if (bias) tensor(L.bias_key, {n});
        if (!bn_prefix.empty()) bn(bn_prefix, n);
        // the FCBlocks' wide Linears can also run on the bf16 matrix cores (three-term split, r3d_kernels.hip)
        L.bf3 = m->use_b3 && !conv && n == MLP_HIDDEN && cin % BK == 0 && cin >= 256;
        L.bf3_conv = false;
        m->layer_index[prefix] = (int)m->layers.size();
        m->layers.push_back(L);
        return (int)m->layers.size() - 1;
    }
    // TemporalBlock, lib/model/rie.py:13-63
    void temporal_block(const std::string &p, int cin) {
        const int C = m->cfg.channels;
        const int le = layer(p + ".expand_conv", 3, cin, C, false, p + ".expand_bn", 0.2f, true);
        // fp32 on the bf16 matrix cores for the fused first level (first_level_taps_b3): its three layers also get a
        // copy in bf16-MFMA operand order
        const bool fl_b3 = m->use_b3 && m->cfg.num_levels >= 2 && C <= N_ALIGN && !m->cfg.dense;
        m->layers[le].bf3_conv = fl_b3;
        int dil = 3;
        for (int i = 1; i < m->cfg.num_levels; ++i, dil *= 3) {
            const std::string a = std::to_string(2 * (i - 1)), b = std::to_string(2 * (i - 1) + 1);
            // 3 taps - or, for the dense ablation (rie.py:49-53), 2 * pad + 1 with pad = the level's dilation 3^i
            const int la = layer(p + ".layers_conv." + a, m->cfg.dense ? 2 * dil + 1 : 3, C, C, false, p + ".layers_bn." + a, 0.2f, true);
            const int lb = layer(p + ".layers_conv." + b, 1, C, C, false, p + ".layers_bn." + b, 0.2f, true);
            m->layers[la].bf3_conv = m->layers[lb].bf3_conv = fl_b3;   // (level 1: in the fused first level; further levels: fused pairs)
            // the register-chained first-level tile (r3d_chain.hpp): 256 channels, an operand of 64 gathered columns, strided 3-tap level
            // (whether the gathered operand is 64 columns wide is known once the first layers are folded: model_finalize)
            if (i == 1 && C == 256 && !m->cfg.dense) { m->layers[le].chain_l1 = la; m->layers[le].chain_l2 = lb; }
        }
        layer(p + ".shrink", 1, C, m->cfg.latent, true, "", 1.0f, true);
    }
    // FCBlock, lib/model/rie.py:138-157
    void fc_block(const std::string &p, int cin, int cout, int nblocks, const std::vector<Layer::Pre> &pre = {}) {
        layer(p + ".fc_1", 1, cin, MLP_HIDDEN, true, p + ".bn_1", 0.2f, false, pre);
        for (int n = 0; n < nblocks; ++n) {
            const std::string q = p + ".layers." + std::to_string(n);
            layer(q + ".w1", 1, MLP_HIDDEN, MLP_HIDDEN, true, q + ".batch_norm1", 0.2f, false);
            layer(q + ".w2", 1, MLP_HIDDEN, MLP_HIDDEN, true, q + ".batch_norm2", 0.2f, false);
        }
        layer(p + ".fc_2", 1, MLP_HIDDEN, cout, true, "", 1.0f, false);
    }
    // Embedding, lib/model/embedding.py:4-18: Linear+BN+LeakyReLU(0.01) twice.  Two (tiny) GEMM layers that ride
    // along the big launches of the conv pyramid - a kernel of their own cost more than their arithmetic.
    void embedding(const std::string &p, int cin, int cout) {
        layer(p + ".w1", 1, cin, EMBED_MID, true, p + ".b1", 0.01f, false);
        layer(p + ".w2", 1, EMBED_MID, cout, true, p + ".b2", 0.01f, false);
    }
};

}  // namespace

Model::~Model() {
    plans_drop(this);
    if (d_arena) (void)hipFree(d_arena);
    if (d_iarena) (void)hipFree(d_iarena);
    if (status_host) (void)hipHostFree(status_host);
    if (order_ev) (void)hipEventDestroy(order_ev);
    lanes_destroy(this);         // (also takes the lanes' streams off the forward-ordering lists: r3d_api.cpp)
    for (auto &r : recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
}

Model *model_create(const r3d_config &cfg) {
    const int J = cfg.num_joints, F = cfg.in_features;
    if (cfg.kind != R3D_KIND_POS && cfg.kind != R3D_KIND_TRJ) { set_error("kind must be R3D_KIND_POS or R3D_KIND_TRJ"); return nullptr; }
    if (J != 14 && J != 15 && J != 17) { set_error("num_joints must be 14, 15 or 17 (got %d)", J); return nullptr; }
    if (F != 2 && F != 3) { set_error("in_features must be 2 or 3 (got %d)", F); return nullptr; }
    if (cfg.num_levels < 1 || cfg.num_levels > 6) { set_error("num_levels must be in 1..6 (got %d)", cfg.num_levels); return nullptr; }
    if (cfg.channels <= 0 || cfg.channels % BK || cfg.latent <= 0 || cfg.latent % BK) {
        set_error("channels and latent must be positive multiples of %d", BK);
        return nullptr;
    }
    const bool emb = cfg.extrinsic_dim > 0 && cfg.embed_dim > 0;
    if (emb && (cfg.embed_dim % BK || cfg.embed_dim > 128 || cfg.extrinsic_dim > 8)) {
        set_error("embed_dim must be a multiple of %d (<= 128) and extrinsic_dim <= 8", BK);
        return nullptr;
    }
    if (cfg.kind == R3D_KIND_POS && cfg.stage < 1) { set_error("stage must be >= 1"); return nullptr; }
    if (cfg.causal != 0 && cfg.causal != 1) { set_error("causal must be 0 or 1 (got %d)", cfg.causal); return nullptr; }
    if (cfg.dense != 0 && cfg.dense != 1) { set_error("dense must be 0 or 1 (got %d)", cfg.dense); return nullptr; }
    if (cfg.bf16x3 != 0 && cfg.bf16x3 != 1) { set_error("bf16x3 must be 0 or 1 (got %d)", cfg.bf16x3); return nullptr; }
    if (cfg.dense && cfg.num_levels > 4) { set_error("dense convolutions are evaluated at every position of a window: num_levels <= 4 (RF <= 81)"); return nullptr; }

    Model *m = new Model();
    static std::atomic<uint64_t> next_id{1};
    m->id = next_id.fetch_add(1);
    m->cfg = cfg;
    {
        const char *e = getenv("R3D_BF16X3");          // overrides the configuration's field when set
        m->use_b3 = e ? atoi(e) != 0 : cfg.bf16x3 != 0;
    }
    // shrink folds into the Linears that read it when that does not widen them (Layer::pre)
    m->fold_shrink = cfg.channels <= cfg.latent && !hook_on("R3D_NO_SHRINK_FOLD");
    if (!emb) m->cfg.extrinsic_dim = m->cfg.embed_dim = 0;
    m->RF = 1;
    for (int i = 0; i < cfg.num_levels; ++i) m->RF *= 3;
    Grammar g{m};
    const int lat = cfg.latent, D = m->cfg.embed_dim;
    if (cfg.kind == R3D_KIND_POS) {
        for (int b = 0; b < 5; ++b) {
            Model::Branch br;
            br.prefix = std::string("LocalLayer_") + kBranchNames[b];
            br.joints = group_joints(J, b);
            br.cin = 3 * (int)br.joints.size() * F;
            br.k0 = 3 * br.cin;
            br.k0pad = round_up(br.k0, BK);
            br.lut_off = br.lut_uv_off = 0;
            m->branches.push_back(br);
            g.temporal_block(br.prefix, br.cin);
        }
        g.fc_block("GlobalInfo", J * F, lat, 2);
        auto shrink_of = [&](int b) { return m->layer_index.at(m->branches[b].prefix + ".shrink"); };
        const int C = cfg.channels;
        if (cfg.stage != 1)
            for (int i = 0; i < 5; ++i) {
                // input: the other four branches' local features in branch order (rie.py:393-394)
                std::vector<Layer::Pre> pre;
                if (m->fold_shrink)
                    for (int sl = 0; sl < 4; ++sl) pre.push_back({sl * lat, lat, C, shrink_of(sl < i ? sl : sl + 1)});
                g.fc_block("FuseBlocks." + std::to_string(i), 4 * lat, lat, 1, pre);
            }
        if (emb) g.embedding("embedder", cfg.extrinsic_dim, D);
        const int dec_in = (cfg.stage == 1 ? 2 : 3) * lat + D;
        for (int b = 0; b < 5; ++b) {
            std::vector<Layer::Pre> pre;
            if (m->fold_shrink) pre.push_back({0, lat, C, shrink_of(b)});
            g.fc_block(std::string("Integration_") + kBranchNames[b], dec_in, 3 * (int)m->branches[b].joints.size(), 1, pre);
        }
    } else {
        Model::Branch br;
        br.prefix = "LocalLayer";
        for (int j = 0; j < J; ++j) br.joints.push_back(j);
        br.cin = 3 * J * F;
        br.k0 = 3 * br.cin;
        br.k0pad = round_up(br.k0, BK);
        br.lut_off = br.lut_uv_off = 0;
        m->branches.push_back(br);
        g.temporal_block(br.prefix, br.cin);
        g.fc_block("GlobalInfo", J * F, lat, 2);
        if (emb) g.embedding("embedder", cfg.extrinsic_dim, D);
        std::vector<Layer::Pre> pre;
        if (m->fold_shrink) pre.push_back({0, lat, cfg.channels, m->layer_index.at(br.prefix + ".shrink")});
        g.fc_block("Integration", 2 * lat + D, 3, 1, pre);
    }
    // GlobalInfo.fc_1 reads the zero-padded current-frame matrix
    m->layers[m->layer_index["GlobalInfo.fc_1"]].Kpad = CUR_LD;
    // ---- fused-prologue tables (layout: r3d_internal.hpp) and the matching column maps of the first layers.
    // Reference channel order inside a branch: cat(x_g, diff_g, diff_t_g) per tap (rie.py:308-315, :540).  The
    // layer is linear in its input, so  W1 x + W2 (x - root) + W3 (x - x_cur)  is evaluated as
    // (W1 + W2 + W3) x - W2 root - W3 x_cur:  a third of the operand columns and no subtraction in the kernel.
    m->iarena.clear();
    const int JF = J * F;
    for (auto &br : m->branches) {
        const int n = (int)br.joints.size(), nF = n * F;
        int root_pos = -1;                                   // position of joint 0 (the root, rie.py:301) in the group
        for (int q = 0; q < n; ++q)
            if (br.joints[q] == 0) root_pos = q;
        const int GX = round_up(3 * nF, 4);                  // x columns (tap-major), padded to 4
        const int GR = root_pos >= 0 ? 0 : round_up(3 * F, 4);   // root-joint columns (only when the group lacks joint 0)
        const int GC = round_up(nF, 4);                      // current-frame columns
        br.k0 = GX + GR + GC;
        br.k0pad = round_up(br.k0, BK);
        Layer &L = m->layers[m->layer_index[br.prefix + ".expand_conv"]];
        L.Kpad = br.k0pad;
        L.colmap.assign(3 * br.cin, -1);
        L.colmap_neg.assign(3 * br.cin, -1);
        std::vector<int> l1(br.k0pad, ENC_INVALID), l1uv(br.k0pad, ENC_INVALID), lk(br.k0pad / 4, 0);
        // UV mode: element (frame, joint, f) comes from pixel coordinate (f > 0) of that joint, f kept in the low bits
        auto uv_off = [&](int frame, int joint, int f) { return ((frame * J + joint) * 2 + (f > 0 ? 1 : 0)) * 4 + f; };
        for (int tap = 0; tap < 3; ++tap)
            for (int c = 0; c < nF; ++c) {
                const int src = br.joints[c / F] * F + c % F;           // element of the (J,F) frame
                const int xcol = tap * nF + c;
                l1[xcol] = (tap * JF + src) * 4;
                l1uv[xcol] = uv_off(tap, br.joints[c / F], c % F);
                // the root term's column: the group's own joint-0 column, or a dedicated one
                const int rcol = root_pos >= 0 ? tap * nF + root_pos * F + c % F : GX + tap * F + c % F;
                if (root_pos < 0) {
                    l1[rcol] = (tap * JF + c % F) * 4;
                    l1uv[rcol] = uv_off(tap, 0, c % F);
                }
                const int ccol = GX + GR + c;
                l1[ccol] = src * 4;
                l1uv[ccol] = uv_off(0, br.joints[c / F], c % F);
                lk[ccol / 4] = 1;                                       // relative to the window's current frame
                for (int kind = 0; kind < 3; ++kind) {
                    const int t = tap * br.cin + kind * nF + c;         // torch column (tap, channel kind*nF + c)
                    L.colmap[t] = xcol;
                    L.colmap_neg[t] = kind == 1 ? rcol : kind == 2 ? ccol : -1;
                }
            }
        br.lut_off = m->iarena.size();
        m->iarena.insert(m->iarena.end(), l1.begin(), l1.end());
        m->iarena.insert(m->iarena.end(), lk.begin(), lk.end());
        br.lut_uv_off = m->iarena.size();
        m->iarena.insert(m->iarena.end(), l1uv.begin(), l1uv.end());
        m->iarena.insert(m->iarena.end(), lk.begin(), lk.end());
        // the per-frame form (Layer::shared_of): the same columns, the current-frame ones gathered at the row's own frame
        if (m->cfg.num_levels >= 2 && m->cfg.channels <= N_ALIGN && !m->cfg.dense && br.k0pad <= 256) {
            std::vector<int> lk0(br.k0pad / 4, 0);
            br.lut_frame_off = m->iarena.size();
            m->iarena.insert(m->iarena.end(), l1.begin(), l1.end());
            m->iarena.insert(m->iarena.end(), lk0.begin(), lk0.end());
            br.lut_frame_uv_off = m->iarena.size();
            m->iarena.insert(m->iarena.end(), l1uv.begin(), l1uv.end());
            m->iarena.insert(m->iarena.end(), lk0.begin(), lk0.end());
            Layer S;
            S.weight_key = S.bias_key = S.bn_prefix = "";
            S.taps = 1;
            S.cin = S.cin_ref = br.k0pad;
            S.N = 2 * m->cfg.channels;
            S.K = S.Kpad = br.k0pad;
            S.Npad = round_up(S.N, N_ALIGN);
            S.slope = 1.0f;
            S.frag = true;
            S.w_off = S.b_off = 0;
            S.shared_of = m->layer_index[br.prefix + ".expand_conv"];
            S.shared_split = GX + GR;
            br.frame_layer = (int)m->layers.size();
            m->layer_index[br.prefix + ".expand_conv@frame"] = br.frame_layer;
            m->layers.push_back(S);
        }
    }
    // GlobalInfo.fc_1 reads in_current = x[:, RF // F] flattened (rie.py:290-292), zero padded to CUR_LD
    {
        std::vector<int> l1(CUR_LD, ENC_INVALID), l1uv(CUR_LD, ENC_INVALID), lk(CUR_LD / 4, 1);
        for (int col = 0; col < JF; ++col) {
            l1[col] = col * 4;
            l1uv[col] = ((col / F) * 2 + (col % F > 0 ? 1 : 0)) * 4 + col % F;
        }
        m->global_lut_off = m->iarena.size();
        m->iarena.insert(m->iarena.end(), l1.begin(), l1.end());
        m->iarena.insert(m->iarena.end(), lk.begin(), lk.end());
        m->global_lut_uv_off = m->iarena.size();
        m->iarena.insert(m->iarena.end(), l1uv.begin(), l1uv.end());
        m->iarena.insert(m->iarena.end(), lk.begin(), lk.end());
    }
    for (auto &kv : m->layer_index)
        if (kv.first.rfind("Integration", 0) == 0 && kv.first.size() > 5 && kv.first.compare(kv.first.size() - 5, 5, ".fc_2") == 0)
            m->layers[kv.second].frag = false;   // consumed by r3d_decode_f32
    m->host_weights.resize(m->specs.size());
    m->have.assign(m->specs.size(), false);
    return m;
}

int model_set_weight(Model *m, const char *key_in, const float *host, const int64_t *shape, int rank) {
    if (!m || !key_in || !host || !shape) { set_error("r3d_set_weight: null argument"); return R3D_ERR_ARG; }
    std::string key(key_in);
    if (key.rfind("module.", 0) == 0) key = key.substr(7);   // nn.DataParallel checkpoints
    auto it = m->spec_index.find(key);
    if (it == m->spec_index.end()) {
        set_error("unexpected key '%s' for this configuration", key.c_str());
        return R3D_ERR_KEY;
    }
    const TensorSpec &t = m->specs[it->second];
    bool ok = (rank == t.rank);
    for (int i = 0; ok && i < rank; ++i) ok = (shape[i] == t.shape[i]);
    if (!ok) {
        std::string got = "[", want = "[";
        for (int i = 0; i < rank && i < 4; ++i) got += std::to_string((long long)shape[i]) + (i + 1 < rank ? "," : "");
        for (int i = 0; i < t.rank; ++i) want += std::to_string((long long)t.shape[i]) + (i + 1 < t.rank ? "," : "");
        set_error("size mismatch for '%s': got %s], expected %s]", key.c_str(), got.c_str(), want.c_str());
        return R3D_ERR_SHAPE;
    }
    auto &dst = m->host_weights[it->second];
    dst.assign(host, host + t.numel());
    m->have[it->second] = true;
    m->dirty = true;
    return R3D_OK;
}

namespace {

struct Folder {
    Model *m;
    const float *get(const std::string &key) const { return m->host_weights[m->spec_index.at(key)].data(); }
    // per-output-channel scale/shift of an eval BatchNorm following a layer with optional bias
    void scale_shift(const Layer &L, std::vector<double> &s, std::vector<double> &t) const {
        s.assign(L.N, 1.0);
        t.assign(L.N, 0.0);
        const float *bias = L.bias_key.empty() ? nullptr : get(L.bias_key);
        if (L.bn_prefix.empty()) {
            for (int o = 0; o < L.N; ++o) t[o] = bias ? (double)bias[o] : 0.0;
            return;
        }
        const float *g = get(L.bn_prefix + ".weight"), *b = get(L.bn_prefix + ".bias");
        const float *mu = get(L.bn_prefix + ".running_mean"), *var = get(L.bn_prefix + ".running_var");
        for (int o = 0; o < L.N; ++o) {
            const double sc = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
            s[o] = sc;
            t[o] = (double)b[o] - (double)mu[o] * sc + (bias ? (double)bias[o] * sc : 0.0);
        }
    }
};

}  // namespace

int model_finalize(Model *m) {
    if (!m) { set_error("r3d_finalize: null model"); return R3D_ERR_ARG; }
    for (size_t i = 0; i < m->specs.size(); ++i)
        if (!m->have[i]) {
            int missing = 0;
            for (size_t j = 0; j < m->specs.size(); ++j) missing += !m->have[j];
            set_error("Missing key(s) in state_dict: '%s' (and %d more)", m->specs[i].key.c_str(), missing - 1);
            return R3D_ERR_KEY;
        }
    // ---- lay out the arenas
    size_t off = 0;
    for (auto &L : m->layers) {
        L.w_off = off;
        off += (size_t)L.Npad * L.Kpad;
        L.b_off = off;
        off += (size_t)L.Npad;
        if (L.bf3 || L.bf3_conv) {
            off = (off + 3) / 4 * 4;                       // 16-byte aligned planes
            L.wb3_off = off;
            off += (size_t)L.Npad * L.Kpad;
        }
        L.chain_off = 0;
        if (L.chain_l1 >= 0 && L.Kpad == 64 && !L.colmap.empty() && hook_on("R3D_CHAIN")) {   // (experiment: hooks build with R3D_CHAIN=1 only)
            off = (off + 255) / 256 * 256;                 // 1 KiB aligned slabs
            L.chain_off = off;
            off += (size_t)(L.Kpad / 16 + 64) * 4096;
        }
    }
    m->arena.assign(off, 0.0f);
    Folder f{m};
    std::vector<double> s, t;
    for (auto &L : m->layers) {
        if (L.shared_of >= 0) {
            // [E | V] from the folded, column-mapped source layer (processed earlier: it precedes this one in the list)
            const Layer &S = m->layers[L.shared_of];
            const float *src = m->arena.data() + S.w_off, *sb = m->arena.data() + S.b_off;
            float *dst = m->arena.data() + L.w_off, *bd = m->arena.data() + L.b_off;
            const int nk = L.Kpad / BK, C = S.N;
            for (int o = 0; o < C; ++o)
                for (int k = 0; k < L.Kpad; ++k) {
                    const float v = src[frag_index(o, k, nk)];
                    dst[frag_index(o, k, nk)] = k < L.shared_split ? v : 0.0f;
                    dst[frag_index(C + o, k, nk)] = k < L.shared_split ? 0.0f : v;
                }
            for (int o = 0; o < C; ++o) { bd[o] = sb[o]; bd[C + o] = 0.0f; }
            continue;
        }
        f.scale_shift(L, s, t);
        const float *w = f.get(L.weight_key);
        float *dst = m->arena.data() + L.w_off;
        // torch layout (N, cin, taps) [Linear: taps == 1]; GEMM column index = tap*cin + c.
        // GEMM layers are stored in MFMA fragment order (see r3d_kernels.hip), the decoder tail row-major.
        const int nk = L.Kpad / BK;
        if (!L.pre.empty()) {
            // W_eff = [.. | W[:, range] S | ..],  t += s * (W[:, range] s_bias): composed in double, one output row at a time
            std::vector<double> rowacc(L.Kpad);
            for (int o = 0; o < L.N; ++o) {
                std::fill(rowacc.begin(), rowacc.end(), 0.0);
                double extra = 0.0;
                int c = 0, k = 0;
                size_t ri = 0;
                while (c < L.cin_ref) {
                    if (ri < L.pre.size() && c == L.pre[ri].ref_col0) {
                        const Layer::Pre &r = L.pre[ri++];
                        const Layer &S = m->layers[r.shrink_layer];           // (N = ref_width, cin = new_width, one tap)
                        const float *sw = f.get(S.weight_key), *sb = S.bias_key.empty() ? nullptr : f.get(S.bias_key);
                        for (int q = 0; q < r.ref_width; ++q) {
                            const double wq = (double)w[(size_t)o * L.cin_ref + c + q];
                            const float *srow = sw + (size_t)q * r.new_width;
                            for (int j = 0; j < r.new_width; ++j) rowacc[k + j] += wq * (double)srow[j];
                            if (sb) extra += wq * (double)sb[q];
                        }
                        c += r.ref_width;
                        k += r.new_width;
                    } else {
                        rowacc[k++] = (double)w[(size_t)o * L.cin_ref + c++];
                    }
                }
                for (int kk = 0; kk < L.Kpad; ++kk) {
                    const float v = (float)(rowacc[kk] * s[o]);
                    dst[L.frag ? frag_index(o, kk, nk) : (size_t)o * L.Kpad + kk] = v;
                }
                t[o] += extra * s[o];
            }
        } else if (L.colmap.empty()) {
            for (int o = 0; o < L.N; ++o)
                for (int c = 0; c < L.cin; ++c)
                    for (int j = 0; j < L.taps; ++j) {
                        const int k = j * L.cin + c;
                        const float v = (float)((double)w[((size_t)o * L.cin + c) * L.taps + j] * s[o]);
                        dst[L.frag ? frag_index(o, k, nk) : (size_t)o * L.Kpad + k] = v;
                    }
        } else {
            // first layers: several reference columns add up (with sign) in one operand column; summed in double
            std::vector<double> rowacc(L.Kpad);
            for (int o = 0; o < L.N; ++o) {
                std::fill(rowacc.begin(), rowacc.end(), 0.0);
                for (int c = 0; c < L.cin; ++c)
                    for (int j = 0; j < L.taps; ++j) {
                        const double v = (double)w[((size_t)o * L.cin + c) * L.taps + j] * s[o];
                        const int t = j * L.cin + c;
                        rowacc[L.colmap[t]] += v;
                        if (L.colmap_neg[t] >= 0) rowacc[L.colmap_neg[t]] -= v;
                    }
                for (int k = 0; k < L.Kpad; ++k) dst[frag_index(o, k, nk)] = (float)rowacc[k];
            }
        }
        float *bd = m->arena.data() + L.b_off;
        for (int o = 0; o < L.N; ++o) bd[o] = (float)t[o];
        if (L.bf3 || L.bf3_conv) {
            // the same folded weights in v_mfma_f32_32x32x16_bf16 operand order (the kernel splits them into three bf16
            // terms in registers): [32-col block][K tile][k16 half][4-float group][lane][4], lane l holding
            // column l % 32 and k = 8 * (l / 32) + 4 * group + e of the half.  Read back from the fragment-ordered
            // copy, which already holds the folded (and, for first layers, column-mapped) values.
            float *pk = m->arena.data() + L.wb3_off;
            for (int o = 0; o < L.N; ++o)
                for (int k = 0; k < L.Kpad; ++k) {
                    const float v = dst[frag_index(o, k, nk)];
                    const int nb = o >> 5, kt = k >> 5, kin = k & 31, h = kin >> 4, kk = kin & 15;
                    const int lane = (kk >> 3) * 32 + (o & 31), j = (kk & 7) >> 2, e = kk & 3;
                    pk[((((size_t)(nb * nk + kt) * 2 + h) * 2 + j) * 64 + lane) * 4 + e] = v;
                }
        }
    }
    // ---- the fused first levels once more as slab streams of the register-chained tile (r3d_chain.hpp).  A slab is 16 fragments of
    // 1 KiB [half group hg][fragment f][lane][element e]; lane l = (channel l & 15 of the fragment's block, feature quarter
    // gq = l >> 4).  WIDE slab m of a layer: output channel 16 (4 hg + f) + (l & 15), K step 4 m + e; NARROW slab m of output group G:
    // output channel 16 (4 G + f) + (l & 15), K step 16 m + 4 hg + e.  The feature a K step takes from lane quarter gq: gathered
    // layer (expand_conv): operand column 4 step + gq (the four lanes of a row gather four consecutive columns); chained layers: channel 16 (step >> 2) + 4 gq + (step & 3) - register
    // step & 3 of channel block step >> 2 of the previous layer's accumulators.  Stream: expand_conv (K0 / 16 slabs), the three taps
    // of the 3-tap convolution in the order the tile visits them (residual tap last; 16 slabs each), the 1x1 convolution (4 groups
    // x 4 slabs).  Values are read back from the fragment-ordered copies, which hold the folded (and column-mapped) weights.
    for (auto &L : m->layers) {
        if (L.chain_off == 0) continue;
        const Layer &L1 = m->layers[L.chain_l1], &L2 = m->layers[L.chain_l2];
        const int C = L.N, K0 = L.Kpad, sl_exp = K0 / 16;
        const float *w0 = m->arena.data() + L.w_off, *w1 = m->arena.data() + L1.w_off, *w2 = m->arena.data() + L2.w_off;
        const int nk0 = L.Kpad / BK, nk1 = L1.Kpad / BK, nk2 = L2.Kpad / BK;
        float *img = m->arena.data() + L.chain_off;
        auto chained = [](int step, int gq) { return 16 * (step >> 2) + 4 * gq + (step & 3); };
        auto put = [&](int slab, int hg, int f, int l, int e, float v) { img[(size_t)slab * 4096 + ((hg * 4 + f) * 64 + l) * 4 + e] = v; };
        const int res_tap = 1 + (m->cfg.causal ? 1 : 0);
        const int order[3] = {0, 3 - res_tap, res_tap};
        for (int hg = 0; hg < 4; ++hg)
            for (int f = 0; f < 4; ++f)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const int o = 16 * (4 * hg + f) + (l & 15), gq = l >> 4;
                        for (int ms = 0; ms < sl_exp; ++ms) put(ms, hg, f, l, e, w0[frag_index(o, 4 * (4 * ms + e) + gq, nk0)]);
                        for (int ts = 0; ts < 3; ++ts)
                            for (int ms = 0; ms < 16; ++ms)
                                put(sl_exp + ts * 16 + ms, hg, f, l, e, w1[frag_index(o, order[ts] * C + chained(4 * ms + e, gq), nk1)]);
                        for (int G = 0; G < 4; ++G)
                            for (int ms = 0; ms < 4; ++ms)
                                put(sl_exp + 48 + G * 4 + ms, hg, f, l, e, w2[frag_index(16 * (4 * G + f) + (l & 15), chained(16 * ms + 4 * hg + e, gq), nk2)]);
                    }
    }
    // ---- upload
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (m->d_arena && m->device != dev) {
        (void)hipFree(m->d_arena); (void)hipFree(m->d_iarena);
        m->d_arena = nullptr; m->d_iarena = nullptr;
        plans_drop(m);
    }
    m->device = dev;
    if (!m->status_host) {             // the word the decoder kernel raises when a dependency wait gave up (r3d_status); portable + mapped:
                                       // the handle may be re-finalised on another device and the word stays valid there
        if ((e = hipHostMalloc((void **)&m->status_host, 64, hipHostMallocPortable | hipHostMallocMapped)) != hipSuccess) return hip_fail(e, "hipHostMalloc(status)");
        *m->status_host = 0u;
    }
    if (!m->d_arena) {
        if ((e = hipMalloc((void **)&m->d_arena, m->arena.size() * sizeof(float))) != hipSuccess) return hip_fail(e, "hipMalloc(weights)");
        if ((e = hipMalloc((void **)&m->d_iarena, m->iarena.size() * sizeof(int))) != hipSuccess) return hip_fail(e, "hipMalloc(luts)");
    }
    // a re-finalisation overwrites weights that forwards in flight on any stream may still be reading (a blocking
    // hipMemcpy only orders against the null stream)
    if (m->finalized && (e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e, "hipDeviceSynchronize");
    if ((e = hipMemcpy(m->d_arena, m->arena.data(), m->arena.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "hipMemcpy(weights)");
    if ((e = hipMemcpy(m->d_iarena, m->iarena.data(), m->iarena.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "hipMemcpy(luts)");
    m->finalized = true;
    m->dirty = false;
    return R3D_OK;
}

}  // namespace r3d
