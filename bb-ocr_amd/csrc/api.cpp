// C-ABI of libbbocr (include/bbocr.h): context, weight folding/packing, detector / box / recogniser pipelines.
// Host orchestration only; every arithmetic stage is a HIP kernel (conv_mfma.hip, craft_misc.hip, ccl.hip,
// crnn_misc.hip, lstm.hip, ctc.hip) or the O(#boxes) geometry in boxpost.cpp.
#include "../../include/bbocr.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "boxpost.h"
#include "common.h"
#include "kernels.h"

namespace {

using clk = std::chrono::steady_clock;
static inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct StatusError {
    int code;
    std::string msg;
};
[[noreturn]] static void fail(int code, const std::string& m) { throw StatusError{code, m}; }
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) fail(BBOCR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

struct DevBuf {   // growable device buffer, freed with its owner (the context)
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        HIPCHK(hipMalloc(&p, n));
        cap = n;
    }
    void ensure_keep(size_t n, size_t used) {   // grow, keeping the first `used` bytes (waits for the device: the old buffer may be in use)
        if (n <= cap) return;
        void* q = nullptr;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMalloc(&q, n));
        if (p && used) HIPCHK(hipMemcpy(q, p, std::min(used, cap), hipMemcpyDeviceToDevice));
        if (p) (void)hipFree(p);
        p = q;
        cap = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinBuf {   // growable pinned host buffer: the source of truly asynchronous H2D copies (a pageable source makes hipMemcpyAsync wait
                  // for the stream first, which stalls the host exactly where it should be queueing work behind a running kernel)
    void* p = nullptr;
    size_t cap = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) { HIPCHK(hipDeviceSynchronize()); (void)hipHostFree(p); }     // a queued copy may still read the old buffer
        p = nullptr;
        cap = 0;
        HIPCHK(hipHostMalloc(&p, n, hipHostMallocDefault));
        cap = n;
    }
};

struct Arena {   // bump allocator over one device buffer; a dry pass sizes it, the real pass carves it
    DevBuf buf;
    size_t off = 0;
    bool dry = true;
    void begin(bool d) { off = 0; dry = d; }
    template <typename T> T* alloc(size_t count) {
        off = align_up(off, 256);
        T* r = dry ? nullptr : (T*)((char*)buf.p + off);
        off += count * sizeof(T);
        return r;
    }
};

struct Act {   // bf16 NHWC activation
    uint16_t* p;
    int N, H, W, C;
};

}  // namespace

struct bbocr_ctx {
    bbocr_config cfg{};
    hipStream_t stream = nullptr;
    DevBuf pp_gray, pp_a, pp_b, pp_c, pp_tab;  // pre-processing chain (f2): planes and small tables
    unsigned int ignore_mask[4] = {0, 0, 0, 0};   // recogniser class mask of the running call (bbocr_params::ignore_mask)
    int beam_width = 0;                           // > 0: decoder='beamsearch' for the running call (bbocr_params::decoder / beam_width)
    hipStream_t cur = nullptr;                // stream the layer helpers launch on
    hipStream_t stream2 = nullptr;            // box extraction of detector sub-batch k runs here while sub-batch k+1 is on `stream`
    std::vector<hipEvent_t> sub_events;       // one per detector sub-batch of a readtext_batch call
    hipEvent_t ccl_t0 = nullptr, ccl_t1 = nullptr;   // GPU span of the CCL kernels of one boxes_impl call
    hipEvent_t det_t0 = nullptr, det_t1 = nullptr;   // detector span on `stream` (the host is busy with boxes meanwhile)
    std::mutex mu;
    std::string err;
    float times[8] = {0};

    // ---- optional per-launch timing of the conv_mfma kernel (HIP events on this context's stream)
    struct ProfRec { hipEvent_t e0, e1; double flops; int group; };
    int profiling = 0;                        // 0 off, 1 = time the detector's conv launches (group 0), 2 = also the recogniser's
    int prof_group = 0;                     // 0 = detector, 1 = recogniser
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[2] = {0, 0}, prof_flops[2] = {0, 0};
    long long prof_launches[2] = {0, 0};

    // ---- detector
    bool craft_loaded = false;
    uint16_t* c11_w = nullptr;
    uint16_t* c11_wf = nullptr;               // conv1_1 weights in the layout of the producer fused into conv1_2
    float* c11_b = nullptr;
    ConvPlan conv1_2, conv2_1, conv2_2, conv3_1, conv3_2, conv3_3, conv4_1, conv4_2, conv4_3, conv5_1, conv5_2, fc6, fc7;
    ConvPlan up1a, up1b, up2b, up3b, up4b, cls0, cls2, cls4;
    // U-net 1x1 layers over cat[up(y), skip], split by linearity: upNy = the columns of y (no bias, run at y's resolution),
    // upNs = the columns of the skip tensor (+ bias), whose epilogue adds the 2x bilinear up-sampling of upNy's output
    ConvPlan up2y, up2s, up3y, up3s, up4y, up4s;
    float* cls_tail = nullptr;   // b1[16] w2[32] b2[2]
    uint16_t* cls_tail_frag = nullptr;   // conv_cls.6 weight as an MFMA A fragment
    // ---- recogniser
    bool crnn_loaded = false;
    float* r0_wb = nullptr;      // w[9][32] (tap-major) b[32]
    ConvPlan r1, r2, r3, r4, r5, r6, xproj[2], lin[2], pred;
    uint16_t* whh[2] = {nullptr, nullptr};
    std::vector<void*> owned;    // every hipMalloc'd weight block

    void* zero_page = nullptr;   // 256 zero bytes (padding source of the LDS-DMA conv variant)
    Arena arena;
    DevBuf heat, gray, resized;
    DevBuf ccl_label, ccl_stat, ccl_slot, ccl_comps, ccl_rowext, ccl_counters;
    DevBuf crop_desc, crop_scratch, crop_hscratch, crop_wscratch, crop_luts, crop_hist;
    DevBuf ctc_idx, ctc_pmax, ctc_out_idx, ctc_out, ctc_probs, crop_desc2;
    PinBuf desc_pin, desc_pin2;               // staging of crop_desc / crop_desc2 uploads
    PinBuf ctc_pin;                           // CTC results land here (pinned: the 1.7 MB D2H copy of a 64-page pass runs at link speed)
    DevBuf seq_v, seq_xp, seq_h, seq_lin, seq_logits, seq_tables;
};

namespace {

// ------------------------------------------------------------------------------------------------ weights
struct TensorMap {
    std::unordered_map<std::string, const bbocr_tensor_desc*> m;
    TensorMap(const bbocr_tensor_desc* d, int n) {
        for (int i = 0; i < n; ++i) {
            std::string k = d[i].name ? d[i].name : "";
            if (k.rfind("module.", 0) == 0) k = k.substr(7);
            m[k] = &d[i];
        }
    }
    const float* get(const std::string& name, size_t numel, bool required = true) const {
        auto it = m.find(name);
        if (it == m.end()) {
            if (required) fail(BBOCR_ERR_WEIGHTS, "missing tensor '" + name + "'");
            return nullptr;
        }
        size_t n = 1;
        for (int i = 0; i < it->second->ndim; ++i) n *= (size_t)it->second->shape[i];
        if (n != numel || !it->second->data)
            fail(BBOCR_ERR_WEIGHTS, "tensor '" + name + "' has " + std::to_string(n) + " elements, expected " + std::to_string(numel));
        return it->second->data;
    }
};

// conv (+ optional BatchNorm in eval mode) -> folded fp32 weight [Cout][Cin][K] and bias [Cout]
static void fold_conv(const TensorMap& tm, const std::string& conv, const std::string& bn, int Cout, int Cin, int K, std::vector<float>& w,
                      std::vector<float>& b) {
    const size_t per = (size_t)Cin * K;
    const float* cw = tm.get(conv + ".weight", (size_t)Cout * per);
    const float* cb = tm.get(conv + ".bias", (size_t)Cout, false);
    w.assign(cw, cw + (size_t)Cout * per);
    b.assign(Cout, 0.f);
    if (cb) std::copy(cb, cb + Cout, b.begin());
    if (!bn.empty()) {
        const float* g = tm.get(bn + ".weight", Cout);
        const float* be = tm.get(bn + ".bias", Cout);
        const float* mu = tm.get(bn + ".running_mean", Cout);
        const float* var = tm.get(bn + ".running_var", Cout);
        for (int o = 0; o < Cout; ++o) {
            const float sc = g[o] / std::sqrt(var[o] + 1e-5f);
            for (size_t i = 0; i < per; ++i) w[(size_t)o * per + i] *= sc;
            b[o] = (b[o] - mu[o]) * sc + be[o];
        }
    }
}

static ConvPlan make_plan(int Cin, int Cout, int KH, int KW, int pad, int dil) {
    ConvPlan p;
    p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.pad_h = pad; p.pad_w = pad; p.dil = dil;
    p.Cin_pad = cdiv(Cin, 32) * 32;
    p.BN = conv_plan_bn(Cout);
    p.Cout_pad = cdiv(Cout, p.BN) * p.BN;
    return p;
}

template <typename T> static T* upload(bbocr_ctx* c, const std::vector<T>& v) {
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, v.size() * sizeof(T)));
    c->owned.push_back(d);
    HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return (T*)d;
}

static void upload_plan(bbocr_ctx* c, ConvPlan& p, const std::vector<float>& w, const std::vector<float>& b) {
    std::vector<uint16_t> pk(conv_packed_elems(p));
    pack_conv_weights(p, w.data(), pk.data());
    std::vector<float> bp(p.Cout_pad, 0.f);
    std::copy(b.begin(), b.end(), bp.begin());
    p.d_w = upload(c, pk);
    p.d_b = upload(c, bp);
}

// 1x1 conv over a channel concat [y (Cy) | skip (Cs)], BN folded, split into the two column blocks
static void load_split_1x1(bbocr_ctx* c, const TensorMap& tm, ConvPlan& py, ConvPlan& ps, const std::string& conv, const std::string& bn, int Cy,
                           int Cs, int Cout) {
    std::vector<float> w, b;
    fold_conv(tm, conv, bn, Cout, Cy + Cs, 1, w, b);
    std::vector<float> wy((size_t)Cout * Cy), ws((size_t)Cout * Cs), zero(Cout, 0.f);
    for (int o = 0; o < Cout; ++o) {
        std::copy(w.begin() + (size_t)o * (Cy + Cs), w.begin() + (size_t)o * (Cy + Cs) + Cy, wy.begin() + (size_t)o * Cy);
        std::copy(w.begin() + (size_t)o * (Cy + Cs) + Cy, w.begin() + (size_t)(o + 1) * (Cy + Cs), ws.begin() + (size_t)o * Cs);
    }
    py = make_plan(Cy, Cout, 1, 1, 0, 1);
    upload_plan(c, py, wy, zero);
    ps = make_plan(Cs, Cout, 1, 1, 0, 1);
    upload_plan(c, ps, ws, b);
}

static void load_layer(bbocr_ctx* c, const TensorMap& tm, ConvPlan& p, const std::string& conv, const std::string& bn, int Cin, int Cout,
                       int K, int pad, int dil) {
    p = make_plan(Cin, Cout, K, K, pad, dil);
    std::vector<float> w, b;
    fold_conv(tm, conv, bn, Cout, Cin, K * K, w, b);
    upload_plan(c, p, w, b);
}

static void free_weights(bbocr_ctx* c) {
    for (void* p : c->owned) (void)hipFree(p);
    c->owned.clear();
    c->craft_loaded = c->crnn_loaded = false;
}

static void load_craft(bbocr_ctx* c, const TensorMap& tm) {
    {
        std::vector<float> w, b;
        fold_conv(tm, "basenet.slice1.0", "basenet.slice1.1", 64, 3, 9, w, b);
        std::vector<uint16_t> pk(2 * 4 * 64 * 8);
        pack_conv1_1_weights(w.data(), pk.data());
        c->c11_w = upload(c, pk);
        pack_conv1_1_weights_fused(w.data(), pk.data());
        c->c11_wf = upload(c, pk);
        c->c11_b = upload(c, b);
    }
    load_layer(c, tm, c->conv1_2, "basenet.slice1.3", "basenet.slice1.4", 64, 64, 3, 1, 1);
    load_layer(c, tm, c->conv2_1, "basenet.slice1.7", "basenet.slice1.8", 64, 128, 3, 1, 1);
    load_layer(c, tm, c->conv2_2, "basenet.slice1.10", "basenet.slice1.11", 128, 128, 3, 1, 1);
    load_layer(c, tm, c->conv3_1, "basenet.slice2.14", "basenet.slice2.15", 128, 256, 3, 1, 1);
    load_layer(c, tm, c->conv3_2, "basenet.slice2.17", "basenet.slice2.18", 256, 256, 3, 1, 1);
    load_layer(c, tm, c->conv3_3, "basenet.slice3.20", "basenet.slice3.21", 256, 256, 3, 1, 1);
    load_layer(c, tm, c->conv4_1, "basenet.slice3.24", "basenet.slice3.25", 256, 512, 3, 1, 1);
    load_layer(c, tm, c->conv4_2, "basenet.slice3.27", "basenet.slice3.28", 512, 512, 3, 1, 1);
    load_layer(c, tm, c->conv4_3, "basenet.slice4.30", "basenet.slice4.31", 512, 512, 3, 1, 1);
    load_layer(c, tm, c->conv5_1, "basenet.slice4.34", "basenet.slice4.35", 512, 512, 3, 1, 1);
    load_layer(c, tm, c->conv5_2, "basenet.slice4.37", "basenet.slice4.38", 512, 512, 3, 1, 1);
    load_layer(c, tm, c->fc6, "basenet.slice5.1", "", 512, 1024, 3, 6, 6);
    load_layer(c, tm, c->fc7, "basenet.slice5.2", "", 1024, 1024, 1, 0, 1);
    load_layer(c, tm, c->up1a, "upconv1.conv.0", "upconv1.conv.1", 1536, 512, 1, 0, 1);
    load_layer(c, tm, c->up1b, "upconv1.conv.3", "upconv1.conv.4", 512, 256, 3, 1, 1);
    load_split_1x1(c, tm, c->up2y, c->up2s, "upconv2.conv.0", "upconv2.conv.1", 256, 512, 256);
    load_layer(c, tm, c->up2b, "upconv2.conv.3", "upconv2.conv.4", 256, 128, 3, 1, 1);
    load_split_1x1(c, tm, c->up3y, c->up3s, "upconv3.conv.0", "upconv3.conv.1", 128, 256, 128);
    load_layer(c, tm, c->up3b, "upconv3.conv.3", "upconv3.conv.4", 128, 64, 3, 1, 1);
    load_split_1x1(c, tm, c->up4y, c->up4s, "upconv4.conv.0", "upconv4.conv.1", 64, 128, 64);
    load_layer(c, tm, c->up4b, "upconv4.conv.3", "upconv4.conv.4", 64, 32, 3, 1, 1);
    load_layer(c, tm, c->cls0, "conv_cls.0", "", 32, 32, 3, 1, 1);
    load_layer(c, tm, c->cls2, "conv_cls.2", "", 32, 32, 3, 1, 1);
    load_layer(c, tm, c->cls4, "conv_cls.4", "", 32, 16, 3, 1, 1);
    {
        const float* w1 = tm.get("conv_cls.6.weight", 256);
        const float* b1 = tm.get("conv_cls.6.bias", 16);
        const float* w2 = tm.get("conv_cls.8.weight", 32);
        const float* b2 = tm.get("conv_cls.8.bias", 2);
        std::vector<float> t(50);
        std::copy(b1, b1 + 16, t.begin());
        std::copy(w2, w2 + 32, t.begin() + 16);
        std::copy(b2, b2 + 2, t.begin() + 48);
        c->cls_tail = upload(c, t);
        std::vector<uint16_t> fr(64 * 8);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int o = l & 15, k = 8 * (l >> 4) + j;
                fr[l * 8 + j] = f32_to_bf16_host(k < 16 ? w1[o * 16 + k] : 0.f);
            }
        c->cls_tail_frag = upload(c, fr);
    }
    c->craft_loaded = true;
}

static void load_crnn(bbocr_ctx* c, const TensorMap& tm) {
    const std::string fe = "FeatureExtraction.ConvNet.";
    {
        std::vector<float> t(32 * 9 + 32);
        const float* w = tm.get(fe + "0.weight", 32 * 9);
        const float* b = tm.get(fe + "0.bias", 32);
        for (int co = 0; co < 32; ++co)                       // tap-major [9][32]: the kernel reads channel PAIRS of one tap (packed FMA)
            for (int tap = 0; tap < 9; ++tap) t[tap * 32 + co] = w[co * 9 + tap];
        std::copy(b, b + 32, t.begin() + 288);
        c->r0_wb = upload(c, t);
    }
    load_layer(c, tm, c->r1, fe + "3", "", 32, 64, 3, 1, 1);
    load_layer(c, tm, c->r2, fe + "6", "", 64, 128, 3, 1, 1);
    load_layer(c, tm, c->r3, fe + "8", "", 128, 128, 3, 1, 1);
    load_layer(c, tm, c->r4, fe + "11", fe + "12", 128, 256, 3, 1, 1);
    load_layer(c, tm, c->r5, fe + "14", fe + "15", 256, 256, 3, 1, 1);
    load_layer(c, tm, c->r6, fe + "18", "", 256, 256, 2, 0, 1);
    for (int l = 0; l < 2; ++l) {
        const std::string sm = "SequenceModeling." + std::to_string(l) + ".";
        // input projection of both directions as one 1x1 conv with the channel permutation the LSTM kernel reads
        std::vector<float> w((size_t)2048 * 256), b(2048);
        for (int d = 0; d < 2; ++d) {
            const std::string sfx = d ? "_reverse" : "";
            const float* wih = tm.get(sm + "rnn.weight_ih_l0" + sfx, (size_t)1024 * 256);
            const float* bih = tm.get(sm + "rnn.bias_ih_l0" + sfx, 1024);
            const float* bhh = tm.get(sm + "rnn.bias_hh_l0" + sfx, 1024);
            for (int g = 0; g < 4; ++g)
                for (int u = 0; u < 256; ++u) {
                    const int src = g * 256 + u, dst = lstm8_xproj_channel(d, g, u);
                    std::copy(wih + (size_t)src * 256, wih + (size_t)src * 256 + 256, w.begin() + (size_t)dst * 256);
                    b[dst] = bih[src] + bhh[src];
                }
        }
        c->xproj[l] = make_plan(256, 2048, 1, 1, 0, 1);
        upload_plan(c, c->xproj[l], w, b);
        const float* hf = tm.get(sm + "rnn.weight_hh_l0", (size_t)1024 * 256);
        const float* hb = tm.get(sm + "rnn.weight_hh_l0_reverse", (size_t)1024 * 256);
        std::vector<uint16_t> pk(lstm_whh_packed_elems());
        pack_lstm_whh8(hf, hb, pk.data());
        c->whh[l] = upload(c, pk);
        std::vector<float> lw(tm.get(sm + "linear.weight", (size_t)256 * 512), tm.get(sm + "linear.weight", (size_t)256 * 512) + 256 * 512);
        std::vector<float> lb(tm.get(sm + "linear.bias", 256), tm.get(sm + "linear.bias", 256) + 256);
        c->lin[l] = make_plan(512, 256, 1, 1, 0, 1);
        upload_plan(c, c->lin[l], lw, lb);
    }
    {
        std::vector<float> pw(tm.get("Prediction.weight", (size_t)97 * 256), tm.get("Prediction.weight", (size_t)97 * 256) + 97 * 256);
        std::vector<float> pb(tm.get("Prediction.bias", 97), tm.get("Prediction.bias", 97) + 97);
        c->pred = make_plan(256, 97, 1, 1, 0, 1);
        upload_plan(c, c->pred, pw, pb);
    }
    c->crnn_loaded = true;
}

// ------------------------------------------------------------------------------------------------ conv helper
static void launch_conv_profiled(bbocr_ctx* c, const ConvPlan& p, ConvArgs a) {
    a.zero = c->zero_page;
    if (c->profiling == 0 || (c->profiling == 1 && c->prof_group != 0)) {
        HIPCHK(launch_conv(p, a, c->cur));
        return;
    }
    auto get_event = [&]() {
        hipEvent_t e;
        if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
        else HIPCHK(hipEventCreate(&e));
        return e;
    };
    bbocr_ctx::ProfRec r;
    r.e0 = get_event();
    r.e1 = get_event();
    const int OH = a.H + 2 * p.pad_h - (p.KH - 1) * p.dil, OW = a.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    r.flops = 2.0 * a.N * OH * OW * (double)p.Cout * p.Cin * p.KH * p.KW;   // algorithmic (unpadded) work
    if (a.c11_w) r.flops += 2.0 * a.N * a.H * a.W * 64.0 * 27.0;            // conv1_1 produced inside this launch
    if (a.tail) r.flops += 2.0 * a.N * OH * OW * (16.0 * 16.0 + 16.0 * 2.0);   // fused classifier tail
    r.group = c->prof_group;
    HIPCHK(hipEventRecord(r.e0, c->cur));
    HIPCHK(launch_conv(p, a, c->cur));
    HIPCHK(hipEventRecord(r.e1, c->cur));
    c->prof_recs.push_back(r);
}

static void run_conv(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, const Act* a1, bool relu1, bool relu_out, void* out,
                     int out_cs, int cout_store, bool out_f32, const Act* addup = nullptr) {
    if (c->arena.dry) return;
    ConvArgs a{};
    if (addup) { a.addup = addup->p; a.up_H = a0.H; a.up_W = a0.W; a.up_cs = addup->C; }
    a.in0 = a0.p; a.C0 = a0.C; a.in0_cs = a0.C;
    if (a1) { a.in1 = a1->p; a.C1 = a1->C; a.in1_cs = a1->C; }
    a.N = a0.N; a.H = a0.H; a.W = a0.W;
    a.relu_in0 = relu0; a.relu_in1 = relu1; a.relu_out = relu_out; a.out_f32 = out_f32;
    a.out = out; a.out_cs = out_cs; a.cout_store = cout_store;
    launch_conv_profiled(c, p, a);
}

// after the stream has drained: fold the recorded launches into the per-group totals
static void prof_collect(bbocr_ctx* c) {
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            c->prof_ms[r.group] += ms;
            c->prof_flops[r.group] += r.flops;
            c->prof_launches[r.group] += 1;
        }
        c->prof_pool.push_back(r.e0);
        c->prof_pool.push_back(r.e1);
    }
    c->prof_recs.clear();
}

// conv producing a fresh bf16 activation with `store` channels (multiple of 16)
static Act conv_act(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, const Act* a1, bool relu1, bool relu_out, int store) {
    const int OH = a0.H + 2 * p.pad_h - (p.KH - 1) * p.dil, OW = a0.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    Act o{c->arena.alloc<uint16_t>((size_t)a0.N * OH * OW * store), a0.N, OH, OW, store};
    run_conv(c, p, a0, relu0, a1, relu1, relu_out, o.p, store, store, false);
    return o;
}

// conv with the max-pool fused into its epilogue.  mode 1 = MaxPool2d(2,2), 2 = MaxPool2d((2,1),(2,1)).  Returns the pooled
// activation; when `full` is given the un-pooled conv output (bias, relu_out) is written too (U-net skip tensors).
struct RgbSource { const uint8_t* rgb; int Himg, Wimg; };   // conv1_2 with conv1_1 fused in: a0 then only carries the canvas shape
static Act conv_pool_act(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, bool relu_out, int store, int mode, bool pool_relu,
                         Act* full, const RgbSource* rgb = nullptr) {
    const int OH = a0.H + 2 * p.pad_h - (p.KH - 1) * p.dil, OW = a0.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    const int PH = OH / 2, PW = mode == 1 ? OW / 2 : OW;
    if (full) *full = Act{c->arena.alloc<uint16_t>((size_t)a0.N * OH * OW * store), a0.N, OH, OW, store};
    Act o{c->arena.alloc<uint16_t>((size_t)a0.N * PH * PW * store), a0.N, PH, PW, store};
    if (c->arena.dry) return o;
    ConvArgs a{};
    a.in0 = a0.p; a.C0 = a0.C; a.in0_cs = a0.C;
    a.N = a0.N; a.H = a0.H; a.W = a0.W;
    a.relu_in0 = relu0; a.relu_out = relu_out; a.out_f32 = 0;
    a.out = full ? (void*)full->p : nullptr; a.out_cs = store; a.cout_store = store;
    a.pool_mode = mode; a.pool_relu = pool_relu; a.store_full = full != nullptr; a.pool_cs = store; a.pool_out = o.p;
    if (rgb) { a.in0 = (const uint16_t*)rgb->rgb; a.c11_w = c->c11_wf; a.c11_b = c->c11_b; a.rgb_H = rgb->Himg; a.rgb_W = rgb->Wimg; }
    launch_conv_profiled(c, p, a);
    return o;
}

static Act pool_act(bbocr_ctx* c, const Act& a, int kh, int kw, int sh, int sw, int ph, int pw, bool relu_in) {
    const int OH = (a.H + 2 * ph - kh) / sh + 1, OW = (a.W + 2 * pw - kw) / sw + 1;
    Act o{c->arena.alloc<uint16_t>((size_t)a.N * OH * OW * a.C), a.N, OH, OW, a.C};
    if (!c->arena.dry) HIPCHK(launch_maxpool(a.p, o.p, a.N, a.H, a.W, a.C, kh, kw, sh, sw, ph, pw, relu_in, c->cur));
    return o;
}

// ------------------------------------------------------------------------------------------------ detector
// rgb: [nb, Himg, Wimg, 3] on a zero canvas H32 x W32 -> heat fp32 [nb, H32/2, W32/2, 2]
static void craft_forward(bbocr_ctx* c, const uint8_t* rgb, int nb, int Himg, int Wimg, int H32, int W32, float* heat) {
    Arena& ar = c->arena;
    c->prof_group = 0;
    // normalise + conv1_1 + ReLU are produced inside conv1_2's prologue (its 64-channel input never reaches HBM);
    // BBOCR_FUSE1=0 runs conv1_1 as its own kernel (A/B runs)
    static const bool fuse1 = [] { const char* e = getenv("BBOCR_FUSE1"); return !(e && e[0] == '0'); }();
    Act p1;
    if (fuse1) {
        const Act canvas{nullptr, nb, H32, W32, 64};
        const RgbSource src{rgb, Himg, Wimg};
        p1 = conv_pool_act(c, c->conv1_2, canvas, false, true, 64, 1, false, nullptr, &src);
    } else {
        Act a1{ar.alloc<uint16_t>((size_t)nb * H32 * W32 * 64), nb, H32, W32, 64};
        if (!ar.dry) HIPCHK(launch_conv1_1(rgb, nb, Himg, Wimg, H32, W32, c->c11_w, c->c11_b, a1.p, c->cur));
        p1 = conv_pool_act(c, c->conv1_2, a1, false, true, 64, 1, false, nullptr);          // conv1_2+BN+ReLU+pool fused
    }
    Act a3 = conv_act(c, c->conv2_1, p1, false, nullptr, false, true, 128);
    Act s1;                                                                        // slice1 ends on BatchNorm (skip tensor),
    Act p2 = conv_pool_act(c, c->conv2_2, a3, false, false, 128, 1, true, &s1);   // slice2 opens with ReLU + pool: both fused
    Act a5 = conv_act(c, c->conv3_1, p2, false, nullptr, false, true, 256);
    Act s2 = conv_act(c, c->conv3_2, a5, false, nullptr, false, false, 256);
    Act p3 = conv_pool_act(c, c->conv3_3, s2, true, true, 256, 1, false, nullptr);  // ReLU applied on load; pool fused
    Act a8 = conv_act(c, c->conv4_1, p3, false, nullptr, false, true, 512);
    Act s3 = conv_act(c, c->conv4_2, a8, false, nullptr, false, false, 512);
    Act p4 = conv_pool_act(c, c->conv4_3, s3, true, true, 512, 1, false, nullptr);
    Act a11 = conv_act(c, c->conv5_1, p4, false, nullptr, false, true, 512);
    Act s4 = conv_act(c, c->conv5_2, a11, false, nullptr, false, false, 512);
    Act p5 = pool_act(c, s4, 3, 3, 1, 1, 1, 1, false);                             // slice5: MaxPool(3,1,1), no ReLU
    Act f6 = conv_act(c, c->fc6, p5, false, nullptr, false, false, 1024);
    Act f7 = conv_act(c, c->fc7, f6, false, nullptr, false, false, 1024);
    Act u1a = conv_act(c, c->up1a, f7, false, &s4, false, true, 512);              // cat([fc7, relu5_3]) -> 1x1
    Act u1b = conv_act(c, c->up1b, u1a, false, nullptr, false, true, 256);
    // cat([up(y), skip]) -> 1x1 + BN + ReLU, with the up-sampling commuted behind the (linear) 1x1: z = W_y y at y's
    // resolution, then ReLU(up(z) + W_s skip + b) in the epilogue of the skip half -- up(y) is never written
    auto up_stage = [&](const ConvPlan& py, const ConvPlan& ps, const Act& y, const Act& skip, bool relu_skip, int cout) {
        Act z = conv_act(c, py, y, false, nullptr, false, false, cout);
        Act o{c->arena.alloc<uint16_t>((size_t)skip.N * skip.H * skip.W * cout), skip.N, skip.H, skip.W, cout};
        run_conv(c, ps, skip, relu_skip, nullptr, false, true, o.p, cout, cout, false, &z);
        return o;
    };
    Act u2a = up_stage(c->up2y, c->up2s, u1b, s3, false, 256);
    Act u2b = conv_act(c, c->up2b, u2a, false, nullptr, false, true, 128);
    Act u3a = up_stage(c->up3y, c->up3s, u2b, s2, false, 128);
    Act u3b = conv_act(c, c->up3b, u3a, false, nullptr, false, true, 64);
    Act u4a = up_stage(c->up4y, c->up4s, u3b, s1, false, 64);
    Act u4b = conv_act(c, c->up4b, u4a, false, nullptr, false, true, 32);
    Act c1 = conv_act(c, c->cls0, u4b, false, nullptr, false, true, 32);
    Act c2 = conv_act(c, c->cls2, c1, false, nullptr, false, true, 32);
    // conv_cls.4 (3x3 32->16 + ReLU) with conv_cls.6/.8 fused into its epilogue: writes the fp32 heat-map directly
    if (!ar.dry) {
        ConvArgs a{};
        a.in0 = c2.p; a.C0 = c2.C; a.in0_cs = c2.C;
        a.N = c2.N; a.H = c2.H; a.W = c2.W;
        a.relu_out = 1; a.out = heat; a.out_cs = 16; a.cout_store = 16; a.tail = c->cls_tail; a.tail_frag = c->cls_tail_frag;
        launch_conv_profiled(c, c->cls4, a);
    }
}

struct DetDims {
    int H32, W32, h, w, th, tw;
    double ratio;
};
static DetDims det_dims(int H, int W, int canvas, double mag) {
    DetDims d;
    double target = mag * (double)std::max(H, W);
    if (target > canvas) target = canvas;
    d.ratio = target / (double)std::max(H, W);
    d.th = (int)(H * d.ratio);
    d.tw = (int)(W * d.ratio);
    d.H32 = d.th % 32 ? d.th + (32 - d.th % 32) : d.th;
    d.W32 = d.tw % 32 ? d.tw + (32 - d.tw % 32) : d.tw;
    d.h = d.H32 / 2;
    d.w = d.W32 / 2;
    return d;
}

static void detect_impl(bbocr_ctx* c, const uint8_t* rgb, int B, int H, int W, const bbocr_params& p, float* heat,
                        const std::function<void(int, int)>& after_sub = nullptr) {
    if (!c->craft_loaded) fail(BBOCR_ERR_STATE, "detector weights not loaded");
    if (B <= 0 || H <= 0 || W <= 0) fail(BBOCR_ERR_ARG, "bad page batch shape");
    const DetDims d = det_dims(H, W, p.canvas_size, p.mag_ratio);
    if (d.th <= 0 || d.tw <= 0) fail(BBOCR_ERR_ARG, "page collapses to zero size");
    // Pages per detector pass.  Explicit det_sub_batch: uniform passes of that size.  Auto: passes as large as a 96 GB
    // activation arena allows (sized by a dry run on one page; at most 64 pages), and -- when the caller overlaps box
    // extraction with the next pass (readtext_batch) -- a short last pass of 8 pages, because only the LAST pass's
    // CCL + host geometry is exposed: 64 pages run as [56, 8].
    std::vector<int> passes;
    if (c->cfg.det_sub_batch > 0) {
        for (int b0 = 0; b0 < B; b0 += c->cfg.det_sub_batch) passes.push_back(std::min(c->cfg.det_sub_batch, B - b0));
    } else {
        c->arena.begin(true);
        craft_forward(c, nullptr, 1, d.th, d.tw, d.H32, d.W32, nullptr);
        const size_t per_page = std::max<size_t>(c->arena.off, 1);
        const int cap = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)96 << 30) / per_page));
        static const int tail_pages = [] { const char* e = getenv("BBOCR_DET_TAIL"); return e ? atoi(e) : 8; }();   // A/B knob
        const int tail = (after_sub && B >= 24 && cap > tail_pages && tail_pages > 0) ? tail_pages : 0;
        const int body = B - tail, nbig = cdiv(body, cap);
        for (int i = 0; i < nbig; ++i) passes.push_back(body / nbig + (i < body % nbig ? 1 : 0));
        if (tail) passes.push_back(tail);
    }
    const int sb = *std::max_element(passes.begin(), passes.end());
    const bool need_resize = (d.th != H || d.tw != W);
    if (need_resize) c->resized.ensure((size_t)sb * d.th * d.tw * 3);
    c->arena.begin(true);
    craft_forward(c, nullptr, sb, d.th, d.tw, d.H32, d.W32, nullptr);
    c->arena.buf.ensure(c->arena.off);
    int b0 = 0;
    for (const int nb : passes) {
        const uint8_t* src = rgb + (size_t)b0 * H * W * 3;
        if (need_resize) {
            HIPCHK(launch_resize_u8(src, nb, H, W, 3, (uint8_t*)c->resized.p, d.th, d.tw, c->stream));
            src = (const uint8_t*)c->resized.p;
        }
        c->arena.begin(false);
        craft_forward(c, src, nb, d.th, d.tw, d.H32, d.W32, heat + (size_t)b0 * d.h * d.w * 2);
        if (after_sub) after_sub(b0, nb);   // everything of this sub-batch is enqueued (nothing has been waited for)
        b0 += nb;
    }
}

// ------------------------------------------------------------------------------------------------ boxes
struct HostBoxes {
    std::vector<std::vector<std::array<int, 8>>> polys;
    std::vector<std::vector<std::array<int, 4>>> hori;
    std::vector<std::vector<std::array<double, 8>>> freeb;
};

static void boxes_impl(bbocr_ctx* c, const float* heat, int B, int h, int w, double ratio, const bbocr_params& p, HostBoxes& hb,
                       hipStream_t st) {
    if (B <= 0 || h <= 0 || w <= 0 || !(ratio > 0)) fail(BBOCR_ERR_ARG, "bad heat-map shape");
    const size_t npx = (size_t)B * h * w;
    // theoretical maxima, so that no heat-map can overflow them: an accepted component has >= 10 pixels (getDetBoxes_core's size
    // filter), and the (component, row) extents cannot outnumber the pixels
    const int cap_comps = (int)std::min<size_t>(0x3fffffff, (size_t)B * ((size_t)h * w / 10 + 1));
    const int cap_rows = (int)std::min<size_t>(0x3fffffff, npx);
    c->ccl_label.ensure(npx * 4);
    c->ccl_stat.ensure(npx * 24);
    c->ccl_slot.ensure(npx * 4);
    c->ccl_comps.ensure((size_t)cap_comps * sizeof(CclOut));
    c->ccl_rowext.ensure((size_t)cap_rows * 8);
    c->ccl_counters.ensure(16);
    if (!c->ccl_t0) { HIPCHK(hipEventCreate(&c->ccl_t0)); HIPCHK(hipEventCreate(&c->ccl_t1)); }
    HIPCHK(hipEventRecord(c->ccl_t0, st));
    HIPCHK(launch_ccl(heat, B, h, w, (float)p.low_text, (float)p.link_threshold, (double)p.text_threshold, (int*)c->ccl_label.p,
                      (int*)c->ccl_stat.p, (int*)c->ccl_slot.p, (CclOut*)c->ccl_comps.p, (int*)c->ccl_rowext.p, (int*)c->ccl_counters.p,
                      cap_comps, cap_rows, st));
    HIPCHK(hipEventRecord(c->ccl_t1, st));
    int counters[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(counters, c->ccl_counters.p, sizeof(counters), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (counters[2]) fail(BBOCR_ERR_OVERFLOW, "component buffers too small for this batch");
    std::vector<CclOut> all_comps(counters[0]);
    std::vector<int> all_rows((size_t)counters[1] * 2);
    if (counters[0]) HIPCHK(hipMemcpyAsync(all_comps.data(), c->ccl_comps.p, all_comps.size() * sizeof(CclOut), hipMemcpyDeviceToHost, st));
    if (counters[1]) HIPCHK(hipMemcpyAsync(all_rows.data(), c->ccl_rowext.p, all_rows.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<std::vector<CclOut>> comps(B);
    for (const CclOut& co : all_comps)
        if (co.img >= 0 && co.img < B) comps[co.img].push_back(co);
    float ccl_ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ccl_ms, c->ccl_t0, c->ccl_t1));
    c->times[1] += ccl_ms;     // GPU span of the CCL kernels (the host wait before it may include the detector of this sub-batch)
    auto t0 = clk::now();
    const double ratio_w = 1.0 / ratio, ratio_h = 1.0 / ratio;
    bbocr::GroupParams gp{p.slope_ths, p.ycenter_ths, p.height_ths, p.width_ths, p.add_margin, p.min_size};
    hb.polys.assign(B, {});
    hb.hori.assign(B, {});
    hb.freeb.assign(B, {});
    // pages are independent: fan the O(#components) geometry out over a few host threads
    auto do_page = [&](int b) {
        std::sort(comps[b].begin(), comps[b].end(), [](const CclOut& x, const CclOut& y) { return x.root < y.root; });
        for (const CclOut& co : comps[b]) {
            bbocr::Component cc{co.root, co.left, co.top, co.right, co.bottom, co.area, co.row_off};
            float box[4][2];
            bbocr::component_box(cc, all_rows.data() + (size_t)co.row_off * 2, w, h, box);
            std::array<int, 8> poly;
            bbocr::box_to_poly(box, ratio_w, ratio_h, poly.data());
            hb.polys[b].push_back(poly);
        }
        bbocr::group_text_box(hb.polys[b], gp, hb.hori[b], hb.freeb[b]);
    };
    const int nthr = std::max(1, std::min({B, 16, (int)std::thread::hardware_concurrency()}));
    if (nthr <= 1) {
        for (int b = 0; b < B; ++b) do_page(b);
    } else {
        std::atomic<int> next{0};
        std::vector<std::thread> pool;
        std::exception_ptr err;
        std::mutex err_mu;
        for (int t = 0; t < nthr; ++t)
            pool.emplace_back([&] {
                try {
                    for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) do_page(b);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    err = std::current_exception();
                }
            });
        for (auto& th : pool) th.join();
        if (err) std::rethrow_exception(err);
    }
    c->times[2] += (float)ms_since(t0);
}

static bbocr_boxlist* export_boxes(const HostBoxes& hb) {
    const int B = (int)hb.polys.size();
    bbocr_boxlist* o = (bbocr_boxlist*)calloc(1, sizeof(bbocr_boxlist));
    o->n_images = B;
    o->poly_off = (int*)calloc(B + 1, sizeof(int));
    o->hori_off = (int*)calloc(B + 1, sizeof(int));
    o->free_off = (int*)calloc(B + 1, sizeof(int));
    for (int b = 0; b < B; ++b) {
        o->poly_off[b + 1] = o->poly_off[b] + (int)hb.polys[b].size();
        o->hori_off[b + 1] = o->hori_off[b] + (int)hb.hori[b].size();
        o->free_off[b + 1] = o->free_off[b] + (int)hb.freeb[b].size();
    }
    o->polys = (int*)calloc((size_t)std::max(1, o->poly_off[B]) * 8, sizeof(int));
    o->hori = (int*)calloc((size_t)std::max(1, o->hori_off[B]) * 4, sizeof(int));
    o->free_q = (double*)calloc((size_t)std::max(1, o->free_off[B]) * 8, sizeof(double));
    for (int b = 0; b < B; ++b) {
        for (size_t i = 0; i < hb.polys[b].size(); ++i) memcpy(o->polys + ((size_t)o->poly_off[b] + i) * 8, hb.polys[b][i].data(), 32);
        for (size_t i = 0; i < hb.hori[b].size(); ++i) memcpy(o->hori + ((size_t)o->hori_off[b] + i) * 4, hb.hori[b][i].data(), 16);
        for (size_t i = 0; i < hb.freeb[b].size(); ++i) memcpy(o->free_q + ((size_t)o->free_off[b] + i) * 8, hb.freeb[b][i].data(), 64);
    }
    return o;
}

static void import_boxes(const bbocr_boxlist* bl, HostBoxes& hb) {
    const int B = bl->n_images;
    hb.polys.assign(B, {});
    hb.hori.assign(B, {});
    hb.freeb.assign(B, {});
    for (int b = 0; b < B; ++b) {
        for (int i = bl->hori_off[b]; i < bl->hori_off[b + 1]; ++i) {
            std::array<int, 4> a;
            memcpy(a.data(), bl->hori + (size_t)i * 4, 16);
            hb.hori[b].push_back(a);
        }
        for (int i = bl->free_off[b]; i < bl->free_off[b + 1]; ++i) {
            std::array<double, 8> a;
            memcpy(a.data(), bl->free_q + (size_t)i * 8, 64);
            hb.freeb[b].push_back(a);
        }
    }
}

// ------------------------------------------------------------------------------------------------ recogniser
struct BoxJob {            // one box to recognise
    int img;
    bool is_free;
    double quad[8];        // reported corners
    CropDesc d;
    std::vector<int> text;
    double conf = 0.0;
};

// AlignCollate / get_image_list geometry of one horizontal box; false = skipped (degenerate)
static bool plan_horizontal(const std::array<int, 4>& box, int img, int H, int W, BoxJob& j) {
    const int x_min = std::max(0, box[0]), x_max = std::min(box[1], W), y_min = std::max(0, box[2]), y_max = std::min(box[3], H);
    const int width = x_max - x_min, height = y_max - y_min;
    if (width <= 0 || height <= 0) return false;
    j.img = img;
    j.is_free = false;
    const double q[8] = {(double)x_min, (double)y_min, (double)x_max, (double)y_min, (double)x_max, (double)y_max, (double)x_min, (double)y_max};
    memcpy(j.quad, q, sizeof(q));
    CropDesc& d = j.d;
    memset(&d, 0, sizeof(d));
    d.img = img; d.sx0 = x_min; d.sy0 = y_min; d.sw = width; d.sh = height; d.warp = 0; d.lut_off = -1;
    double ratio = (double)width / (double)height;
    if (ratio < 1.0) {
        ratio = 1.0 / ratio;
        d.rw = 64; d.rh = (int)(64 * ratio);
    } else {
        d.rw = (int)(64 * ratio); d.rh = 64;
    }
    if ((int)(64 * ratio) == 0) return false;
    d.imgW = (int)std::ceil(std::max(ratio, 1.0)) * 64;
    const double r2 = (double)d.rw / (double)d.rh;
    const int cw = (int)std::ceil(64 * r2);
    d.fw = cw > d.imgW ? d.imgW : cw;
    return d.rw > 0 && d.rh > 0 && d.fw > 0;
}

static bool plan_free(const std::array<double, 8>& fq, int img, BoxJob& j) {
    float rect[4][2];
    for (int i = 0; i < 4; ++i) { rect[i][0] = (float)fq[2 * i]; rect[i][1] = (float)fq[2 * i + 1]; }
    auto dist = [&](int a, int b) {
        const float dx = rect[a][0] - rect[b][0], dy = rect[a][1] - rect[b][1];
        const float t0 = dx * dx, t1 = dy * dy;
        return std::sqrt(t0 + t1);   // float32, as numpy on a float32 array
    };
    const int maxW = std::max((int)dist(2, 3), (int)dist(1, 0));
    const int maxH = std::max((int)dist(1, 2), (int)dist(0, 3));
    if (maxW <= 0 || maxH <= 0) return false;
    j.img = img;
    j.is_free = true;
    memcpy(j.quad, fq.data(), 64);
    CropDesc& d = j.d;
    memset(&d, 0, sizeof(d));
    d.img = img; d.sx0 = 0; d.sy0 = 0; d.sw = maxW; d.sh = maxH; d.warp = 1; d.lut_off = -1;
    bbocr::perspective_inverse(rect, maxW, maxH, d.Minv);
    double ratio = (double)maxW / (double)maxH;
    if (ratio < 1.0) {
        ratio = 1.0 / ratio;
        d.rw = 64; d.rh = (int)(64 * ratio);
    } else {
        d.rw = (int)(64 * ratio); d.rh = 64;
    }
    if ((int)(64 * ratio) == 0) return false;
    d.imgW = (int)std::ceil(std::max(ratio, 1.0)) * 64;
    const double r2 = (double)d.rw / (double)d.rh;
    const int cw = (int)std::ceil(64 * r2);
    d.fw = cw > d.imgW ? d.imgW : cw;
    return d.rw > 0 && d.rh > 0 && d.fw > 0;
}

// conv stack of the recogniser for n normalised crops of one padded width: bf16 [n,64,imgW] -> v bf16 [n*T, 256]
static void crnn_features(bbocr_ctx* c, const uint16_t* crops, int n, int imgW, uint16_t* v_out) {
    Arena& ar = c->arena;
    c->prof_group = 1;
    const int T = imgW / 4 - 1;
    Act c0{ar.alloc<uint16_t>((size_t)n * 32 * (imgW / 2) * 32), n, 32, imgW / 2, 32};
    if (!ar.dry) HIPCHK(launch_crnn_conv0(crops, c->r0_wb, c->r0_wb + 288, c0.p, n, imgW, c->cur));
    Act q1 = conv_pool_act(c, c->r1, c0, false, true, 64, 1, false, nullptr);
    Act c2 = conv_act(c, c->r2, q1, false, nullptr, false, true, 128);
    Act q2 = conv_pool_act(c, c->r3, c2, false, true, 128, 2, false, nullptr);
    Act c4 = conv_act(c, c->r4, q2, false, nullptr, false, true, 256);
    Act q3 = conv_pool_act(c, c->r5, c4, false, true, 256, 2, false, nullptr);
    Act c6 = conv_act(c, c->r6, q3, false, nullptr, false, true, 256);   // [n,3,T,256]
    if (!ar.dry) HIPCHK(launch_rowmean3(c6.p, v_out, n, T, 256, c->cur));
}

// The same conv stack over the WIDE image of a recognition pass: every crop side by side with 4 zero columns between
// neighbours (CropDesc::slot = first column), so each layer is ONE launch over [H, Wt] whatever the mix of width buckets.  The
// separator columns are each layer's zero padding; convolutions write into them, so they are cleared on every layer output
// (4 >> shift columns per crop).  The 3-row mean is gathered straight into every crop's pooled rows (CropDesc::pad_).
static void crnn_features_wide(bbocr_ctx* c, const uint16_t* wide, int Wt, const CropDesc* descs, int first, int count, uint16_t* seq_v) {
    Arena& ar = c->arena;
    c->prof_group = 1;
    Act c0{ar.alloc<uint16_t>((size_t)32 * (Wt / 2) * 32), 1, 32, Wt / 2, 32};
    if (!ar.dry) HIPCHK(launch_crnn_conv0(wide, c->r0_wb, c->r0_wb + 288, c0.p, 1, Wt, c->cur));
    auto gaps = [&](const Act& a, int shift) {
        if (!ar.dry) HIPCHK(launch_crnn_zero_gaps(a.p, descs, first, count, a.H, a.W, a.C, shift, c->cur));
    };
    gaps(c0, 1);
    Act q1 = conv_pool_act(c, c->r1, c0, false, true, 64, 1, false, nullptr);
    gaps(q1, 2);
    Act c2 = conv_act(c, c->r2, q1, false, nullptr, false, true, 128);
    gaps(c2, 2);
    Act q2 = conv_pool_act(c, c->r3, c2, false, true, 128, 2, false, nullptr);
    gaps(q2, 2);
    Act c4 = conv_act(c, c->r4, q2, false, nullptr, false, true, 256);
    gaps(c4, 2);
    Act q3 = conv_pool_act(c, c->r5, c4, false, true, 256, 2, false, nullptr);
    gaps(q3, 2);
    Act c6 = conv_act(c, c->r6, q3, false, nullptr, false, true, 256);   // [1, 3, Wt/4 - 1, 256]
    if (!ar.dry) HIPCHK(launch_rowmean3_gather(c6.p, c6.W, 256, descs, first, count, seq_v, c->cur));
}

// Sequence half of the recogniser over the POOLED time steps of every bucket (rows = sum n_i*T_i, padded to x256):
// v bf16 [rows,256] (ctx->seq_v) -> logits fp32 [rows,112].  The two linear layers and both input projections are
// single GEMMs over all rows; each BiLSTM layer is ONE launch whose workgroups carry their own sequence length.
static void crnn_sequence(bbocr_ctx* c, size_t rows_pad, const int* tiles_dev, int ntiles, float* logits) {
    c->prof_group = 1;
    c->arena.dry = false;
    c->seq_xp.ensure(rows_pad * 2048 * 2);
    c->seq_h.ensure(rows_pad * 512 * 2);
    c->seq_lin.ensure(rows_pad * 256 * 2);
    const int Hh = (int)(rows_pad / 256);
    Act cur{(uint16_t*)c->seq_v.p, 1, Hh, 256, 256};
    for (int l = 0; l < 2; ++l) {
        run_conv(c, c->xproj[l], cur, false, nullptr, false, false, c->seq_xp.p, 2048, 2048, false);
        HIPCHK(launch_lstm((const uint16_t*)c->seq_xp.p, c->whh[l], (uint16_t*)c->seq_h.p, tiles_dev, ntiles, c->stream));
        Act hh{(uint16_t*)c->seq_h.p, 1, Hh, 256, 512};
        uint16_t* dst = (uint16_t*)(l == 0 ? c->seq_lin.p : c->seq_v.p);
        run_conv(c, c->lin[l], hh, false, nullptr, false, false, dst, 256, 256, false);
        cur.p = dst;
    }
    run_conv(c, c->pred, cur, false, nullptr, false, false, logits, 112, 112, true);
}

struct RecChunk {
    int imgW, T, first, n;   // descriptors [first, first+n) share the padded width imgW
    size_t row0;             // first pooled row of the chunk in the pass's sequence tensors
};

// A recognition pass = feature PARTS + one sequence stage.  A part is a set of crops standing side by side in ONE wide image
// (CropDesc::slot = first column, 4 zero columns after every crop; ::pad_ = first pooled row), so each CRNN conv layer is a single
// launch per part whatever the mix of widths; every part gathers its pooled time steps into the pass's shared [rows,256] tensor,
// and the sequence stage (input projections, both BiLSTM layers, linear layers, class projection, CTC) then runs ONCE over all rows.
// readtext_batch uses two parts: the crops of the first detector pass's pages go through the conv stack while the last pass's boxes
// are still being extracted (CCL + host geometry), so that stretch no longer leaves the card idle.
struct RecPart {
    std::vector<CropDesc> descs;
    std::vector<int> order;          // position in the pass's result vectors for each descriptor
    std::vector<RecChunk> chunks;
    size_t rows = 0, cols = 0;       // pooled rows / wide-image columns of this part
    bool any_warp = false, any_tall = false;
};
struct RecRun {
    std::vector<int> tiles, seqs, seq_k;   // int4 {row0, n, T, 0} per LSTM workgroup; int2 {row0, T} and result position per sequence
    size_t rows = 0;
    int n_results = 0;
};
constexpr int REC_GAP = 4;
// pooled time steps per sequence pass (~6.6 KB of work buffers each); bbocr_config::rec_max_cols (pixel columns, 4 per time step) overrides
static size_t rec_max_rows(const bbocr_ctx* c) { return c->cfg.rec_max_cols > 0 ? (size_t)std::max(64, c->cfg.rec_max_cols / 4) : (size_t)1500000; }

// lay the crops `sel` (indices into jobs; result position = res0 + position in sel) out as one part whose rows start at row_base
static void rec_plan_part(const std::vector<BoxJob>& jobs, const std::vector<int>& sel, int res0, size_t row_base, RecPart& part) {
    std::map<int, std::vector<int>> buckets;          // by padded width, box order kept inside a bucket
    for (size_t k = 0; k < sel.size(); ++k) buckets[jobs[sel[k]].d.imgW].push_back((int)k);
    size_t rows = row_base, cols = 0;
    for (auto& kv : buckets) {
        const int imgW = kv.first, T = imgW / 4 - 1;
        RecChunk ch{imgW, T, (int)part.descs.size(), (int)kv.second.size(), rows};
        for (int i = 0; i < ch.n; ++i) {
            const int k = kv.second[i];
            CropDesc d = jobs[sel[k]].d;
            if (cols > 0x7ff00000u) fail(BBOCR_ERR_OVERFLOW, "recogniser pass wider than 2^31 columns");
            d.slot = (int)cols;
            d.pad_ = (int)(rows + (size_t)i * T);
            cols += (size_t)imgW + REC_GAP;
            part.any_warp |= d.warp != 0;
            part.any_tall |= !(d.fw == d.rw && d.rh == 64);
            part.descs.push_back(d);
            part.order.push_back(res0 + k);
        }
        rows += (size_t)ch.n * T;
        part.chunks.push_back(ch);
    }
    part.rows = rows - row_base;
    part.cols = cols;
}

// enqueue a part: descriptor upload, crops (stage A = gather / warp + cv2 resize when asked, stage B = AlignCollate into the wide image),
// conv stack, pooled rows into seq_v.  Nothing here waits for the device (the buffers it needs are sized by the caller).
static void rec_launch_part(bbocr_ctx* c, const uint8_t* gray, int H, int W, const RecPart& part, DevBuf& desc_buf, bool stage_a) {
    if (part.descs.empty()) return;
    const size_t bytes = part.descs.size() * sizeof(CropDesc);
    desc_buf.ensure(bytes);
    PinBuf& pin = (&desc_buf == &c->crop_desc2) ? c->desc_pin2 : c->desc_pin;
    pin.ensure(bytes);
    memcpy(pin.p, part.descs.data(), bytes);
    const CropDesc* dd = (const CropDesc*)desc_buf.p;
    const int n = (int)part.descs.size(), Wt = (int)part.cols;
    HIPCHK(hipMemcpyAsync(desc_buf.p, pin.p, bytes, hipMemcpyHostToDevice, c->stream));
    if (stage_a)
        HIPCHK(launch_crops(gray, H, W, dd, 0, n, 0, part.any_warp, part.any_tall, (uint8_t*)c->crop_wscratch.p, (uint8_t*)c->crop_scratch.p,
                            (uint8_t*)c->crop_hscratch.p, (const uint8_t*)c->crop_luts.p, nullptr, 1, c->stream));
    for (int pass = 0; pass < 2; ++pass) {
        c->arena.begin(pass == 0);
        uint16_t* wide = c->arena.alloc<uint16_t>((size_t)64 * Wt);
        if (pass == 1)
            HIPCHK(launch_crops(gray, H, W, dd, 0, n, 0, part.any_warp, part.any_tall, (uint8_t*)c->crop_wscratch.p, (uint8_t*)c->crop_scratch.p,
                                (uint8_t*)c->crop_hscratch.p, (const uint8_t*)c->crop_luts.p, wide, 2, c->stream, Wt, REC_GAP));
        crnn_features_wide(c, wide, Wt, dd, 0, n, (uint16_t*)c->seq_v.p);
        if (pass == 0) c->arena.buf.ensure(c->arena.off);
    }
}

static void rec_add_tables(RecRun& run, const RecPart& part) {
    for (const RecChunk& ch : part.chunks) {
        for (int s0 = 0; s0 < ch.n; s0 += 16) {
            run.tiles.push_back((int)(ch.row0 + (size_t)s0 * ch.T));
            run.tiles.push_back(std::min(16, ch.n - s0));
            run.tiles.push_back(ch.T);
            run.tiles.push_back(0);
        }
        for (int i = 0; i < ch.n; ++i) {
            run.seqs.push_back((int)(ch.row0 + (size_t)i * ch.T));
            run.seqs.push_back(ch.T);
            run.seq_k.push_back(part.order[ch.first + i]);
        }
    }
    run.rows += part.rows;
}

// sequence stage + CTC over every row the parts of `run` produced; texts / confs are indexed by result position
static void rec_finish(bbocr_ctx* c, RecRun& run, std::vector<std::vector<int>>& texts, std::vector<double>& confs) {
    const size_t rows = run.rows;
    if (rows == 0) return;
    const size_t rows_pad = align_up(rows, 256);
    auto t0 = clk::now();
    {   // longest sequences first: the launch ends with the shortest tails
        std::vector<int>& tiles = run.tiles;
        const size_t nt = tiles.size() / 4;
        std::vector<size_t> perm(nt);
        for (size_t i = 0; i < nt; ++i) perm[i] = i;
        std::stable_sort(perm.begin(), perm.end(), [&](size_t x, size_t y) { return tiles[x * 4 + 2] > tiles[y * 4 + 2]; });
        std::vector<int> t2(tiles.size());
        for (size_t i = 0; i < nt; ++i) memcpy(&t2[i * 4], &tiles[perm[i] * 4], 16);
        tiles.swap(t2);
    }
    const std::vector<int>&tiles = run.tiles, &seqs = run.seqs;
    const int ntiles = (int)(tiles.size() / 4), nseq = (int)(seqs.size() / 2);
    c->seq_logits.ensure(rows_pad * 112 * 4);
    c->seq_tables.ensure((tiles.size() + seqs.size()) * 4);
    int* tiles_dev = (int*)c->seq_tables.p;
    int* seqs_dev = tiles_dev + tiles.size();
    HIPCHK(hipMemcpyAsync(tiles_dev, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(seqs_dev, seqs.data(), seqs.size() * 4, hipMemcpyHostToDevice, c->stream));
    crnn_sequence(c, rows_pad, tiles_dev, ntiles, (float*)c->seq_logits.p);
    HIPCHK(hipStreamSynchronize(c->stream));
    c->times[4] += (float)ms_since(t0);
    t0 = clk::now();
    c->ctc_idx.ensure(rows * 4);
    c->ctc_pmax.ensure(rows * 4);
    c->ctc_out_idx.ensure(rows * 4);
    c->ctc_out.ensure((size_t)nseq * sizeof(CtcOut));
    const bool beam = c->beam_width > 0;
    if (beam) c->ctc_probs.ensure(rows * 112 * sizeof(float));
    HIPCHK(launch_ctc((const float*)c->seq_logits.p, rows, 97, 112, seqs_dev, nseq, (int*)c->ctc_idx.p, (float*)c->ctc_pmax.p,
                      (int*)c->ctc_out_idx.p, (CtcOut*)c->ctc_out.p, c->stream, c->ignore_mask, beam ? (float*)c->ctc_probs.p : nullptr));
    const size_t oo_off = align_up(rows * 4, 16);
    c->ctc_pin.ensure(oo_off + (size_t)nseq * sizeof(CtcOut));
    const int* oidx = (const int*)c->ctc_pin.p;
    const CtcOut* oo = (const CtcOut*)((const char*)c->ctc_pin.p + oo_off);
    std::vector<float> probs(beam ? rows * 112 : 0);
    std::vector<std::vector<int>> beam_texts;
    HIPCHK(hipMemcpyAsync(c->ctc_pin.p, c->ctc_out_idx.p, rows * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync((char*)c->ctc_pin.p + oo_off, c->ctc_out.p, (size_t)nseq * sizeof(CtcOut), hipMemcpyDeviceToHost, c->stream));
    if (beam) HIPCHK(hipMemcpyAsync(probs.data(), c->ctc_probs.p, probs.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (beam) ctc_beam_search_batch(probs.data(), seqs.data(), nseq, 97, 112, c->beam_width, beam_texts);   // the confidence stays the greedy path's
    for (int i = 0; i < nseq; ++i) {
        const int k = run.seq_k[i];
        const size_t r0 = (size_t)seqs[2 * i];
        if (beam) texts[k] = beam_texts[i];
        else texts[k].assign(oidx + r0, oidx + r0 + oo[i].len);
        // custom_mean: prod ** (2 / sqrt(len)); an all-blank sequence scores np.array([0])
        confs[k] = oo[i].cnt > 0 ? std::pow((double)oo[i].prod, 2.0 / std::sqrt((double)oo[i].cnt)) : 0.0;
    }
    c->times[5] += (float)ms_since(t0);
}

// split `sel` into runs whose pooled rows fit one sequence pass (in width order, like the wide image)
static std::vector<std::vector<int>> rec_split_runs(const bbocr_ctx* c, const std::vector<BoxJob>& jobs, const std::vector<int>& sel) {
    std::vector<int> byw(sel.size());
    for (size_t k = 0; k < sel.size(); ++k) byw[k] = (int)k;
    std::stable_sort(byw.begin(), byw.end(), [&](int x, int y) { return jobs[sel[x]].d.imgW < jobs[sel[y]].d.imgW; });
    std::vector<std::vector<int>> runs(1);
    size_t rows = 0;
    for (int k : byw) {
        const size_t t = (size_t)(jobs[sel[k]].d.imgW / 4 - 1);
        if (!runs.back().empty() && rows + t > rec_max_rows(c)) { runs.emplace_back(); rows = 0; }
        runs.back().push_back(k);
        rows += t;
    }
    return runs;
}

// run one recognition pass over `sel` (indices into jobs); descs must already carry lut_off for a contrast pass.
static void recognise_pass(bbocr_ctx* c, const uint8_t* gray, int H, int W, std::vector<BoxJob>& jobs, const std::vector<int>& sel,
                           bool stage_a, std::vector<std::vector<int>>& texts, std::vector<double>& confs) {
    texts.assign(sel.size(), {});
    confs.assign(sel.size(), 0.0);
    if (sel.empty()) return;
    for (const std::vector<int>& ks : rec_split_runs(c, jobs, sel)) {
        // positions inside sel -> a sub-selection whose result positions are the positions in sel
        std::vector<int> sub(ks.size());
        for (size_t i = 0; i < ks.size(); ++i) sub[i] = sel[ks[i]];
        RecPart part;
        rec_plan_part(jobs, sub, 0, 0, part);
        for (size_t i = 0; i < part.order.size(); ++i) part.order[i] = ks[part.order[i]];
        auto t0 = clk::now();
        c->seq_v.ensure(align_up(part.rows, 256) * 256 * 2);
        rec_launch_part(c, gray, H, W, part, c->crop_desc, stage_a);
        c->times[3] += (float)ms_since(t0);
        RecRun run;
        rec_add_tables(run, part);
        rec_finish(c, run, texts, confs);
    }
}

// np.percentile(img, q) (method 'linear') from a 256-bin histogram of n uint8 samples
static double percentile_u8(const unsigned int* hist, size_t n, double q) {
    const double virt = (double)(n - 1) * (q / 100.0);
    const double prev = std::floor(virt);
    const double gamma = virt - prev;
    const size_t i0 = (size_t)prev, i1 = std::min(i0 + 1, n - 1);
    auto at = [&](size_t idx) {
        size_t acc = 0;
        for (int v = 0; v < 256; ++v) {
            acc += hist[v];
            if (idx < acc) return v;
        }
        return 255;
    };
    const int a = at(i0), b = at(i1);
    const double diff = (double)(b - a);
    double r = (double)a + diff * gamma;
    if (gamma >= 0.5) r = (double)b - diff * (1 - gamma);
    return r;
}

// State of a recognition whose FIRST feature part (the crops of pages [0, pages)) was enqueued before the boxes of the remaining pages
// existed (readtext_batch: while the last detector pass's CCL + host geometry run).  recognize_impl picks it up and adds the rest.
struct RecEarly {
    bool active = false;
    int pages = 0;
    std::vector<BoxJob> jobs;
    std::vector<int> box_off;        // [pages + 1]
    size_t a_total = 0, w_total = 0; // crop scratch consumed by those jobs
    RecPart part;
};

static void rec_check_params(bbocr_ctx* c, const bbocr_params& p) {
    if (!c->crnn_loaded) fail(BBOCR_ERR_STATE, "recogniser weights not loaded");
    if (p.ignore_mask[0] & 1u) fail(BBOCR_ERR_ARG, "the CTC blank (class 0) cannot be ignored");
    for (int i = 0; i < 4; ++i) c->ignore_mask[i] = p.ignore_mask[i];
    if (p.decoder != BBOCR_DECODER_GREEDY && p.decoder != BBOCR_DECODER_BEAMSEARCH) fail(BBOCR_ERR_ARG, "unknown decoder");
    if (p.decoder == BBOCR_DECODER_BEAMSEARCH && p.beam_width <= 0) fail(BBOCR_ERR_ARG, "beam_width must be positive");
    c->beam_width = p.decoder == BBOCR_DECODER_BEAMSEARCH ? p.beam_width : 0;
}

// Reader.recognize's per-box branch: horizontal boxes first, then free boxes, page by page
static void rec_plan_pages(const HostBoxes& hb, int b0, int b1, int H, int W, std::vector<BoxJob>& jobs, std::vector<int>& box_off) {
    for (int b = b0; b < b1; ++b) {
        for (const auto& hbx : hb.hori[b]) {
            BoxJob j;
            if (plan_horizontal(hbx, b, H, W, j)) jobs.push_back(j);
        }
        for (const auto& fq : hb.freeb[b]) {
            BoxJob j;
            if (plan_free(fq, b, j)) jobs.push_back(j);
        }
        box_off[b + 1] = (int)jobs.size();
    }
}

// crop scratch offsets of jobs [first, end), continuing at a_total / w_total
static void rec_layout_scratch(std::vector<BoxJob>& jobs, size_t first, size_t& a_total, size_t& w_total) {
    for (size_t i = first; i < jobs.size(); ++i) {
        BoxJob& j = jobs[i];
        j.d.a_off = (int)a_total;
        a_total += align_up((size_t)j.d.rw * j.d.rh, 16);
        if (j.d.warp) {
            j.d.warp_off = (int)w_total;
            w_total += align_up((size_t)j.d.sw * j.d.sh, 16);
        }
        if (a_total > 0x7fffffff || w_total > 0x7fffffff) fail(BBOCR_ERR_OVERFLOW, "crop scratch exceeds 2 GiB");
    }
}

// enqueue the feature part of pages [0, pages) of a B-page batch; the buffers that must survive until the rest arrives (stage-A
// crops for the contrast retry, pooled rows) are sized for the whole batch by extrapolation
static void rec_early_begin(bbocr_ctx* c, const uint8_t* gray, int pages, int B, int H, int W, const HostBoxes& hb, const bbocr_params& p,
                            RecEarly& e) {
    rec_check_params(c, p);
    e.box_off.assign(pages + 1, 0);
    rec_plan_pages(hb, 0, pages, H, W, e.jobs, e.box_off);
    if (e.jobs.empty()) return;
    rec_layout_scratch(e.jobs, 0, e.a_total, e.w_total);
    std::vector<int> all(e.jobs.size());
    for (size_t i = 0; i < all.size(); ++i) all[i] = (int)i;
    rec_plan_part(e.jobs, all, 0, 0, e.part);
    const double grow = 1.25 * (double)B / (double)pages;
    if ((double)e.part.rows * grow > (double)rec_max_rows(c)) { e = RecEarly(); return; }   // would not fit one sequence pass: no early part
    c->crop_scratch.ensure(std::max<size_t>((size_t)((double)e.a_total * grow), 16));
    c->crop_hscratch.ensure(std::max<size_t>((size_t)((double)e.a_total * grow), 16));
    c->crop_wscratch.ensure(std::max<size_t>((size_t)((double)e.w_total * grow), 16));
    c->crop_luts.ensure(256);
    c->seq_v.ensure(align_up((size_t)((double)e.part.rows * grow), 256) * 256 * 2);
    auto t0 = clk::now();
    rec_launch_part(c, gray, H, W, e.part, c->crop_desc, true);
    c->times[3] += (float)ms_since(t0);
    e.pages = pages;
    e.active = true;
}

static void recognize_impl(bbocr_ctx* c, const uint8_t* gray, int B, int H, int W, const HostBoxes& hb, const bbocr_params& p,
                           std::vector<BoxJob>& jobs, std::vector<int>& box_off, RecEarly* early = nullptr) {
    rec_check_params(c, p);
    jobs.clear();
    box_off.assign(B + 1, 0);
    // rotation_info: Reader.recognize then takes its batched branch -- get_image_list over the whole page (free boxes first, result sorted
    // by the top y of the first corner), ONE padded width for every crop of the page (max_width), the list extended by np.rot90 copies of
    // every crop per angle (make_rotated_img_list), and per box the most confident variant kept (set_result_with_confidence)
    int angles[4] = {0, 0, 0, 0}, nrot = 0;
    for (int i = 0; i < 4 && p.rotation_info[i] != 0; ++i) {
        const int a = p.rotation_info[i];
        if (a != 90 && a != 180 && a != 270) fail(BBOCR_ERR_ARG, "rotation_info angles must be 90, 180 or 270");
        angles[nrot++] = a;
    }
    const bool resume = early && early->active && nrot == 0;
    size_t n_early = 0;
    if (resume) {                                  // pages [0, early->pages) are planned and their feature part is on the device
        jobs = std::move(early->jobs);
        n_early = jobs.size();
        for (int b = 0; b <= early->pages; ++b) box_off[b] = early->box_off[b];
        rec_plan_pages(hb, early->pages, B, H, W, jobs, box_off);
    }
    for (int b = 0; b < B && !resume; ++b) {
        if (nrot == 0) {
            rec_plan_pages(hb, b, b + 1, H, W, jobs, box_off);
            continue;
        } else {
            std::vector<BoxJob> page;
            for (const auto& fq : hb.freeb[b]) {
                BoxJob j;
                if (plan_free(fq, b, j)) page.push_back(j);
            }
            for (const auto& hbx : hb.hori[b]) {
                BoxJob j;
                if (plan_horizontal(hbx, b, H, W, j)) page.push_back(j);
            }
            std::stable_sort(page.begin(), page.end(), [](const BoxJob& x, const BoxJob& y) { return x.quad[1] < y.quad[1]; });
            int page_w = 64;                      // max(max_width, imgH); max_width = ceil(max ratio) * 64 = the widest own bucket
            for (const BoxJob& j : page) page_w = std::max(page_w, j.d.imgW);
            for (BoxJob& j : page) {
                j.d.imgW = page_w;
                const int cw = (int)std::ceil(64 * ((double)j.d.rw / (double)j.d.rh));
                j.d.fw = cw > page_w ? page_w : cw;
                jobs.push_back(j);
            }
        }
        box_off[b + 1] = (int)jobs.size();
    }
    const size_t n_base = jobs.size();
    for (int r = 0; r < nrot; ++r)
        for (size_t i = 0; i < n_base; ++i) {
            BoxJob j = jobs[i];
            j.d.rot = angles[r] / 90;
            if (j.d.rot & 1) std::swap(j.d.rw, j.d.rh);
            const int cw = (int)std::ceil(64 * ((double)j.d.rw / (double)j.d.rh));       // AlignCollate on the rotated image
            j.d.fw = cw > j.d.imgW ? j.d.imgW : cw;
            jobs.push_back(j);
        }
    // the variants only live inside this function: whatever path returns, the caller sees one job per box
    struct Collapse {
        std::vector<BoxJob>& jobs; size_t n_base; int nrot;
        ~Collapse() {
            if (nrot == 0 || jobs.size() != n_base * (size_t)(nrot + 1)) return;
            for (size_t i = 0; i < n_base; ++i) {
                size_t best = i;
                for (int r = 1; r <= nrot; ++r)
                    if (jobs[(size_t)r * n_base + i].conf > jobs[best].conf) best = (size_t)r * n_base + i;     // first maximum wins
                if (best != i) { jobs[i].text = jobs[best].text; jobs[i].conf = jobs[best].conf; }
            }
            jobs.resize(n_base);
        }
    } collapse{jobs, n_base, nrot};
    if (jobs.empty()) return;
    std::vector<std::vector<int>> texts;
    std::vector<double> confs;
    if (resume) {
        size_t a_total = early->a_total, w_total = early->w_total;
        rec_layout_scratch(jobs, n_early, a_total, w_total);
        c->crop_scratch.ensure_keep(std::max<size_t>(a_total, 16), early->a_total);   // part 1's stage-A crops feed the contrast retry
        c->crop_hscratch.ensure_keep(std::max<size_t>(a_total, 16), 0);
        c->crop_wscratch.ensure_keep(std::max<size_t>(w_total, 16), 0);
        texts.assign(jobs.size(), {});
        confs.assign(jobs.size(), 0.0);
        std::vector<int> rest(jobs.size() - n_early);
        for (size_t i = 0; i < rest.size(); ++i) rest[i] = (int)(n_early + i);
        RecPart part2;
        rec_plan_part(jobs, rest, (int)n_early, early->part.rows, part2);
        RecRun run;
        rec_add_tables(run, early->part);
        if (early->part.rows + part2.rows <= rec_max_rows(c)) {
            c->seq_v.ensure_keep(align_up(early->part.rows + part2.rows, 256) * 256 * 2, early->part.rows * 256 * 2);
            auto t0 = clk::now();
            rec_launch_part(c, gray, H, W, part2, c->crop_desc2, true);
            c->times[3] += (float)ms_since(t0);
            rec_add_tables(run, part2);
            rec_finish(c, run, texts, confs);
        } else {                                   // the rest does not fit the same sequence pass: finish part 1, then the rest on its own
            rec_finish(c, run, texts, confs);
            std::vector<std::vector<int>> t2;
            std::vector<double> c2;
            recognise_pass(c, gray, H, W, jobs, rest, true, t2, c2);
            for (size_t i = 0; i < rest.size(); ++i) { texts[rest[i]] = t2[i]; confs[rest[i]] = c2[i]; }
        }
        early->active = false;
    } else {
        size_t a_total = 0, w_total = 0;
        rec_layout_scratch(jobs, 0, a_total, w_total);
        c->crop_scratch.ensure(std::max<size_t>(a_total, 16));
        c->crop_hscratch.ensure(std::max<size_t>(a_total, 16));
        c->crop_wscratch.ensure(std::max<size_t>(w_total, 16));
        c->crop_luts.ensure(256);
        std::vector<int> all(jobs.size());
        for (size_t i = 0; i < jobs.size(); ++i) all[i] = (int)i;
        recognise_pass(c, gray, H, W, jobs, all, true, texts, confs);
    }
    for (size_t i = 0; i < jobs.size(); ++i) { jobs[i].text = texts[i]; jobs[i].conf = confs[i]; }
    // second round: adjust_contrast_grey for low-confidence boxes
    std::vector<int> low;
    for (size_t i = 0; i < jobs.size(); ++i)
        if (jobs[i].conf < p.contrast_ths) low.push_back((int)i);
    if (low.empty() || !(p.adjust_contrast > 0)) return;
    auto t0 = clk::now();
    std::vector<CropDesc> ld(low.size());
    for (size_t k = 0; k < low.size(); ++k) ld[k] = jobs[low[k]].d;
    c->crop_desc.ensure(ld.size() * sizeof(CropDesc));
    c->crop_hist.ensure(ld.size() * 256 * 4);
    HIPCHK(hipMemcpyAsync(c->crop_desc.p, ld.data(), ld.size() * sizeof(CropDesc), hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_crop_hist((const uint8_t*)c->crop_scratch.p, (const CropDesc*)c->crop_desc.p, 0, (int)ld.size(), (unsigned int*)c->crop_hist.p,
                            c->stream));
    std::vector<unsigned int> hist(ld.size() * 256);
    HIPCHK(hipMemcpyAsync(hist.data(), c->crop_hist.p, hist.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // adjust_contrast_grey leaves a crop whose contrast is already >= the target untouched: its second prediction would be computed
    // from bit-identical input, equals the first one, and get_text's `pred1[1] > pred2[1] ? pred1 : pred2` picks the same pair either
    // way.  Only the crops that really change are run again.
    std::vector<int> redo;
    std::vector<uint8_t> luts;
    for (size_t k = 0; k < low.size(); ++k) {
        const size_t npx = (size_t)ld[k].rw * ld[k].rh;
        const double high = percentile_u8(&hist[k * 256], npx, 90.0), lowp = percentile_u8(&hist[k * 256], npx, 10.0);
        const double contrast = (high - lowp) / std::max(10.0, high + lowp);
        if (!(contrast < p.adjust_contrast)) continue;
        const double ratio = 200.0 / std::max(10.0, high - lowp);
        jobs[low[k]].d.lut_off = (int)luts.size();
        for (int v = 0; v < 256; ++v) {
            double x = ((double)v - lowp + 25) * ratio;
            x = std::max(0.0, std::min(255.0, x));
            luts.push_back((uint8_t)x);
        }
        redo.push_back(low[k]);
    }
    if (!redo.empty()) {
        c->crop_luts.ensure(luts.size());
        HIPCHK(hipMemcpyAsync(c->crop_luts.p, luts.data(), luts.size(), hipMemcpyHostToDevice, c->stream));   // `luts` outlives the pass below, which ends synchronised
        std::vector<std::vector<int>> t2;
        std::vector<double> c2;
        recognise_pass(c, gray, H, W, jobs, redo, false, t2, c2);
        for (size_t k = 0; k < redo.size(); ++k) {
            BoxJob& j = jobs[redo[k]];
            j.d.lut_off = -1;
            if (!(j.conf > c2[k])) { j.text = t2[k]; j.conf = c2[k]; }
        }
    }
    c->times[6] += (float)ms_since(t0);
}

static bbocr_result* export_result(int B, const std::vector<BoxJob>& jobs, const std::vector<int>& box_off) {
    bbocr_result* r = (bbocr_result*)calloc(1, sizeof(bbocr_result));
    const size_t nb = jobs.size();
    r->n_images = B;
    r->box_off = (int*)calloc(B + 1, sizeof(int));
    for (int b = 0; b <= B; ++b) r->box_off[b] = box_off[b];
    r->quads = (double*)calloc(std::max<size_t>(1, nb) * 8, sizeof(double));
    r->is_free = (int*)calloc(std::max<size_t>(1, nb), sizeof(int));
    r->text_off = (int*)calloc(nb + 1, sizeof(int));
    r->conf = (double*)calloc(std::max<size_t>(1, nb), sizeof(double));
    size_t nt = 0;
    for (const BoxJob& j : jobs) nt += j.text.size();
    r->text_idx = (int*)calloc(std::max<size_t>(1, nt), sizeof(int));
    size_t o = 0;
    for (size_t i = 0; i < nb; ++i) {
        memcpy(r->quads + i * 8, jobs[i].quad, 64);
        r->is_free[i] = jobs[i].is_free;
        r->conf[i] = jobs[i].conf;
        r->text_off[i] = (int)o;
        for (int v : jobs[i].text) r->text_idx[o++] = v;
    }
    r->text_off[nb] = (int)o;
    return r;
}

// ------------------------------------------------------------------------------------------------ pre-processing chain (f2)
// Host-side constants of the chain, computed exactly like oracle/preprocess.py (float32 where the C sources use float).
struct CubicAxis { std::vector<int> first; std::vector<short> coef; };
static CubicAxis cubic_axis(int dst, int src) {   // imgproc resize.cpp: fx in float, cvFloor, interpolateCubic (A = -0.75), cvRound(c * 2048)
    CubicAxis a;
    a.first.resize(dst);
    a.coef.resize((size_t)dst * 4);
    const double scale = (double)src / (double)dst;
    const float A = -0.75f;
    for (int d = 0; d < dst; ++d) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        const int s = (int)std::floor(f);
        f = f - (float)s;
        float c[4];
        const float x = f;
        c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
        c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
        c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
        c[3] = 1.f - c[0] - c[1] - c[2];
        a.first[d] = s - 1;
        for (int k = 0; k < 4; ++k) {
            long v = std::lrint((double)c[k] * 2048.0);        // cvRound: half to even
            a.coef[(size_t)d * 4 + k] = (short)std::max<long>(-32768, std::min<long>(32767, v));
        }
    }
    return a;
}
static void gaussian_taps3(double sigma, int k[3]) {   // getGaussianKernelBitExact -> 8.8 fixed point, error diffusion (sum 256)
    double v[3], tot = 0;
    for (int i = 0; i < 3; ++i) { v[i] = std::exp(-((double)(i - 1) * (i - 1)) / (2.0 * sigma * sigma)); tot += v[i]; }
    double err = 0;
    for (int i = 0; i < 3; ++i) {
        const double t = v[i] / tot * 256.0;
        const int r = (int)std::floor(t + err + 0.5);
        err += t - r;
        k[i] = r;
    }
}
static void pil_blend_lut(int in1, float alpha, uint8_t lut[256]) {   // libImaging/Blend.c with a constant first image
    for (int v = 0; v < 256; ++v) {
        const float t = (float)in1 + alpha * (float)(v - in1);
        if (alpha >= 0.f && alpha <= 1.f) lut[v] = (uint8_t)t;
        else lut[v] = t <= 0.f ? 0 : (t >= 255.f ? 255 : (uint8_t)t);
    }
}
static float pil_box_radius(float radius, int passes) {   // libImaging/BoxBlur.c::_gaussian_blur_radius
    const float sigma2 = radius * radius / (float)passes;
    const float L = (float)std::sqrt(12.0 * (double)sigma2 + 1.0);
    const float l = (float)std::floor(((double)L - 1.0) / 2.0);
    float a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a = a / (6 * (sigma2 - (l + 1) * (l + 1)));
    return l + a;
}

static void pp_resize(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw) {
    const CubicAxis ax = cubic_axis(dw, W), ay = cubic_axis(dh, H);
    const size_t bytes = (size_t)dw * 4 + (size_t)dw * 8 + (size_t)dh * 4 + (size_t)dh * 8;
    c->pp_tab.ensure(bytes);
    unsigned char* t = (unsigned char*)c->pp_tab.p;
    int* x0 = (int*)t;                         t += (size_t)dw * 4;
    int* y0 = (int*)t;                         t += (size_t)dh * 4;
    short* cx = (short*)t;                     t += (size_t)dw * 8;
    short* cy = (short*)t;
    HIPCHK(hipMemcpyAsync(x0, ax.first.data(), (size_t)dw * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(y0, ay.first.data(), (size_t)dh * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(cx, ax.coef.data(), (size_t)dw * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(cy, ay.coef.data(), (size_t)dh * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_pp_resize_cubic(src, H, W, dst, dh, dw, x0, cx, y0, cy, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // the host tables must outlive the copies
}
// GaussianBlur 3x3; returns the sum of the output pixels (for the following Contrast step)
static unsigned long long pp_gauss(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, double sigma) {
    int k[3];
    gaussian_taps3(sigma, k);
    c->pp_tab.ensure(64);
    HIPCHK(hipMemsetAsync(c->pp_tab.p, 0, 8, c->stream));
    HIPCHK(launch_pp_gauss3(src, H, W, dst, k[0], k[1], k[2], (unsigned long long*)c->pp_tab.p, c->stream));
    unsigned long long sum = 0;
    HIPCHK(hipMemcpyAsync(&sum, c->pp_tab.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return sum;
}
// CLAHE of lut[src] (lut = pointwise steps folded in front of it; identity if null)
static void pp_clahe(bbocr_ctx* c, const uint8_t* src, int H, int W, const uint8_t* lut_host, uint8_t* dst, double clip_limit) {
    const int tx = 8, ty = 8;
    // clahe.cpp pads BOTH axes by tiles - (size % tiles) as soon as ONE of them is not a multiple of the grid -- a whole extra
    // 8 rows / columns on the axis that did divide (upstream quirk, restated as is)
    const bool pad = (H % ty) || (W % tx);
    const int EH = pad ? H + (ty - H % ty) : H, EW = pad ? W + (tx - W % tx) : W;
    if (EH - H >= H || EW - W >= W) fail(BBOCR_ERR_ARG, "image smaller than the CLAHE tile grid");
    const int th = EH / ty, tw = EW / tx;
    c->pp_tab.ensure(256 + (size_t)tx * ty * 256 * 4 + (size_t)tx * ty * 256);
    uint8_t* d_lut = (uint8_t*)c->pp_tab.p;
    unsigned int* d_hist = (unsigned int*)((unsigned char*)c->pp_tab.p + 256);
    uint8_t* d_tl = (uint8_t*)c->pp_tab.p + 256 + (size_t)tx * ty * 256 * 4;
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
    HIPCHK(hipMemcpyAsync(d_lut, lut_host ? lut_host : ident, 256, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)tx * ty * 256 * 4, c->stream));
    HIPCHK(launch_pp_clahe_hist(src, H, W, d_lut, tw, th, tx, ty, d_hist, c->stream));
    std::vector<unsigned int> hist((size_t)tx * ty * 256);
    HIPCHK(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // imgproc clahe.cpp: clip + redistribute, LUT = cvRound(cumsum * 255 / tile_area) in float
    const int area = th * tw;
    const float lut_scale = 255.0f / (float)area;
    int clip = 0;
    if (clip_limit > 0) clip = std::max((int)(clip_limit * area / 256), 1);
    std::vector<uint8_t> tl((size_t)tx * ty * 256);
    for (int t = 0; t < tx * ty; ++t) {
        long long h[256];
        for (int i = 0; i < 256; ++i) h[i] = hist[(size_t)t * 256 + i];
        if (clip > 0) {
            long long clipped = 0;
            for (int i = 0; i < 256; ++i) if (h[i] > clip) { clipped += h[i] - clip; h[i] = clip; }
            const long long batch = clipped / 256;
            long long residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) h[i] += batch;
            if (residual) {
                const int step = std::max((int)(256 / residual), 1);
                for (int i = 0; i < 256 && residual > 0; i += step, --residual) h[i] += 1;
            }
        }
        long long sum = 0;
        for (int i = 0; i < 256; ++i) {
            sum += h[i];
            const long v = std::lrintf((float)sum * lut_scale);
            tl[(size_t)t * 256 + i] = (uint8_t)std::max<long>(0, std::min<long>(255, v));
        }
    }
    HIPCHK(hipMemcpyAsync(d_tl, tl.data(), tl.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_pp_clahe_apply(src, H, W, d_lut, d_tl, tw, th, tx, ty, dst, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
}
// PIL UnsharpMask on src -> dst; tmp1/tmp2: two scratch planes of the same size
static void pp_unsharp(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, uint8_t* tmp1, uint8_t* tmp2, float radius, int percent,
                       int threshold) {
    const float fr = pil_box_radius(radius, 3);
    const int r = (int)fr;
    const unsigned int ww = (unsigned int)((float)(1 << 24) / (fr * 2.f + 1.f));
    const unsigned int fw = ((1u << 24) - (unsigned int)(r * 2 + 1) * ww) / 2;
    const uint8_t* cur = src;
    uint8_t* bufs[2] = {tmp1, tmp2};
    int w = 0;
    for (int pass = 0; pass < 6; ++pass) {
        HIPCHK(launch_pp_box_pass(cur, bufs[w], H, W, pass >= 3, r, ww, fw, c->stream));
        cur = bufs[w];
        w ^= 1;
    }
    HIPCHK(launch_pp_unsharp(src, cur, dst, (size_t)H * W, percent, threshold, c->stream));
}

static void preprocess_book_cover_impl(bbocr_ctx* c, const uint8_t* bgr, int H, int W, uint8_t* out, int dh, int dw) {
    const size_t n = (size_t)dh * dw;
    c->pp_gray.ensure((size_t)H * W);
    c->pp_a.ensure(n);
    c->pp_b.ensure(n);
    c->pp_c.ensure(n);
    uint8_t *g = (uint8_t*)c->pp_gray.p, *a = (uint8_t*)c->pp_a.p, *b = (uint8_t*)c->pp_b.p, *cc = (uint8_t*)c->pp_c.p;
    HIPCHK(launch_gray(bgr, g, (size_t)H * W, c->stream));             // channels as given: B 1868, G 9617, R 4899 for cv2.imread's BGR
    pp_resize(c, g, H, W, a, dh, dw);
    const unsigned long long sum = pp_gauss(c, a, dh, dw, b, 3.0);
    // ImageEnhance.Contrast(1.9) then Brightness(1.2): two pointwise maps, folded into one LUT in front of CLAHE
    const int mean = (int)((double)sum / (double)n + 0.5);
    uint8_t l1[256], l2[256], lut[256];
    pil_blend_lut(mean, 1.9f, l1);
    pil_blend_lut(0, 1.2f, l2);
    for (int i = 0; i < 256; ++i) lut[i] = l2[l1[i]];
    pp_clahe(c, b, dh, dw, lut, a, 2.5);
    pp_unsharp(c, a, dh, dw, out, b, cc, 1.0f, 30, 3);
    HIPCHK(hipStreamSynchronize(c->stream));
}

template <typename F> static int guarded(bbocr_ctx* ctx, F&& f) {
    if (!ctx) return BBOCR_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        hipError_t e = hipSetDevice(ctx->cfg.device);
        if (e != hipSuccess) fail(BBOCR_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
        ctx->cur = ctx->stream;
        // the library runs on its own non-blocking streams: what the caller queued on the default stream (torch's) -- fills of
        // output buffers, input copies -- must be complete before our kernels touch the same memory
        e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) fail(BBOCR_ERR_HIP, std::string("default stream: ") + hipGetErrorString(e));
        f();
        return BBOCR_OK;
    } catch (const StatusError& se) {
        ctx->err = se.msg;
        (void)hipGetLastError();
        return se.code;
    } catch (const std::exception& ex) {
        ctx->err = ex.what();
        return BBOCR_ERR_INTERNAL;
    } catch (...) {
        ctx->err = "unknown failure";
        return BBOCR_ERR_INTERNAL;
    }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

void bbocr_default_params(bbocr_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->text_threshold = 0.7; p->low_text = 0.4; p->link_threshold = 0.4; p->canvas_size = 2560; p->mag_ratio = 1.0;
    p->slope_ths = 0.1; p->ycenter_ths = 0.5; p->height_ths = 0.5; p->width_ths = 0.5; p->add_margin = 0.1; p->min_size = 20;
    p->contrast_ths = 0.1; p->adjust_contrast = 0.5;
    p->decoder = BBOCR_DECODER_GREEDY; p->beam_width = 5;   /* rotation_info: zeros (memset) */
}

int bbocr_create(const bbocr_config* cfg, bbocr_ctx** out) {
    if (!out) return BBOCR_ERR_ARG;
    *out = nullptr;
    bbocr_ctx* c = nullptr;
    try {
        c = new bbocr_ctx();
        if (cfg) c->cfg = *cfg;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || c->cfg.device < 0 || c->cfg.device >= ndev) {
            delete c;
            return BBOCR_ERR_HIP;
        }
        if (hipSetDevice(c->cfg.device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&c->zero_page, 256) != hipSuccess || hipMemset(c->zero_page, 0, 256) != hipSuccess) {
            delete c;
            return BBOCR_ERR_HIP;
        }
    } catch (...) {
        delete c;
        return BBOCR_ERR_INTERNAL;
    }
    c->cur = c->stream;
    *out = c;
    return BBOCR_OK;
}

void bbocr_destroy(bbocr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    for (hipEvent_t e : c->sub_events) (void)hipEventDestroy(e);
    if (c->det_t0) { (void)hipEventDestroy(c->det_t0); (void)hipEventDestroy(c->det_t1); }
    if (c->ccl_t0) { (void)hipEventDestroy(c->ccl_t0); (void)hipEventDestroy(c->ccl_t1); }
    for (auto& r : c->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
    free_weights(c);
    DevBuf* bufs[] = {&c->arena.buf, &c->heat, &c->gray, &c->resized, &c->ccl_label, &c->ccl_stat, &c->ccl_slot, &c->ccl_comps, &c->ccl_rowext,
                      &c->ccl_counters, &c->crop_desc, &c->crop_scratch, &c->crop_hscratch, &c->crop_wscratch, &c->crop_luts, &c->crop_hist,
                      &c->ctc_idx, &c->ctc_pmax, &c->ctc_out_idx, &c->ctc_out, &c->seq_v, &c->seq_xp, &c->seq_h, &c->seq_lin, &c->seq_logits,
                      &c->seq_tables, &c->pp_gray, &c->pp_a, &c->pp_b, &c->pp_c, &c->pp_tab, &c->ctc_probs};
    for (DevBuf* b : bufs) b->release();
    if (c->zero_page) (void)hipFree(c->zero_page);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;
}

const char* bbocr_last_error(bbocr_ctx* c) { return c ? c->err.c_str() : "null context"; }

int bbocr_load_weights(bbocr_ctx* ctx, int which, const bbocr_tensor_desc* descs, int n) {
    return guarded(ctx, [&] {
        if (!descs || n <= 0 || (which != 0 && which != 1)) fail(BBOCR_ERR_ARG, "bad weight descriptor table");
        TensorMap tm(descs, n);
        if (which == 0) load_craft(ctx, tm);
        else load_crnn(ctx, tm);
    });
}

int bbocr_detect_dims(int H, int W, int canvas_size, double mag_ratio, int* H32, int* W32, int* rh, int* rw, double* ratio) {
    if (H <= 0 || W <= 0 || canvas_size <= 0) return BBOCR_ERR_ARG;
    const DetDims d = det_dims(H, W, canvas_size, (double)mag_ratio);
    if (H32) *H32 = d.H32;
    if (W32) *W32 = d.W32;
    if (rh) *rh = d.h;
    if (rw) *rw = d.w;
    if (ratio) *ratio = d.ratio;
    return BBOCR_OK;
}

int bbocr_detect(bbocr_ctx* ctx, const uint8_t* dev_rgb, int B, int H, int W, const bbocr_params* p, float* dev_heat_out) {
    return guarded(ctx, [&] {
        bbocr_params pp;
        bbocr_default_params(&pp);
        if (p) pp = *p;
        if (!dev_rgb || !dev_heat_out) fail(BBOCR_ERR_ARG, "null device pointer");
        memset(ctx->times, 0, sizeof(ctx->times));
        auto t0 = clk::now();
        detect_impl(ctx, dev_rgb, B, H, W, pp, dev_heat_out);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        prof_collect(ctx);
        ctx->times[0] = ctx->times[7] = (float)ms_since(t0);
    });
}

int bbocr_boxes(bbocr_ctx* ctx, const float* dev_heat, int B, int h, int w, double ratio, const bbocr_params* p, bbocr_boxlist** out) {
    return guarded(ctx, [&] {
        bbocr_params pp;
        bbocr_default_params(&pp);
        if (p) pp = *p;
        if (!dev_heat || !out) fail(BBOCR_ERR_ARG, "null pointer");
        memset(ctx->times, 0, sizeof(ctx->times));
        auto t0 = clk::now();
        HostBoxes hb;
        boxes_impl(ctx, dev_heat, B, h, w, ratio, pp, hb, ctx->stream);
        *out = export_boxes(hb);
        ctx->times[7] = (float)ms_since(t0);
    });
}

int bbocr_recognize(bbocr_ctx* ctx, const uint8_t* dev_gray, int B, int H, int W, const bbocr_boxlist* boxes, const bbocr_params* p,
                    bbocr_result** out) {
    return guarded(ctx, [&] {
        bbocr_params pp;
        bbocr_default_params(&pp);
        if (p) pp = *p;
        if (!dev_gray || !boxes || !out || boxes->n_images != B) fail(BBOCR_ERR_ARG, "bad recognise arguments");
        memset(ctx->times, 0, sizeof(ctx->times));
        auto t0 = clk::now();
        HostBoxes hb;
        import_boxes(boxes, hb);
        std::vector<BoxJob> jobs;
        std::vector<int> off;
        recognize_impl(ctx, dev_gray, B, H, W, hb, pp, jobs, off);
        prof_collect(ctx);
        *out = export_result(B, jobs, off);
        ctx->times[7] = (float)ms_since(t0);
    });
}

int bbocr_readtext_batch(bbocr_ctx* ctx, const uint8_t* dev_rgb, const uint8_t* dev_gray, int B, int H, int W, const bbocr_params* p,
                         bbocr_result** out) {
    return guarded(ctx, [&] {
        bbocr_params pp;
        bbocr_default_params(&pp);
        if (p) pp = *p;
        if (!dev_rgb || !out) fail(BBOCR_ERR_ARG, "null pointer");
        memset(ctx->times, 0, sizeof(ctx->times));
        auto t_all = clk::now();
        const DetDims d = det_dims(H, W, pp.canvas_size, pp.mag_ratio);
        ctx->heat.ensure((size_t)B * d.h * d.w * 2 * sizeof(float));
        if (!dev_gray) {
            ctx->gray.ensure((size_t)B * H * W);
            HIPCHK(launch_gray(dev_rgb, (uint8_t*)ctx->gray.p, (size_t)B * H * W, ctx->stream));
            dev_gray = (const uint8_t*)ctx->gray.p;
        }
        auto t0 = clk::now();
        // The whole detector is enqueued first (one event per sub-batch, no host wait); box extraction of sub-batch k then
        // runs on the second stream + host threads while sub-batch k+1 is still in the detector.
        std::vector<std::pair<int, int>> subs;
        if (!ctx->det_t0) { HIPCHK(hipEventCreate(&ctx->det_t0)); HIPCHK(hipEventCreate(&ctx->det_t1)); }
        HIPCHK(hipEventRecord(ctx->det_t0, ctx->stream));
        detect_impl(ctx, dev_rgb, B, H, W, pp, (float*)ctx->heat.p, [&](int b0, int nb) {
            if (subs.size() >= ctx->sub_events.size()) {
                hipEvent_t e;
                HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ctx->sub_events.push_back(e);
            }
            HIPCHK(hipEventRecord(ctx->sub_events[subs.size()], ctx->stream));
            subs.push_back({b0, nb});
        });
        HIPCHK(hipEventRecord(ctx->det_t1, ctx->stream));
        HostBoxes hb;
        hb.polys.resize(B); hb.hori.resize(B); hb.freeb.resize(B);
        RecEarly early;
        static const bool early_on = [] { const char* e = getenv("BBOCR_REC_EARLY"); return !(e && e[0] == '0'); }();   // A/B knob
        for (size_t k = 0; k < subs.size(); ++k) {
            const int b0 = subs[k].first, nb = subs[k].second;
            HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->sub_events[k], 0));
            HostBoxes part;
            boxes_impl(ctx, (const float*)ctx->heat.p + (size_t)b0 * d.h * d.w * 2, nb, d.h, d.w, d.ratio, pp, part, ctx->stream2);
            for (int i = 0; i < nb; ++i) {
                hb.polys[b0 + i] = std::move(part.polys[i]);
                hb.hori[b0 + i] = std::move(part.hori[i]);
                hb.freeb[b0 + i] = std::move(part.freeb[i]);
            }
            // every page but the last pass's has its boxes: their crops go through the recogniser's conv stack (queued behind the
            // detector on `stream`) while the last pass's CCL + host geometry run -- that stretch would otherwise leave the card idle
            if (early_on && subs.size() >= 2 && k + 2 == subs.size() && pp.rotation_info[0] == 0 && ctx->crnn_loaded)
                rec_early_begin(ctx, dev_gray, b0 + nb, B, H, W, hb, pp, early);
        }
        HIPCHK(hipEventSynchronize(ctx->det_t1));   // the detector's end, not the stream's: the early recogniser part may be running behind it
        float det_ms = 0.f;
        HIPCHK(hipEventElapsedTime(&det_ms, ctx->det_t0, ctx->det_t1));
        ctx->times[0] = det_ms;          // GPU span of the detector; box extraction (times[1], times[2]) overlaps it except for the last sub-batch
        (void)t0;
        std::vector<BoxJob> jobs;
        std::vector<int> off;
        recognize_impl(ctx, dev_gray, B, H, W, hb, pp, jobs, off, &early);
        prof_collect(ctx);
        *out = export_result(B, jobs, off);
        ctx->times[7] = (float)ms_since(t_all);
    });
}

int bbocr_preprocess_book_cover(bbocr_ctx* ctx, const uint8_t* dev_bgr, int H, int W, uint8_t* dev_out, int* out_h, int* out_w) {
    return guarded(ctx, [&] {
        if (H <= 0 || W <= 0) fail(BBOCR_ERR_ARG, "bad image shape");
        const int dh = (int)(H * 1.5), dw = (int)(W * 1.5);
        if (out_h) *out_h = dh;
        if (out_w) *out_w = dw;
        if (!dev_out) return;
        if (!dev_bgr) fail(BBOCR_ERR_ARG, "null device pointer");
        if (dh < 16 || dw < 16) fail(BBOCR_ERR_ARG, "image too small for the 8x8 CLAHE tile grid");
        preprocess_book_cover_impl(ctx, dev_bgr, H, W, dev_out, dh, dw);
    });
}

int bbocr_op_preprocess_stage(bbocr_ctx* ctx, int stage, const uint8_t* dev_src, int H, int W, uint8_t* dev_dst, int dh, int dw, double param) {
    return guarded(ctx, [&] {
        if (!dev_src || !dev_dst || H <= 0 || W <= 0 || dh <= 0 || dw <= 0) fail(BBOCR_ERR_ARG, "bad arguments");
        if (stage != 0 && (dh != H || dw != W)) fail(BBOCR_ERR_ARG, "only stage 0 changes the size");
        const size_t n = (size_t)H * W;
        if (stage == 0) {
            pp_resize(ctx, dev_src, H, W, dev_dst, dh, dw);
        } else if (stage == 1) {
            (void)pp_gauss(ctx, dev_src, H, W, dev_dst, param);
        } else if (stage == 2 || stage == 3) {
            // pointwise PIL enhancers: the chain folds them into CLAHE's input LUT; stand-alone they are one lookup pass
            uint8_t lut[256];
            if (stage == 2) {
                ctx->pp_a.ensure(n);
                ctx->pp_tab.ensure(512);
                HIPCHK(hipMemsetAsync(ctx->pp_tab.p, 0, 8, ctx->stream));
                // mean of the input: the 3x3 smoothing kernel with taps (0, 256, 0) is the identity and sums its output
                HIPCHK(launch_pp_gauss3(dev_src, H, W, (uint8_t*)ctx->pp_a.p, 0, 256, 0, (unsigned long long*)ctx->pp_tab.p, ctx->stream));
                unsigned long long sm = 0;
                HIPCHK(hipMemcpyAsync(&sm, ctx->pp_tab.p, 8, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                pil_blend_lut((int)((double)sm / (double)n + 0.5), (float)param, lut);
            } else {
                pil_blend_lut(0, (float)param, lut);
            }
            ctx->pp_tab.ensure(512);
            HIPCHK(hipMemcpyAsync((unsigned char*)ctx->pp_tab.p + 256, lut, 256, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(launch_pp_lut(dev_src, dev_dst, (const uint8_t*)ctx->pp_tab.p + 256, n, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
        } else if (stage == 4) {
            pp_clahe(ctx, dev_src, H, W, nullptr, dev_dst, param);
        } else if (stage == 5) {
            ctx->pp_b.ensure(n);
            ctx->pp_c.ensure(n);
            pp_unsharp(ctx, dev_src, H, W, dev_dst, (uint8_t*)ctx->pp_b.p, (uint8_t*)ctx->pp_c.p, (float)param, 30, 3);
            HIPCHK(hipStreamSynchronize(ctx->stream));
        } else {
            fail(BBOCR_ERR_ARG, "unknown pre-processing stage");
        }
    });
}

void bbocr_free_boxlist(bbocr_boxlist* b) {
    if (!b) return;
    free(b->poly_off); free(b->polys); free(b->hori_off); free(b->hori); free(b->free_off); free(b->free_q);
    free(b);
}

void bbocr_free_result(bbocr_result* r) {
    if (!r) return;
    free(r->box_off); free(r->quads); free(r->is_free); free(r->text_off); free(r->text_idx); free(r->conf);
    free(r);
}

int bbocr_set_profiling(bbocr_ctx* ctx, int on) {
    if (!ctx) return BBOCR_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    for (int g = 0; g < 2; ++g) { ctx->prof_ms[g] = 0; ctx->prof_flops[g] = 0; ctx->prof_launches[g] = 0; }
    return BBOCR_OK;
}

int bbocr_conv_profile(bbocr_ctx* ctx, int group, double* ms, double* flops, long long* launches) {
    if (!ctx || group < 0 || group > 1) return BBOCR_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ms) *ms = ctx->prof_ms[group];
    if (flops) *flops = ctx->prof_flops[group];
    if (launches) *launches = ctx->prof_launches[group];
    return BBOCR_OK;
}

int bbocr_stage_times(bbocr_ctx* ctx, float* ms, int n) {
    if (!ctx || !ms || n <= 0) return BBOCR_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int i = 0; i < n && i < 8; ++i) ms[i] = ctx->times[i];
    return BBOCR_OK;
}

// ---------------------------------------------------------------------------------------- host-only geometry (no GPU needed)
int bbocr_host_component_polys(const int* comps, const int* rowext, int n, int w, int h, double ratio, int* polys_out) {
    if (!comps || !rowext || !polys_out || n < 0 || w <= 0 || h <= 0 || !(ratio > 0)) return BBOCR_ERR_ARG;
    try {
        for (int i = 0; i < n; ++i) {
            const int* q = comps + (size_t)i * 7;
            bbocr::Component cc{q[0], q[1], q[2], q[3], q[4], q[5], q[6]};
            float box[4][2];
            bbocr::component_box(cc, rowext + (size_t)cc.row_off * 2, w, h, box);
            bbocr::box_to_poly(box, 1.0 / ratio, 1.0 / ratio, polys_out + (size_t)i * 8);
        }
    } catch (...) {
        return BBOCR_ERR_INTERNAL;
    }
    return BBOCR_OK;
}

int bbocr_host_group_boxes(const int* polys, int n, const bbocr_params* p, bbocr_boxlist** out) {
    if ((!polys && n > 0) || n < 0 || !out) return BBOCR_ERR_ARG;
    try {
        bbocr_params pp;
        bbocr_default_params(&pp);
        if (p) pp = *p;
        HostBoxes hb;
        hb.polys.assign(1, {});
        hb.hori.assign(1, {});
        hb.freeb.assign(1, {});
        for (int i = 0; i < n; ++i) {
            std::array<int, 8> a;
            memcpy(a.data(), polys + (size_t)i * 8, 32);
            hb.polys[0].push_back(a);
        }
        bbocr::GroupParams gp{pp.slope_ths, pp.ycenter_ths, pp.height_ths, pp.width_ths, pp.add_margin, pp.min_size};
        bbocr::group_text_box(hb.polys[0], gp, hb.hori[0], hb.freeb[0]);
        *out = export_boxes(hb);
    } catch (...) {
        return BBOCR_ERR_INTERNAL;
    }
    return BBOCR_OK;
}

int bbocr_host_ctc_beam(const float* probs, int n, int T, int C, int cs, int beam_width, int* text_off, int* text_idx) {
    if (!probs || !text_off || !text_idx || n <= 0 || T <= 0 || C <= 0 || C > cs || beam_width <= 0) return BBOCR_ERR_ARG;
    try {
        std::vector<int> seqs;
        for (int i = 0; i < n; ++i) { seqs.push_back(i * T); seqs.push_back(T); }
        std::vector<std::vector<int>> texts;
        ctc_beam_search_batch(probs, seqs.data(), n, C, cs, beam_width, texts);
        int o = 0;
        for (int i = 0; i < n; ++i) {
            text_off[i] = o;
            for (int v : texts[i]) text_idx[o++] = v;
        }
        text_off[n] = o;
    } catch (...) {
        return BBOCR_ERR_INTERNAL;
    }
    return BBOCR_OK;
}

// ---------------------------------------------------------------------------------------- single-operator entry points
int bbocr_op_conv2d(bbocr_ctx* ctx, const uint16_t* dev_in, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout, int KH,
                    int KW, int pad, int dil, int relu_in, int relu_out, int out_f32, void* dev_out, int pool_mode, int pool_relu,
                    uint16_t* dev_pool_out) {
    return guarded(ctx, [&] {
        if (!dev_in || !w || Cin <= 0 || (Cin & 31) || Cout <= 0 || KH <= 0 || KW <= 0) fail(BBOCR_ERR_ARG, "bad conv arguments");
        if (pool_mode < 0 || pool_mode > 2 || (pool_mode ? (!dev_pool_out || out_f32) : !dev_out)) fail(BBOCR_ERR_ARG, "bad conv output arguments");
        ConvPlan p = make_plan(Cin, Cout, KH, KW, pad, dil);
        std::vector<float> wv(w, w + (size_t)Cout * Cin * KH * KW), bv(Cout, 0.f);
        if (bias) std::copy(bias, bias + Cout, bv.begin());
        const size_t owned0 = ctx->owned.size();
        upload_plan(ctx, p, wv, bv);
        const int store = cdiv(Cout, 16) * 16;
        ConvArgs a{};
        a.in0 = dev_in; a.C0 = Cin; a.in0_cs = Cin;
        a.N = N; a.H = H; a.W = W;
        a.relu_in0 = relu_in != 0; a.relu_out = relu_out != 0; a.out_f32 = out_f32 != 0;
        a.out = dev_out; a.out_cs = store; a.cout_store = store;
        a.pool_mode = pool_mode; a.pool_relu = pool_relu != 0; a.store_full = (pool_mode && dev_out) ? 1 : 0; a.pool_cs = store; a.pool_out = dev_pool_out;
        a.zero = ctx->zero_page;
        const hipError_t e = launch_conv(p, a, ctx->stream);
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);
        while (ctx->owned.size() > owned0) { (void)hipFree(ctx->owned.back()); ctx->owned.pop_back(); }
        HIPCHK(e);
        HIPCHK(e2);
    });
}

int bbocr_crnn_logits(bbocr_ctx* ctx, const uint16_t* dev_crops, int n, int imgW, float* dev_logits) {
    return guarded(ctx, [&] {
        if (!ctx->crnn_loaded) fail(BBOCR_ERR_STATE, "recogniser weights not loaded");
        if (!dev_crops || !dev_logits || n <= 0 || imgW < 64 || (imgW & 63)) fail(BBOCR_ERR_ARG, "bad crop batch");
        const int T = imgW / 4 - 1;
        const size_t rows = (size_t)n * T, rows_pad = align_up(rows, 256);
        ctx->seq_v.ensure(rows_pad * 256 * 2);
        ctx->seq_logits.ensure(rows_pad * 112 * 4);
        ctx->arena.begin(true);
        crnn_features(ctx, dev_crops, n, imgW, nullptr);
        ctx->arena.buf.ensure(ctx->arena.off);
        ctx->arena.begin(false);
        crnn_features(ctx, dev_crops, n, imgW, (uint16_t*)ctx->seq_v.p);
        std::vector<int> tiles;
        for (int s0 = 0; s0 < n; s0 += 16) {
            tiles.push_back(s0 * T); tiles.push_back(std::min(16, n - s0)); tiles.push_back(T); tiles.push_back(0);
        }
        ctx->seq_tables.ensure(tiles.size() * 4);
        HIPCHK(hipMemcpyAsync(ctx->seq_tables.p, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        crnn_sequence(ctx, rows_pad, (const int*)ctx->seq_tables.p, (int)(tiles.size() / 4), (float*)ctx->seq_logits.p);
        HIPCHK(hipMemcpyAsync(dev_logits, ctx->seq_logits.p, rows * 112 * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    });
}

int bbocr_op_ctc(bbocr_ctx* ctx, const float* dev_logits, int n, int T, int C, int cs, int* text_off, int* text_idx, double* conf,
                 const unsigned int* ignore_mask, int beam_width) {
    return guarded(ctx, [&] {
        if (!dev_logits || !text_off || !text_idx || !conf || n <= 0 || T <= 0 || C <= 0 || C > cs) fail(BBOCR_ERR_ARG, "bad ctc arguments");
        const bool beam = beam_width > 0;
        const size_t rows = (size_t)n * T;
        ctx->ctc_idx.ensure(rows * 4);
        ctx->ctc_pmax.ensure(rows * 4);
        ctx->ctc_out_idx.ensure(rows * 4);
        ctx->ctc_out.ensure((size_t)n * sizeof(CtcOut));
        std::vector<int> seqs;
        for (int i = 0; i < n; ++i) { seqs.push_back(i * T); seqs.push_back(T); }
        ctx->seq_tables.ensure(seqs.size() * 4);
        HIPCHK(hipMemcpyAsync(ctx->seq_tables.p, seqs.data(), seqs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (beam) ctx->ctc_probs.ensure(rows * cs * sizeof(float));
        HIPCHK(launch_ctc(dev_logits, rows, C, cs, (const int*)ctx->seq_tables.p, n, (int*)ctx->ctc_idx.p, (float*)ctx->ctc_pmax.p,
                          (int*)ctx->ctc_out_idx.p, (CtcOut*)ctx->ctc_out.p, ctx->stream, ignore_mask, beam ? (float*)ctx->ctc_probs.p : nullptr));
        std::vector<int> oidx(rows);
        std::vector<CtcOut> oo(n);
        std::vector<float> probs(beam ? rows * cs : 0);
        std::vector<std::vector<int>> beam_texts;
        HIPCHK(hipMemcpyAsync(oidx.data(), ctx->ctc_out_idx.p, oidx.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(oo.data(), ctx->ctc_out.p, oo.size() * sizeof(CtcOut), hipMemcpyDeviceToHost, ctx->stream));
        if (beam) HIPCHK(hipMemcpyAsync(probs.data(), ctx->ctc_probs.p, probs.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (beam) ctc_beam_search_batch(probs.data(), seqs.data(), n, C, cs, beam_width, beam_texts);
        int o = 0;
        for (int i = 0; i < n; ++i) {
            text_off[i] = o;
            if (beam) for (int v : beam_texts[i]) text_idx[o++] = v;
            else for (int k = 0; k < oo[i].len; ++k) text_idx[o++] = oidx[(size_t)i * T + k];
            conf[i] = oo[i].cnt > 0 ? std::pow((double)oo[i].prod, 2.0 / std::sqrt((double)oo[i].cnt)) : 0.0;
        }
        text_off[n] = o;
    });
}

int bbocr_op_resize_u8(bbocr_ctx* ctx, const uint8_t* dev_src, int N, int sh, int sw, int C, uint8_t* dev_dst, int dh, int dw) {
    return guarded(ctx, [&] {
        if (!dev_src || !dev_dst || N <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || C <= 0) fail(BBOCR_ERR_ARG, "bad resize arguments");
        HIPCHK(launch_resize_u8(dev_src, N, sh, sw, C, dev_dst, dh, dw, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    });
}

int bbocr_op_crops(bbocr_ctx* ctx, const uint8_t* dev_gray, int H, int W, const int* hori, int n_hori, const double* free_q, int n_free, int imgW,
                   float contrast, uint16_t* dev_out, int* n_out, int mode) {
    return guarded(ctx, [&] {
        if (!dev_gray || !dev_out || !n_out || imgW < 64 || (imgW & 63) || mode < 0 || mode > 4) fail(BBOCR_ERR_ARG, "bad crop arguments");
        std::vector<BoxJob> jobs;
        auto take = [&](BoxJob& j) {
            if (mode == 0) {                       // per-box branch: the boxes whose own padded width is imgW
                if (j.d.imgW == imgW) jobs.push_back(j);
                return;
            }
            j.d.imgW = imgW;                       // batched branch (rotation_info): forced width, np.rot90(crop, mode - 1)
            j.d.rot = mode - 1;
            if (j.d.rot & 1) std::swap(j.d.rw, j.d.rh);
            const int cw = (int)std::ceil(64 * ((double)j.d.rw / (double)j.d.rh));
            j.d.fw = cw > imgW ? imgW : cw;
            jobs.push_back(j);
        };
        for (int i = 0; i < n_hori; ++i) {
            BoxJob j;
            std::array<int, 4> b;
            memcpy(b.data(), hori + (size_t)i * 4, 16);
            if (plan_horizontal(b, 0, H, W, j)) take(j);
        }
        for (int i = 0; i < n_free; ++i) {
            BoxJob j;
            std::array<double, 8> f;
            memcpy(f.data(), free_q + (size_t)i * 8, 64);
            if (plan_free(f, 0, j)) take(j);
        }
        *n_out = (int)jobs.size();
        if (jobs.empty()) return;
        size_t a_total = 0, w_total = 0;
        bool any_warp = false, any_tall = false;
        std::vector<CropDesc> descs;
        for (size_t i = 0; i < jobs.size(); ++i) {
            CropDesc& d = jobs[i].d;
            d.a_off = (int)a_total;
            a_total += align_up((size_t)d.rw * d.rh, 16);
            if (d.warp) { d.warp_off = (int)w_total; w_total += align_up((size_t)d.sw * d.sh, 16); }
            d.slot = (int)i;
            any_warp |= d.warp != 0;
            any_tall |= !(d.fw == d.rw && d.rh == 64);
        }
        ctx->crop_scratch.ensure(std::max<size_t>(a_total, 16));
        ctx->crop_hscratch.ensure(std::max<size_t>(a_total, 16));
        ctx->crop_wscratch.ensure(std::max<size_t>(w_total, 16));
        for (auto& j : jobs) descs.push_back(j.d);
        ctx->crop_desc.ensure(descs.size() * sizeof(CropDesc));
        ctx->crop_luts.ensure(descs.size() * 256);
        HIPCHK(hipMemcpyAsync(ctx->crop_desc.p, descs.data(), descs.size() * sizeof(CropDesc), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(launch_crops(dev_gray, H, W, (const CropDesc*)ctx->crop_desc.p, 0, (int)descs.size(), imgW, any_warp, any_tall,
                            (uint8_t*)ctx->crop_wscratch.p, (uint8_t*)ctx->crop_scratch.p, (uint8_t*)ctx->crop_hscratch.p,
                            (const uint8_t*)ctx->crop_luts.p, dev_out, 1, ctx->stream));
        if (contrast > 0) {
            ctx->crop_hist.ensure(descs.size() * 256 * 4);
            HIPCHK(launch_crop_hist((const uint8_t*)ctx->crop_scratch.p, (const CropDesc*)ctx->crop_desc.p, 0, (int)descs.size(),
                                    (unsigned int*)ctx->crop_hist.p, ctx->stream));
            std::vector<unsigned int> hist(descs.size() * 256);
            HIPCHK(hipMemcpyAsync(hist.data(), ctx->crop_hist.p, hist.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            std::vector<uint8_t> luts(descs.size() * 256);
            for (size_t k = 0; k < descs.size(); ++k) {
                const size_t npx = (size_t)descs[k].rw * descs[k].rh;
                const double high = percentile_u8(&hist[k * 256], npx, 90.0), lowp = percentile_u8(&hist[k * 256], npx, 10.0);
                const double con = (high - lowp) / std::max(10.0, high + lowp);
                for (int v = 0; v < 256; ++v) {
                    if (con < (double)contrast) {
                        double x = ((double)v - lowp + 25) * (200.0 / std::max(10.0, high - lowp));
                        x = std::max(0.0, std::min(255.0, x));
                        luts[k * 256 + v] = (uint8_t)x;
                    } else {
                        luts[k * 256 + v] = (uint8_t)v;
                    }
                }
                descs[k].lut_off = (int)(k * 256);
            }
            HIPCHK(hipMemcpyAsync(ctx->crop_luts.p, luts.data(), luts.size(), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(ctx->crop_desc.p, descs.data(), descs.size() * sizeof(CropDesc), hipMemcpyHostToDevice, ctx->stream));
        }
        HIPCHK(launch_crops(dev_gray, H, W, (const CropDesc*)ctx->crop_desc.p, 0, (int)descs.size(), imgW, any_warp, any_tall,
                            (uint8_t*)ctx->crop_wscratch.p, (uint8_t*)ctx->crop_scratch.p, (uint8_t*)ctx->crop_hscratch.p,
                            (const uint8_t*)ctx->crop_luts.p, dev_out, 2, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    });
}

}  // extern "C"
