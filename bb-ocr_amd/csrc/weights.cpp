// Weight tables: eval-mode BatchNorm folded into the convolutions, bf16 rounding, MFMA fragment packing (easyocr Reader.__init__ / get_detector / get_recognizer).
#include "ctx.h"

// ------------------------------------------------------------------------------------------------ weights

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

ConvPlan make_plan(int Cin, int Cout, int KH, int KW, int pad, int dil, int el, int bn) {
    ConvPlan p;
    p.el = el;
    p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.pad_h = pad; p.pad_w = pad; p.dil = dil;
    p.Cin_pad = cdiv(Cin, 32) * 32;
    p.BN = bn > 0 ? bn : conv_plan_bn(Cout);
    p.Cout_pad = cdiv(Cout, p.BN) * p.BN;
    return p;
}

void upload_plan(bbocr_ctx* c, ConvPlan& p, const std::vector<float>& w, const std::vector<float>& b) {
    std::vector<uint16_t> pk(conv_packed_elems(p));
    pack_conv_weights(p, w.data(), pk.data());
    std::vector<float> bp(p.Cout_pad, 0.f);
    std::copy(b.begin(), b.end(), bp.begin());
    p.d_w = upload(c, pk);
    p.d_b = upload(c, bp);
}

// Split-fp16 plan of the exact recogniser mode: activations arrive as [hi | lo] pairs of fp16 tensors and the launch reads the channel
// sequence [a_hi | 2048 a_lo | a_hi] (in0 = the pair, in1 = its hi half again), so the packed weights are [w_hi | w_hi / 2048 | w_lo]
// along Cin: ONE launch accumulates a_hi w_hi + a_lo w_hi + a_hi w_lo in fp32 (the dropped lo*lo term is 2^-22 relative).  The weights are
// scaled by a power of two so that their largest magnitude sits in [512, 1024): w_lo = fp16(w 2^s - w_hi) then stays in fp16's
// normal range for every weight that matters; the epilogue multiplies the accumulators by 2^-s (exact).
static void upload_split_plan(bbocr_ctx* c, ConvPlan& p, int Cin, int Cout, int KH, int KW, int pad, const std::vector<float>& w,
                              const std::vector<float>& b, float lo_scale = SPLIT_LO_SCALE, int dil = 1) {    // lo_scale: what the producer multiplied its lo half by
    if (Cin % 32) fail(BBOCR_ERR_INTERNAL, "split plan: Cin must be a multiple of 32");
    const int taps = KH * KW;
    float mx = 0.f;
    for (float v : w) mx = std::max(mx, std::fabs(v));
    int s = 0;
    if (mx > 0.f) s = 9 - (int)std::floor(std::log2((double)mx));          // 2^9 <= mx * 2^s < 2^10
    s = std::max(-14, std::min(24, s));
    const float sc = std::ldexp(1.f, s);
    std::vector<float> w3((size_t)Cout * 3 * Cin * taps);
    for (int o = 0; o < Cout; ++o)
        for (int i = 0; i < Cin; ++i)
            for (int t = 0; t < taps; ++t) {
                const float v = w[((size_t)o * Cin + i) * taps + t] * sc;
                const float hi = f16_to_f32_host(f32_to_f16_host(v));
                const float lo = f16_to_f32_host(f32_to_f16_host(v - hi));
                float* row = w3.data() + (size_t)o * 3 * Cin * taps;
                row[((size_t)i) * taps + t] = hi;
                row[((size_t)Cin + i) * taps + t] = hi * (1.f / lo_scale);            // the lo activations are stored x lo_scale (2048; 1 behind the LSTM)
                row[((size_t)2 * Cin + i) * taps + t] = lo;
            }
    p = make_plan(3 * Cin, Cout, KH, KW, pad, dil, 1);
    p.acc_scale = std::ldexp(1.f, -s);
    p.split = 1;
    upload_plan(c, p, w3, b);
}

// 1x1 conv over a channel concat [y (Cy) | skip (Cs)], BN folded, split into the two column blocks
static void load_split_1x1(bbocr_ctx* c, const TensorMap& tm, ConvPlan& py, ConvPlan& ps, const std::string& conv, const std::string& bn, int Cy,
                           int Cs, int Cout, std::vector<float>* wy_out = nullptr) {
    std::vector<float> w, b;
    fold_conv(tm, conv, bn, Cout, Cy + Cs, 1, w, b);
    std::vector<float> wy((size_t)Cout * Cy), ws((size_t)Cout * Cs), zero(Cout, 0.f);
    for (int o = 0; o < Cout; ++o) {
        std::copy(w.begin() + (size_t)o * (Cy + Cs), w.begin() + (size_t)o * (Cy + Cs) + Cy, wy.begin() + (size_t)o * Cy);
        std::copy(w.begin() + (size_t)o * (Cy + Cs) + Cy, w.begin() + (size_t)(o + 1) * (Cy + Cs), ws.begin() + (size_t)o * Cs);
    }
    if (wy_out) *wy_out = wy;
    py = make_plan(Cy, Cout, 1, 1, 0, 1, det_el(c));
    upload_plan(c, py, wy, zero);
    ps = make_plan(Cs, Cout, 1, 1, 0, 1, det_el(c));
    upload_plan(c, ps, ws, b);
}

static void load_layer(bbocr_ctx* c, const TensorMap& tm, ConvPlan& p, const std::string& conv, const std::string& bn, int Cin, int Cout,
                       int K, int pad, int dil, int el, bool split = false, int tile_bn = 0) {
    std::vector<float> w, b;
    fold_conv(tm, conv, bn, Cout, Cin, K * K, w, b);
    if (split) {
        upload_split_plan(c, p, Cin, Cout, K, K, pad, w, b, SPLIT_LO_SCALE, dil);
        return;
    }
    p = make_plan(Cin, Cout, K, K, pad, dil, el, tile_bn);
    upload_plan(c, p, w, b);
}

void free_weights(bbocr_ctx* c) {
    for (void* p : c->owned) (void)hipFree(p);
    c->owned.clear();
    c->owned_bytes.clear();
    c->craft_loaded = c->crnn_loaded = false;
}

// ------------------------------------------------------------------------------------------------ weight blob (multi-GPU broadcast)
// Everything bbocr_load_weights leaves on the device -- BN-folded, rounded to the context's element type, packed in MFMA fragment
// order -- as ONE contiguous device buffer: [header 64 B][block 0][block 1]...[trailer: value-dependent host scalars], blocks padded
// to 256 B.  Rank 0 exports it, RCCL broadcasts it device-to-device over xGMI, the other ranks (whose plans were laid out by
// bbocr_alloc_weights) import it: no fp32 state-dict, no host hop and no re-packing on the receivers (~49 MB in
// bf16 instead of 98 MB of fp32).
namespace {
struct BlobHeader { unsigned long long magic; int precision, nblocks; unsigned long long bytes; int craft, crnn; unsigned long long layout_hash; int pad[6]; };
static_assert(sizeof(BlobHeader) == 64, "blob header");
constexpr unsigned long long BLOB_MAGIC = 0x62626f6372776231ULL;   // "bbocrwb1"
unsigned long long layout_hash(const bbocr_ctx* c) {     // FNV-1a over the per-block sizes: equal totals with different blocks do not pass
    unsigned long long h = 1469598103934665603ULL;
    for (size_t b : c->owned_bytes)
        for (int i = 0; i < 8; ++i) { h ^= (unsigned long long)((b >> (8 * i)) & 0xff); h *= 1099511628211ULL; }
    return h;
}
std::vector<float*> blob_scalars(bbocr_ctx* c) {       // host scalars that depend on the weight VALUES (power-of-two scales of split plans)
    std::vector<float*> v;
    for (ConvPlan* p : {&c->r1, &c->r2, &c->r3, &c->r4, &c->r5, &c->r6, &c->xproj[0], &c->xproj[1], &c->lin[0], &c->lin[1], &c->pred}) v.push_back(&p->acc_scale);
    v.push_back(&c->whh_scale[0]);
    v.push_back(&c->whh_scale[1]);
    for (ConvPlan* p : {&c->conv1_2, &c->conv2_1, &c->conv2_2, &c->conv3_1, &c->conv3_2, &c->conv3_3, &c->conv4_1, &c->conv4_2, &c->conv4_3, &c->conv5_1,
                        &c->conv5_2, &c->fc6, &c->fc7, &c->up1a, &c->up1b, &c->up2s, &c->up2b, &c->up3s, &c->up3b, &c->up4s, &c->up4b, &c->cls0, &c->cls2, &c->cls4})
        v.push_back(&p->acc_scale);       // exact mode: the detector's split plans (1 otherwise)
    return v;
}
}  // namespace

size_t weights_blob_bytes(const bbocr_ctx* c) {
    size_t n = sizeof(BlobHeader);
    for (size_t b : c->owned_bytes) n += align_up(b, 256);
    return n + 256;                                     // trailer: 64 floats (blob_scalars: 37 used)
}

void weights_export(bbocr_ctx* c, void* dev_dst, size_t bytes) {
    if (!c->craft_loaded && !c->crnn_loaded) fail(BBOCR_ERR_STATE, "no weights loaded");
    if (bytes != weights_blob_bytes(c)) fail(BBOCR_ERR_ARG, "weight blob: wrong size");
    BlobHeader h{};
    h.magic = BLOB_MAGIC; h.precision = c->cfg.precision; h.nblocks = (int)c->owned.size(); h.bytes = bytes;
    h.craft = c->craft_loaded; h.crnn = c->crnn_loaded; h.layout_hash = layout_hash(c);
    char* d = (char*)dev_dst;
    HIPCHK(hipMemcpyAsync(d, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    size_t off = sizeof(h);
    for (size_t i = 0; i < c->owned.size(); ++i) {
        HIPCHK(hipMemcpyAsync(d + off, c->owned[i], c->owned_bytes[i], hipMemcpyDeviceToDevice, c->stream));
        off += align_up(c->owned_bytes[i], 256);
    }
    float tr[64] = {0};
    const std::vector<float*> sc = blob_scalars(c);
    for (size_t i = 0; i < sc.size(); ++i) tr[i] = *sc[i];
    HIPCHK(hipMemcpyAsync(d + off, tr, sizeof(tr), hipMemcpyHostToDevice, c->stream));
    slot_sync(c, c->stream);
}

void weights_import(bbocr_ctx* c, const void* dev_src, size_t bytes) {
    if (bytes != weights_blob_bytes(c)) fail(BBOCR_ERR_ARG, "weight blob: size does not match this context's layout (same precision / networks on every rank?)");
    const char* s = (const char*)dev_src;
    BlobHeader h{};
    HIPCHK(hipMemcpy(&h, s, sizeof(h), hipMemcpyDeviceToHost));
    if (h.magic != BLOB_MAGIC || h.precision != c->cfg.precision || h.nblocks != (int)c->owned.size() || h.bytes != bytes ||
        h.craft != (int)c->craft_loaded || h.crnn != (int)c->crnn_loaded || h.layout_hash != layout_hash(c))
        fail(BBOCR_ERR_WEIGHTS, "weight blob does not match this context (precision, networks or layout differ)");
    size_t off = sizeof(h);
    for (size_t i = 0; i < c->owned.size(); ++i) {
        HIPCHK(hipMemcpyAsync(c->owned[i], s + off, c->owned_bytes[i], hipMemcpyDeviceToDevice, c->stream));
        off += align_up(c->owned_bytes[i], 256);
    }
    float tr[64];
    HIPCHK(hipMemcpyAsync(tr, s + off, sizeof(tr), hipMemcpyDeviceToHost, c->stream));
    slot_sync(c, c->stream);
    const std::vector<float*> sc = blob_scalars(c);
    for (size_t i = 0; i < sc.size(); ++i) *sc[i] = tr[i];
}

// EXACT mode: every detector layer as a split-fp16 plan over pair tensors; the layers the fast modes fuse or commute (conv1_1 inside
// conv1_2, the U-net 1x1s split by linearity, upconv4 in one launch, the classifier tail in conv_cls.4's epilogue) run in the reference's
// own operation order instead (detector.cpp::craft_forward_exact).
static void load_craft_exact(bbocr_ctx* c, const TensorMap& tm) {
    {
        std::vector<float> w, b;
        fold_conv(tm, "basenet.slice1.0", "basenet.slice1.1", 64, 3, 9, w, b);
        c->c11_w32 = upload(c, w);
        c->c11_b = upload(c, b);
    }
    auto L = [&](ConvPlan& p, const char* conv, const char* bn, int Cin, int Cout, int K, int pad, int dil) {
        load_layer(c, tm, p, conv, bn, Cin, Cout, K, pad, dil, 1, true);
    };
    L(c->conv1_2, "basenet.slice1.3", "basenet.slice1.4", 64, 64, 3, 1, 1);
    L(c->conv2_1, "basenet.slice1.7", "basenet.slice1.8", 64, 128, 3, 1, 1);
    L(c->conv2_2, "basenet.slice1.10", "basenet.slice1.11", 128, 128, 3, 1, 1);
    L(c->conv3_1, "basenet.slice2.14", "basenet.slice2.15", 128, 256, 3, 1, 1);
    L(c->conv3_2, "basenet.slice2.17", "basenet.slice2.18", 256, 256, 3, 1, 1);
    L(c->conv3_3, "basenet.slice3.20", "basenet.slice3.21", 256, 256, 3, 1, 1);
    L(c->conv4_1, "basenet.slice3.24", "basenet.slice3.25", 256, 512, 3, 1, 1);
    L(c->conv4_2, "basenet.slice3.27", "basenet.slice3.28", 512, 512, 3, 1, 1);
    L(c->conv4_3, "basenet.slice4.30", "basenet.slice4.31", 512, 512, 3, 1, 1);
    L(c->conv5_1, "basenet.slice4.34", "basenet.slice4.35", 512, 512, 3, 1, 1);
    L(c->conv5_2, "basenet.slice4.37", "basenet.slice4.38", 512, 512, 3, 1, 1);
    L(c->fc6, "basenet.slice5.1", "", 512, 1024, 3, 6, 6);
    L(c->fc7, "basenet.slice5.2", "", 1024, 1024, 1, 0, 1);
    L(c->up1a, "upconv1.conv.0", "upconv1.conv.1", 1536, 512, 1, 0, 1);
    L(c->up1b, "upconv1.conv.3", "upconv1.conv.4", 512, 256, 3, 1, 1);
    L(c->up2s, "upconv2.conv.0", "upconv2.conv.1", 768, 256, 1, 0, 1);        // over cat[up(y), skip], channels in torch.cat's order
    L(c->up2b, "upconv2.conv.3", "upconv2.conv.4", 256, 128, 3, 1, 1);
    L(c->up3s, "upconv3.conv.0", "upconv3.conv.1", 384, 128, 1, 0, 1);
    L(c->up3b, "upconv3.conv.3", "upconv3.conv.4", 128, 64, 3, 1, 1);
    L(c->up4s, "upconv4.conv.0", "upconv4.conv.1", 192, 64, 1, 0, 1);
    L(c->up4b, "upconv4.conv.3", "upconv4.conv.4", 64, 32, 3, 1, 1);
    L(c->cls0, "conv_cls.0", "", 32, 32, 3, 1, 1);
    L(c->cls2, "conv_cls.2", "", 32, 32, 3, 1, 1);
    L(c->cls4, "conv_cls.4", "", 32, 16, 3, 1, 1);
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
        c->cls6_w32 = upload(c, std::vector<float>(w1, w1 + 256));
    }
    c->craft_loaded = true;
}

void load_craft(bbocr_ctx* c, const TensorMap& tm) {
    if (det_split(c)) { load_craft_exact(c, tm); return; }
    {
        std::vector<float> w, b;
        fold_conv(tm, "basenet.slice1.0", "basenet.slice1.1", 64, 3, 9, w, b);
        std::vector<uint16_t> pk(4 * 64 * 8);
        pack_conv1_1_weights_fused(w.data(), pk.data(), det_el(c));
        c->c11_wf = upload(c, pk);
        c->c11_b = upload(c, b);
    }
    load_layer(c, tm, c->conv1_2, "basenet.slice1.3", "basenet.slice1.4", 64, 64, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv2_1, "basenet.slice1.7", "basenet.slice1.8", 64, 128, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv2_2, "basenet.slice1.10", "basenet.slice1.11", 128, 128, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv3_1, "basenet.slice2.14", "basenet.slice2.15", 128, 256, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv3_2, "basenet.slice2.17", "basenet.slice2.18", 256, 256, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv3_3, "basenet.slice3.20", "basenet.slice3.21", 256, 256, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv4_1, "basenet.slice3.24", "basenet.slice3.25", 256, 512, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv4_2, "basenet.slice3.27", "basenet.slice3.28", 512, 512, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv4_3, "basenet.slice4.30", "basenet.slice4.31", 512, 512, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv5_1, "basenet.slice4.34", "basenet.slice4.35", 512, 512, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->conv5_2, "basenet.slice4.37", "basenet.slice4.38", 512, 512, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->fc6, "basenet.slice5.1", "", 512, 1024, 3, 6, 6, det_el(c));
    // the long-K 1x1 GEMMs run 256-cout tiles (8 waves): their 128-cout tiles moved 24 KB of LDS-DMA per 2.1 MFLOP k-step and sat at the
    // DMA fill rate (~8 TB/s, 0.71 PFLOP/s); 32 KB per 4.2 MFLOP now
    static const int bn1x1 = diag_knob("BBOCR_BN256_1X1", 1) ? 256 : 0;
    load_layer(c, tm, c->fc7, "basenet.slice5.2", "", 1024, 1024, 1, 0, 1, det_el(c), false, bn1x1);
    load_layer(c, tm, c->up1a, "upconv1.conv.0", "upconv1.conv.1", 1536, 512, 1, 0, 1, det_el(c), false, bn1x1);
    load_layer(c, tm, c->up1b, "upconv1.conv.3", "upconv1.conv.4", 512, 256, 3, 1, 1, det_el(c));
    load_split_1x1(c, tm, c->up2y, c->up2s, "upconv2.conv.0", "upconv2.conv.1", 256, 512, 256);
    load_layer(c, tm, c->up2b, "upconv2.conv.3", "upconv2.conv.4", 256, 128, 3, 1, 1, det_el(c));
    load_split_1x1(c, tm, c->up3y, c->up3s, "upconv3.conv.0", "upconv3.conv.1", 128, 256, 128);
    load_layer(c, tm, c->up3b, "upconv3.conv.3", "upconv3.conv.4", 128, 64, 3, 1, 1, det_el(c));
    {
        std::vector<float> wy;
        load_split_1x1(c, tm, c->up4y, c->up4s, "upconv4.conv.0", "upconv4.conv.1", 64, 128, 64, &wy);
        std::vector<uint16_t> pk(2 * 4 * 64 * 8);
        pack_post1x1_weights(wy.data(), pk.data(), det_el(c));      // z = W_y u3b in upconv3.3x3's epilogue (detector.cpp)
        c->up4y_post = upload(c, pk);
    }
    load_layer(c, tm, c->up4b, "upconv4.conv.3", "upconv4.conv.4", 64, 32, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->cls0, "conv_cls.0", "", 32, 32, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->cls2, "conv_cls.2", "", 32, 32, 3, 1, 1, det_el(c));
    load_layer(c, tm, c->cls4, "conv_cls.4", "", 32, 16, 3, 1, 1, det_el(c));
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
                fr[l * 8 + j] = f32_to_el_host(det_el(c), k < 16 ? w1[o * 16 + k] : 0.f);
            }
        c->cls_tail_frag = upload(c, fr);
    }
    c->craft_loaded = true;
}

void load_crnn(bbocr_ctx* c, const TensorMap& tm) {
    const std::string fe = "FeatureExtraction.ConvNet.";
    {
        std::vector<float> t(32 * 9 + 32);
        const float* w = tm.get(fe + "0.weight", 32 * 9);
        const float* b = tm.get(fe + "0.bias", 32);
        for (int co = 0; co < 32; ++co)                       // tap-major [9][32]: the kernel reads channel PAIRS of one tap (packed FMA)
            for (int tap = 0; tap < 9; ++tap) t[tap * 32 + co] = w[co * 9 + tap];
        std::copy(b, b + 32, t.begin() + 288);
        c->r0_wb = upload(c, t);
        if (!rec_split(c)) {
            std::vector<uint16_t> af(2 * 64 * 8);
            pack_crnn_conv0_mfma(t.data(), af.data(), rec_el(c));
            c->r0_afrag = upload(c, af);
        }
    }
    load_layer(c, tm, c->r1, fe + "3", "", 32, 64, 3, 1, 1, rec_el(c), rec_split(c));
    load_layer(c, tm, c->r2, fe + "6", "", 64, 128, 3, 1, 1, rec_el(c), rec_split(c));
    load_layer(c, tm, c->r3, fe + "8", "", 128, 128, 3, 1, 1, rec_el(c), rec_split(c));
    load_layer(c, tm, c->r4, fe + "11", fe + "12", 128, 256, 3, 1, 1, rec_el(c), rec_split(c));
    load_layer(c, tm, c->r5, fe + "14", fe + "15", 256, 256, 3, 1, 1, rec_el(c), rec_split(c));
    load_layer(c, tm, c->r6, fe + "18", "", 256, 256, 2, 0, 1, rec_el(c), rec_split(c));
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
        const float* hf = tm.get(sm + "rnn.weight_hh_l0", (size_t)1024 * 256);
        const float* hb = tm.get(sm + "rnn.weight_hh_l0_reverse", (size_t)1024 * 256);
        if (rec_split(c)) {
            upload_split_plan(c, c->xproj[l], 256, 2048, 1, 1, 0, w, b);
            std::vector<uint16_t> pk(lstm_whh_split_packed_elems());
            c->whh_scale[l] = pack_lstm_whh_split(hf, hb, pk.data());
            c->whh[l] = upload(c, pk);
        } else {
            c->xproj[l] = make_plan(256, 2048, 1, 1, 0, 1, rec_el(c), diag_knob("BBOCR_BN256_XPROJ", 1) ? 256 : 0);
            upload_plan(c, c->xproj[l], w, b);
            std::vector<uint16_t> pk(lstm_whh_packed_elems());
            pack_lstm_whh8(hf, hb, pk.data(), rec_el(c));
            c->whh[l] = upload(c, pk);
        }
        std::vector<float> lw(tm.get(sm + "linear.weight", (size_t)256 * 512), tm.get(sm + "linear.weight", (size_t)256 * 512) + 256 * 512);
        std::vector<float> lb(tm.get(sm + "linear.bias", 256), tm.get(sm + "linear.bias", 256) + 256);
        if (rec_split(c)) {
            upload_split_plan(c, c->lin[l], 512, 256, 1, 1, 0, lw, lb, 1.f);     // reads lstm_exact_kernel's pair: lo = fp16(h - hi), unscaled
        } else {
            c->lin[l] = make_plan(512, 256, 1, 1, 0, 1, rec_el(c), diag_knob("BBOCR_BN256_LIN", 1) ? 256 : 0);
            upload_plan(c, c->lin[l], lw, lb);
        }
    }
    {
        std::vector<float> pw(tm.get("Prediction.weight", (size_t)97 * 256), tm.get("Prediction.weight", (size_t)97 * 256) + 97 * 256);
        std::vector<float> pb(tm.get("Prediction.bias", 97), tm.get("Prediction.bias", 97) + 97);
        if (rec_split(c)) {
            upload_split_plan(c, c->pred, 256, 97, 1, 1, 0, pw, pb);
        } else {
            c->pred = make_plan(256, 97, 1, 1, 0, 1, rec_el(c));
            upload_plan(c, c->pred, pw, pb);
        }
    }
    c->crnn_loaded = true;
}
