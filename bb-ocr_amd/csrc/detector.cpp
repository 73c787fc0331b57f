// CRAFT forward pass orchestration (easyocr craft.py / detection.py::test_net) and box extraction (craft_utils.py::getDetBoxes_core + utils.py::group_text_box).
#include "ctx.h"

// ------------------------------------------------------------------------------------------------ conv helper
hipError_t launch_conv_profiled(bbocr_ctx* c, const ConvPlan& p, ConvArgs a, bool may_decline) {
    a.zero = c->zero_page;
    if (c->profiling == 0 || (c->profiling == 1 && c->prof_group != 0)) {
        const hipError_t e = launch_conv(p, a, c->cur);
        if (may_decline && e == hipErrorNotSupported) return e;
        HIPCHK(e);
        return hipSuccess;
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
    r.flops = 2.0 * a.N * OH * OW * (double)p.Cout * (p.split ? p.Cin / 3 : p.Cin) * p.KH * p.KW;   // algorithmic (unpadded) work; a split-fp16 plan
                                                                                                     // executes three product terms per MAC: counted once
    if (a.c11_w) r.flops += 2.0 * a.N * a.H * a.W * 64.0 * 27.0;            // conv1_1 produced inside this launch
    if (a.tail) r.flops += 2.0 * a.N * OH * OW * (16.0 * 16.0 + 16.0 * 2.0);   // fused classifier tail
    if (a.post_w) r.flops += 2.0 * a.N * OH * OW * 64.0 * 64.0;                 // 1x1 applied in the epilogue
    r.group = c->prof_group;
    HIPCHK(hipEventRecord(r.e0, c->cur));
    const hipError_t e = launch_conv(p, a, c->cur);
    HIPCHK(hipEventRecord(r.e1, c->cur));
    if (may_decline && e == hipErrorNotSupported) {
        c->prof_pool.push_back(r.e0);
        c->prof_pool.push_back(r.e1);
        return e;
    }
    HIPCHK(e);
    c->prof_recs.push_back(r);
    return hipSuccess;
}

// the fused upconv4 launch, timed like two conv launches (algorithmic FLOPs of the 1x1 over s1 and of the 3x3)
static hipError_t launch_up4_profiled(bbocr_ctx* c, const ConvArgs& a) {
    if (c->profiling == 0 || (c->profiling == 1 && c->prof_group != 0)) return launch_up4_fused(c->up4s, c->up4b, a, c->cur);
    hipEvent_t e0, e1;
    auto get_event = [&](hipEvent_t& e) {
        if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
        else HIPCHK(hipEventCreate(&e));
    };
    get_event(e0);
    get_event(e1);
    HIPCHK(hipEventRecord(e0, c->cur));
    const hipError_t r = launch_up4_fused(c->up4s, c->up4b, a, c->cur);
    HIPCHK(hipEventRecord(e1, c->cur));
    if (r != hipSuccess) { c->prof_pool.push_back(e0); c->prof_pool.push_back(e1); return r; }
    bbocr_ctx::ProfRec rec;
    rec.e0 = e0; rec.e1 = e1; rec.group = c->prof_group;
    rec.flops = 2.0 * a.N * a.H * a.W * (128.0 * 64.0 + 64.0 * 32.0 * 9.0);
    c->prof_recs.push_back(rec);
    return hipSuccess;
}

// Split-fp16 plans (ConvPlan::split, exact recogniser mode): the activation `a0` is a pair tensor [hi | lo] whose Act::C counts BOTH
// halves; the launch reads [hi | lo | hi] (in1 = the hi half again) and, unless it writes fp32, stores its output as a pair too.
static void conv_sources(const ConvPlan& p, ConvArgs& a, const Act& a0, const Act* a1) {
    a.in0 = a0.p; a.C0 = a0.C; a.in0_cs = a0.C;
    if (p.split) {
        if (a1 || (a0.C & 63)) fail(BBOCR_ERR_INTERNAL, "split-fp16 conv: one pair tensor with a multiple of 32 logical channels expected");
        a.in1 = a0.p; a.C1 = a0.C / 2; a.in1_cs = a0.C;
    } else if (a1) {
        a.in1 = a1->p; a.C1 = a1->C; a.in1_cs = a1->C;
    }
}

void run_conv(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, const Act* a1, bool relu1, bool relu_out, void* out,
                     int out_cs, int cout_store, bool out_f32, const Act* addup) {
    if (c->arena.dry) return;
    ConvArgs a{};
    if (addup) { a.addup = addup->p; a.up_H = a0.H; a.up_W = a0.W; a.up_cs = addup->C; }
    conv_sources(p, a, a0, a1);
    a.N = a0.N; a.H = a0.H; a.W = a0.W;
    a.relu_in0 = relu0; a.relu_in1 = p.split ? relu0 : relu1; a.relu_out = relu_out; a.out_f32 = out_f32;
    a.out = out; a.out_cs = out_cs; a.cout_store = cout_store;
    if (p.split && !out_f32) a.split_off = cout_store;          // out_cs is 2 * cout_store then
    (void)launch_conv_profiled(c, p, a);
}

// after the stream has drained: fold the recorded launches into the per-group totals
void prof_collect(bbocr_ctx* c) {
    bbocr_ctx* root = c->root;
    std::lock_guard<std::mutex> lk(root->pool_mu);
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            root->prof_ms[r.group] += ms;
            root->prof_flops[r.group] += r.flops;
            root->prof_launches[r.group] += 1;
        }
        c->prof_pool.push_back(r.e0);
        c->prof_pool.push_back(r.e1);
    }
    c->prof_recs.clear();
}

// conv producing a fresh bf16 activation with `store` channels (multiple of 16)
Act conv_act(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, const Act* a1, bool relu1, bool relu_out, int store) {
    const int OH = a0.H + 2 * p.pad_h - (p.KH - 1) * p.dil, OW = a0.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    const int cs = p.split ? 2 * store : store;                  // pair tensors carry [hi | lo]
    Act o{c->arena.alloc<uint16_t>((size_t)a0.N * OH * OW * cs), a0.N, OH, OW, cs};
    run_conv(c, p, a0, relu0, a1, relu1, relu_out, o.p, cs, store, false);
    return o;
}

// conv with the max-pool fused into its epilogue.  mode 1 = MaxPool2d(2,2), 2 = MaxPool2d((2,1),(2,1)).  Returns the pooled
// activation; when `full` is given the un-pooled conv output (bias, relu_out) is written too (U-net skip tensors).
Act conv_pool_act(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, bool relu_out, int store, int mode, bool pool_relu,
                         Act* full, const RgbSource* rgb) {
    const int OH = a0.H + 2 * p.pad_h - (p.KH - 1) * p.dil, OW = a0.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    const int PH = OH / 2, PW = mode == 1 ? OW / 2 : OW;
    const int cs = p.split ? 2 * store : store;
    if (full) *full = Act{c->arena.alloc<uint16_t>((size_t)a0.N * OH * OW * cs), a0.N, OH, OW, cs};
    Act o{c->arena.alloc<uint16_t>((size_t)a0.N * PH * PW * cs), a0.N, PH, PW, cs};
    if (c->arena.dry) return o;
    ConvArgs a{};
    if (!rgb) conv_sources(p, a, a0, nullptr);
    else { a.C0 = a0.C; a.in0_cs = a0.C; }
    a.N = a0.N; a.H = a0.H; a.W = a0.W;
    a.relu_in0 = relu0; a.relu_in1 = relu0; a.relu_out = relu_out; a.out_f32 = 0;
    a.out = full ? (void*)full->p : nullptr; a.out_cs = cs; a.cout_store = store;
    if (p.split) a.split_off = store;
    a.pool_mode = mode; a.pool_relu = pool_relu; a.store_full = full != nullptr; a.pool_cs = cs; a.pool_out = o.p;
    if (rgb) { a.in0 = (const uint16_t*)rgb->rgb; a.c11_w = c->c11_wf; a.c11_b = c->c11_b; a.rgb_H = rgb->Himg; a.rgb_W = rgb->Wimg; }
    (void)launch_conv_profiled(c, p, a);
    return o;
}

Act pool_act(bbocr_ctx* c, const Act& a, int kh, int kw, int sh, int sw, int ph, int pw, bool relu_in) {
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
    // normalise + conv1_1 + ReLU are produced inside conv1_2's prologue (its 64-channel input never reaches HBM)
    const Act canvas{nullptr, nb, H32, W32, 64};
    const RgbSource src{rgb, Himg, Wimg};
    Act p1 = conv_pool_act(c, c->conv1_2, canvas, false, true, 64, 1, false, nullptr, &src);
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
    // upconv3's 3x3 with z = W_y u3b (the y half of upconv4's 1x1, linear) applied in its epilogue: u3b itself has no other reader and is
    // not stored, one launch and 2 x 9.8 MB per page less; shapes the kernel declines take the two launches
    Act u3b{c->arena.alloc<uint16_t>((size_t)u3a.N * u3a.H * u3a.W * 64), u3a.N, u3a.H, u3a.W, 64};
    Act z{c->arena.alloc<uint16_t>((size_t)u3a.N * u3a.H * u3a.W * 64), u3a.N, u3a.H, u3a.W, 64};
    if (!ar.dry) {
        static const bool post_on = (diag_knob("BBOCR_UP3_POST", 1) != 0);     // A/B knob
        hipError_t e = hipErrorNotSupported;
        if (post_on && c->up4y_post && !c->up3b.split) {
            ConvArgs a{};
            a.in0 = u3a.p; a.C0 = u3a.C; a.in0_cs = u3a.C;
            a.N = u3a.N; a.H = u3a.H; a.W = u3a.W;
            a.relu_out = 1; a.out = z.p; a.out_cs = 64; a.cout_store = 64; a.post_w = c->up4y_post;
            e = launch_conv_profiled(c, c->up3b, a, true);
        }
        if (e != hipSuccess) {
            run_conv(c, c->up3b, u3a, false, nullptr, false, true, u3b.p, 64, 64, false);
            run_conv(c, c->up4y, u3b, false, nullptr, false, false, z.p, 64, 64, false);
        }
    }
    // upconv4 as ONE launch behind z = W_y u3b: the 3x3 produces its own input patch (1x1 over s1 + up(z) + ReLU) in LDS, so the
    // 64-channel u4a never reaches HBM (conv_mfma.hip::conv3x3_up4_kernel); any other shape takes the two launches
    Act u4b{nullptr, s1.N, s1.H, s1.W, 32};
    {
        u4b.p = c->arena.alloc<uint16_t>((size_t)s1.N * s1.H * s1.W * 32);
        bool fused = false;
        if (!ar.dry) {
            ConvArgs a{};
            a.in0 = s1.p; a.C0 = s1.C; a.in0_cs = s1.C;
            a.N = s1.N; a.H = s1.H; a.W = s1.W;
            a.addup = z.p; a.up_H = s1.H; a.up_W = s1.W; a.up_cs = z.C;
            a.relu_out = 1; a.out = u4b.p; a.out_cs = 32; a.cout_store = 32;
            a.zero = c->zero_page;
            const hipError_t e = launch_up4_profiled(c, a);
            if (e == hipSuccess) fused = true;
            else if (e != hipErrorNotSupported) HIPCHK(e);
        }
        if (!fused) {
            Act u4a{c->arena.alloc<uint16_t>((size_t)s1.N * s1.H * s1.W * 64), s1.N, s1.H, s1.W, 64};
            run_conv(c, c->up4s, s1, false, nullptr, false, true, u4a.p, 64, 64, false, &z);
            run_conv(c, c->up4b, u4a, false, nullptr, false, true, u4b.p, 32, 32, false);
        }
    }
    Act c1 = conv_act(c, c->cls0, u4b, false, nullptr, false, true, 32);
    Act c2 = conv_act(c, c->cls2, c1, false, nullptr, false, true, 32);
    // conv_cls.4 (3x3 32->16 + ReLU) with conv_cls.6/.8 fused into its epilogue: writes the fp32 heat-map directly
    if (!ar.dry) {
        ConvArgs a{};
        a.in0 = c2.p; a.C0 = c2.C; a.in0_cs = c2.C;
        a.N = c2.N; a.H = c2.H; a.W = c2.W;
        a.relu_out = 1; a.out = heat; a.out_cs = 16; a.cout_store = 16; a.tail = c->cls_tail; a.tail_frag = c->cls_tail_frag;
        (void)launch_conv_profiled(c, c->cls4, a);
    }
}

// EXACT mode: the same network on pair tensors [hi | lo] (22 significand bits) with split-fp16 plans in every layer, in the reference's own
// operation order (easyocr/craft.py::CRAFT.forward): conv1_1 as its own fp32 launch, ReLU / pool5 / F.interpolate + torch.cat as
// element-wise passes on the pair values (craft_pair.hip), the U-net 1x1s over the materialised concat, the classifier tail in fp32.
// 3x the MFMA work of the fp16 pass plus the un-fused intermediates: the price of threshold decisions that follow the fp32 CPU path's
// on ANY heat-map, not only on maps with margins.
static void craft_forward_exact(bbocr_ctx* c, const uint8_t* rgb, int nb, int Himg, int Wimg, int H32, int W32, float* heat) {
    Arena& ar = c->arena;
    c->prof_group = 0;
    auto pair = [&](int N, int H, int W, int C) { return Act{ar.alloc<uint16_t>((size_t)N * H * W * C * 2), N, H, W, C * 2}; };   // Act::C counts both halves
    auto relu = [&](const Act& a) {
        Act o = pair(a.N, a.H, a.W, a.C / 2);
        if (!ar.dry) HIPCHK(launch_pair_relu(a.p, o.p, (size_t)a.N * a.H * a.W, a.C / 2, c->cur));
        return o;
    };
    auto upcat = [&](const Act& y, const Act& skip) {
        Act o = pair(skip.N, skip.H, skip.W, y.C / 2 + skip.C / 2);
        if (!ar.dry) HIPCHK(launch_pair_upcat(y.p, y.H, y.W, y.C / 2, skip.p, skip.C / 2, o.p, skip.N, skip.H, skip.W, c->cur));
        return o;
    };
    auto conv = [&](const ConvPlan& p, const Act& a, bool relu_out, int store) { return conv_act(c, p, a, false, nullptr, false, relu_out, store); };
    Act x0 = pair(nb, H32, W32, 64);
    if (!ar.dry) HIPCHK(launch_pair_conv1_1(rgb, nb, Himg, Wimg, H32, W32, c->c11_w32, c->c11_b, x0.p, c->cur));
    Act p1 = conv_pool_act(c, c->conv1_2, x0, false, true, 64, 1, false, nullptr);
    Act a3 = conv(c->conv2_1, p1, true, 128);
    Act s1;
    Act p2 = conv_pool_act(c, c->conv2_2, a3, false, false, 128, 1, true, &s1);     // slice1 ends on BN (s1); slice2 opens with ReLU + pool
    Act a5 = conv(c->conv3_1, p2, true, 256);
    Act s2 = conv(c->conv3_2, a5, false, 256);
    Act p3 = conv_pool_act(c, c->conv3_3, relu(s2), false, true, 256, 1, false, nullptr);
    Act a8 = conv(c->conv4_1, p3, true, 512);
    Act s3 = conv(c->conv4_2, a8, false, 512);
    Act p4 = conv_pool_act(c, c->conv4_3, relu(s3), false, true, 512, 1, false, nullptr);
    Act a11 = conv(c->conv5_1, p4, true, 512);
    Act s4 = conv(c->conv5_2, a11, false, 512);
    Act p5 = pair(s4.N, s4.H, s4.W, 512);
    if (!ar.dry) HIPCHK(launch_pair_maxpool3x3s1(s4.p, p5.p, s4.N, s4.H, s4.W, 512, c->cur));
    Act f6 = conv(c->fc6, p5, false, 1024);
    Act f7 = conv(c->fc7, f6, false, 1024);
    Act u1b = conv(c->up1b, conv(c->up1a, upcat(f7, s4), true, 512), true, 256);
    Act u2b = conv(c->up2b, conv(c->up2s, upcat(u1b, s3), true, 256), true, 128);
    Act u3b = conv(c->up3b, conv(c->up3s, upcat(u2b, s2), true, 128), true, 64);
    Act u4b = conv(c->up4b, conv(c->up4s, upcat(u3b, s1), true, 64), true, 32);
    Act c3 = conv(c->cls4, conv(c->cls2, conv(c->cls0, u4b, true, 32), true, 32), true, 16);
    if (!ar.dry) HIPCHK(launch_pair_cls_tail(c3.p, c->cls6_w32, c->cls_tail, heat, (size_t)c3.N * c3.H * c3.W, c->cur));
}

static void craft_forward_any(bbocr_ctx* c, const uint8_t* rgb, int nb, int Himg, int Wimg, int H32, int W32, float* heat) {
    if (det_split(c)) craft_forward_exact(c, rgb, nb, Himg, Wimg, H32, W32, heat);
    else craft_forward(c, rgb, nb, Himg, Wimg, H32, W32, heat);
}

DetDims det_dims(int H, int W, int canvas, double mag) {
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

void detect_impl(bbocr_ctx* c, const uint8_t* rgb, int B, int H, int W, const bbocr_params& p, float* heat,
                        const std::function<void(int, int)>& after_sub) {
    if (!c->craft_loaded) fail(BBOCR_ERR_STATE, "detector weights not loaded");
    if (B <= 0 || H <= 0 || W <= 0) fail(BBOCR_ERR_ARG, "bad page batch shape");
    const DetDims d = det_dims(H, W, p.canvas_size, p.mag_ratio);
    if (d.th <= 0 || d.tw <= 0) fail(BBOCR_ERR_ARG, "page collapses to zero size");
    // Pages per detector pass.  Explicit det_sub_batch: uniform passes of that size.  Auto: passes as large as a 96 GB
    // activation arena allows (sized by a dry run on one page; at most 64 pages), and -- when the caller overlaps box
    // extraction with the next pass (readtext_batch) -- a short last pass of 8 pages (of 1280x960; fewer, larger ones by pixel count),
    // because only the LAST pass's CCL + host geometry is exposed: 64 pages run as [56, 8], 16 A4@300dpi scans as [14, 2] -- when the call
    // is alone on its context.
    std::vector<int> passes;
    int sb_other = 0;
    if (c->cfg.det_sub_batch > 0) {
        for (int b0 = 0; b0 < B; b0 += c->cfg.det_sub_batch) passes.push_back(std::min(c->cfg.det_sub_batch, B - b0));
    } else {
        c->arena.begin(true);
        craft_forward_any(c, nullptr, 1, d.th, d.tw, d.H32, d.W32, nullptr);
        const size_t per_page = std::max<size_t>(c->arena.off, 1);
        const int cap = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)96 << 30) / per_page));
        static const int tail_knob = diag_knob("BBOCR_DET_TAIL", 8);   // A/B knob: tail length in 1280x960-page equivalents
        // pages larger than 1280x960 (an A4@300dpi canvas is 3.8 of them) count by their pixels: 16 A4 scans run as [14, 2]
        const double equiv = std::max(1.0, (double)d.th * d.tw / (960.0 * 1280.0));
        const int tail_pages = tail_knob > 0 ? std::max(1, (int)std::lround(tail_knob / equiv)) : 0;
        // ... unless another call is in flight on this context (bbocr_config::call_slots): its kernels fill the card while this call's
        // last pass is turned into boxes, so the short pass has nothing left to hide and only costs its own inefficiency (round 4, two calls
        // in flight: [56, 8] 942, one pass of 64 959, [32, 32] 947 images/s)
        const bool alone = !other_call_in_flight(c);
        const int tail = (alone && after_sub && B * equiv >= 24.0 && cap > tail_pages && B > tail_pages && tail_pages > 0) ? tail_pages : 0;
        const int body = B - tail, nbig = cdiv(body, cap);
        for (int i = 0; i < nbig; ++i) passes.push_back(body / nbig + (i < body % nbig ? 1 : 0));
        if (tail) passes.push_back(tail);
        sb_other = cdiv(B, cdiv(B, cap));     // largest pass of the schedule WITHOUT a tail (taken when another call is in flight)
    }
    // work buffers are sized for whichever of the two schedules has the larger pass: a context whose calls are sometimes alone and
    // sometimes not must not re-allocate a 60-GB arena when the schedule flips
    const int sb = std::max(sb_other, *std::max_element(passes.begin(), passes.end()));
    const bool need_resize = (d.th != H || d.tw != W);
    if (need_resize) c->resized.ensure((size_t)sb * d.th * d.tw * 3);
    c->arena.begin(true);
    craft_forward_any(c, nullptr, sb, d.th, d.tw, d.H32, d.W32, nullptr);
    c->arena.buf.ensure(c->arena.off);
    int b0 = 0;
    for (const int nb : passes) {
        EnqLock enq(c);                       // one detector pass = one contiguous block on the compute stream
        const uint8_t* src = rgb + (size_t)b0 * H * W * 3;
        if (need_resize) {
            HIPCHK(launch_resize_u8(src, nb, H, W, 3, (uint8_t*)c->resized.p, d.th, d.tw, c->stream));
            src = (const uint8_t*)c->resized.p;
        }
        c->arena.begin(false);
        craft_forward_any(c, src, nb, d.th, d.tw, d.H32, d.W32, heat + (size_t)b0 * d.h * d.w * 2);
        if (after_sub) after_sub(b0, nb);   // everything of this sub-batch is enqueued (nothing has been waited for)
        b0 += nb;
    }
}

// ------------------------------------------------------------------------------------------------ boxes

void boxes_impl(bbocr_ctx* c, const float* heat, int B, int h, int w, double ratio, const bbocr_params& p, HostBoxes& hb,
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
    slot_sync(c, st);
    if (counters[2]) fail(BBOCR_ERR_OVERFLOW, "component buffers too small for this batch");
    std::vector<CclOut> all_comps(counters[0]);
    std::vector<int> all_rows((size_t)counters[1] * 2);
    if (counters[0]) HIPCHK(hipMemcpyAsync(all_comps.data(), c->ccl_comps.p, all_comps.size() * sizeof(CclOut), hipMemcpyDeviceToHost, st));
    if (counters[1]) HIPCHK(hipMemcpyAsync(all_rows.data(), c->ccl_rowext.p, all_rows.size() * 4, hipMemcpyDeviceToHost, st));
    slot_sync(c, st);
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
    host_pool(c).parallel_for(B, do_page);      // the slot's persistent pool, sized from the process's CPU share (hostpool.h)
    c->times[2] += (float)ms_since(t0);
}

bbocr_boxlist* export_boxes(const HostBoxes& hb) {
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

void import_boxes(const bbocr_boxlist* bl, HostBoxes& hb) {
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
