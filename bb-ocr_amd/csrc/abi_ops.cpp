// C ABI of libbbocr (include/bbocr.h), stage-level entry points: single kernels and host stages exposed so that the parity tests can pin each one.
#include "ctx.h"

extern "C" {

int bbocr_op_preprocess_stage(bbocr_ctx* ctx, int stage, const uint8_t* dev_src, int H, int W, uint8_t* dev_dst, int dh, int dw, double param) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!dev_src || !dev_dst || H <= 0 || W <= 0 || dh <= 0 || dw <= 0) fail(BBOCR_ERR_ARG, "bad arguments");
        if (stage != 0 && (dh != H || dw != W)) fail(BBOCR_ERR_ARG, "only stage 0 changes the size");
        const size_t n = (size_t)H * W;
        if (stage == 0) {
            pp_resize(ctx, dev_src, H, W, dev_dst, dh, dw);
        } else if (stage == 1) {
            pp_gauss(ctx, dev_src, H, W, dev_dst, param);
        } else if (stage == 2 || stage == 3) {
            // pointwise PIL enhancers: the chain folds them into CLAHE's input LUT; stand-alone they are one lookup pass
            if (stage == 2) {
                ctx->pp_a.ensure(n);
                // mean of the input: the 3x3 smoothing kernel with taps (0, 256, 0) is the identity and sums its output
                pp_gauss(ctx, dev_src, H, W, (uint8_t*)ctx->pp_a.p, 0.0);
            }
            HIPCHK(launch_pp_lut(dev_src, dev_dst, pp_fold_lut(ctx, n, stage == 2 ? param : 0.0, stage == 3 ? param : 0.0), n, ctx->stream));
        } else if (stage == 4) {
            pp_clahe(ctx, dev_src, H, W, pp_fold_lut(ctx, n, 0.0, 0.0), dev_dst, param);
        } else if (stage == 5) {
            ctx->pp_b.ensure(n);
            ctx->pp_c.ensure(n);
            pp_unsharp(ctx, dev_src, H, W, dev_dst, (uint8_t*)ctx->pp_b.p, (uint8_t*)ctx->pp_c.p, (float)param, 30, 3);
        } else if (stage == 7) {
            ctx->pp_b.ensure(n);
            ctx->pp_c.ensure(n);
            pp_unsharp(ctx, dev_src, H, W, dev_dst, (uint8_t*)ctx->pp_b.p, (uint8_t*)ctx->pp_c.p, 1.0f, (int)param, 3);
        } else if (stage == 6) {
            // cv2.cvtColor(BGR2GRAY) on an interleaved 3-channel plane [H,W,3] (the gray plane reformat_input derives from arrays)
            HIPCHK(launch_gray(dev_src, dev_dst, n, ctx->stream));
        } else {
            fail(BBOCR_ERR_ARG, "unknown pre-processing stage");
        }
        slot_sync(ctx, ctx->stream);                            // every stage only enqueues
    });
}

int bbocr_host_cpu_share(void) { return host_cpu_share(); }

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
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!dev_in || !w || Cin <= 0 || (Cin & 31) || Cout <= 0 || KH <= 0 || KW <= 0) fail(BBOCR_ERR_ARG, "bad conv arguments");
        if (pool_mode < 0 || pool_mode > 2 || (pool_mode ? (!dev_pool_out || out_f32) : !dev_out)) fail(BBOCR_ERR_ARG, "bad conv output arguments");
        ConvPlan p = make_plan(Cin, Cout, KH, KW, pad, dil, det_el(ctx));     // element type of the context's precision (bf16 / fp16)
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
        while (ctx->owned.size() > owned0) { (void)hipFree(ctx->owned.back()); ctx->owned.pop_back(); ctx->owned_bytes.pop_back(); }
        HIPCHK(e);
        HIPCHK(e2);
    }, /*exclusive=*/true);      // uploads a temporary plan into the root's weight list
}

int bbocr_crnn_logits(bbocr_ctx* ctx, const uint16_t* dev_crops, int n, int imgW, float* dev_logits) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!ctx->crnn_loaded) fail(BBOCR_ERR_STATE, "recogniser weights not loaded");
        if (!dev_crops || !dev_logits || n <= 0 || imgW < 64 || (imgW & 63)) fail(BBOCR_ERR_ARG, "bad crop batch");
        const int T = imgW / 4 - 1;
        const size_t rows = (size_t)n * T, rows_pad = align_up(rows, 256);
        ctx->seq_v.ensure(rows_pad * 256 * 2 * (rec_split(ctx) ? 2 : 1));
        ctx->seq_logits.ensure(rows_pad * 112 * 4);
        ctx->arena.begin(true);
        crnn_features(ctx, dev_crops, n, imgW, nullptr);
        ctx->arena.buf.ensure(ctx->arena.off);
        ctx->arena.begin(false);
        crnn_features(ctx, dev_crops, n, imgW, (uint16_t*)ctx->seq_v.p);
        std::vector<int> tiles;
        const int ts = lstm_tile_seqs(rec_mode(ctx));
        for (int s0 = 0; s0 < n; s0 += ts) {
            tiles.push_back(s0 * T); tiles.push_back(std::min(ts, n - s0)); tiles.push_back(T); tiles.push_back(0);
        }
        ctx->seq_tables.ensure(tiles.size() * 4);
        HIPCHK(hipMemcpyAsync(ctx->seq_tables.p, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        crnn_sequence(ctx, rows_pad, (const int*)ctx->seq_tables.p, (int)(tiles.size() / 4), (float*)ctx->seq_logits.p);
        HIPCHK(hipMemcpyAsync(dev_logits, ctx->seq_logits.p, rows * 112 * 4, hipMemcpyDeviceToDevice, ctx->stream));
        slot_sync(ctx, ctx->stream);
    });
}

int bbocr_op_ctc(bbocr_ctx* ctx, const float* dev_logits, int n, int T, int C, int cs, int* text_off, int* text_idx, double* conf,
                 const unsigned int* ignore_mask, int beam_width) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
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
        slot_sync(ctx, ctx->stream);
        if (beam) ctc_beam_search_batch(probs.data(), seqs.data(), n, C, cs, beam_width, beam_texts, &host_pool(ctx));
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
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!dev_src || !dev_dst || N <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || C <= 0) fail(BBOCR_ERR_ARG, "bad resize arguments");
        HIPCHK(launch_resize_u8(dev_src, N, sh, sw, C, dev_dst, dh, dw, ctx->stream));
        slot_sync(ctx, ctx->stream);
    });
}

int bbocr_op_ycc_to_rgb(bbocr_ctx* ctx, const uint8_t* dev_ycc, size_t npix, int pixel_stride, uint8_t* dev_rgb, uint8_t* dev_gray) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!dev_ycc || !dev_rgb || npix == 0 || (pixel_stride != 3 && pixel_stride != 4)) fail(BBOCR_ERR_ARG, "bad colour-conversion arguments");
        HIPCHK(launch_ycc_to_rgb_gray(dev_ycc, pixel_stride, dev_rgb, dev_gray, npix, ctx->stream));
        slot_sync(ctx, ctx->stream);
    });
}

int bbocr_upload_pages(bbocr_ctx* root, const void* const* host_pages, int n, size_t bytes_each, void* dev_dst) {
    // NOT a call slot's work: an upload stage feeding two calls in flight must not wait for one of them to return.  Own stream, one upload at
    // a time per context; errors reported like every other entry point's.
    if (!root) return BBOCR_ERR_ARG;
    auto set_err = [&](const std::string& m) {
        std::lock_guard<std::mutex> lk(root->pool_mu);
        root->err = m;
    };
    try {
        if (!host_pages || !dev_dst || n <= 0 || bytes_each == 0) fail(BBOCR_ERR_ARG, "bad upload arguments");
        HIPCHK(hipSetDevice(root->cfg.device));
        std::lock_guard<std::mutex> up(root->upload_mu);
        if (!root->upload_stream) HIPCHK(hipStreamCreateWithFlags(&root->upload_stream, hipStreamNonBlocking));
        for (int k = 0; k < n; ++k) {
            if (!host_pages[k]) fail(BBOCR_ERR_ARG, "null page");
            HIPCHK(hipMemcpyAsync((unsigned char*)dev_dst + (size_t)k * bytes_each, host_pages[k], bytes_each, hipMemcpyHostToDevice, root->upload_stream));
        }
        HIPCHK(hipStreamSynchronize(root->upload_stream));
        return BBOCR_OK;
    } catch (const StatusError& se) {
        set_err(se.msg);
        return se.code;
    } catch (const std::exception& ex) {
        set_err(ex.what());
        return BBOCR_ERR_INTERNAL;
    }
}

int bbocr_op_crops(bbocr_ctx* ctx, const uint8_t* dev_gray, int H, int W, const int* hori, int n_hori, const double* free_q, int n_free, int imgW,
                   float contrast, uint16_t* dev_out, int* n_out, int mode) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
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
            slot_sync(ctx, ctx->stream);
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
                            (const uint8_t*)ctx->crop_luts.p, dev_out, 2, ctx->stream, 0, 0, rec_mode(ctx)));
        slot_sync(ctx, ctx->stream);
    });
}

}  // extern "C"
