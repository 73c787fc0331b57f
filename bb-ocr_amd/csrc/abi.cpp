// C ABI of libbbocr (include/bbocr.h): context life cycle, weights, the detect / boxes / recognise / readtext pipelines, pre-processing chain, profiling.
#include "ctx.h"

// the sequence stage's stream.  (A HIGH-PRIORITY stream here -- so that its few latency-bound workgroups would be dispatched ahead of the
// other call's thousands of conv workgroups -- was measured: the mere existence of a priority stream made every other stream of the
// process slower, CCL 1.2 -> 2.3 ms, detector 53 -> 57 ms per step, 922 -> 873 images/s; profiles/r04_inflight_ab.txt.  Plain stream.)
static hipError_t create_seq_stream(hipStream_t* s) {
    static const bool prio = (diag_knob("BBOCR_SEQ_PRIO", 0) != 0);      // A/B knob (diagnostic builds)
    int lo = 0, hi = 0;
    if (prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

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
        c->root = c;
        if (cfg) c->cfg = *cfg;
        if (c->cfg.precision < BBOCR_PREC_BF16 || c->cfg.precision > BBOCR_PREC_EXACT_REC || c->cfg.call_slots < 0 || c->cfg.call_slots > kMaxSlots) {
            delete c;                      // an unknown value must not silently mean one of the modes
            return BBOCR_ERR_ARG;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || c->cfg.device < 0 || c->cfg.device >= ndev) {
            delete c;
            return BBOCR_ERR_HIP;
        }
        if (hipSetDevice(c->cfg.device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
            create_seq_stream(&c->seq_stream) != hipSuccess ||
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

}  // extern "C"

// a further call slot of `root`: own work buffers (grown on first use), side stream and events; weights and compute stream are the root's
bbocr_ctx* slot_create(bbocr_ctx* root) {
    bbocr_ctx* s = new bbocr_ctx();
    s->root = root;
    s->cfg = root->cfg;
    static const bool own_stream = (diag_knob("BBOCR_SLOT_OWN_STREAM", 0) != 0);   // A/B knob: concurrent calls on separate compute streams
    hipError_t e = hipSuccess;
    if (own_stream) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    else s->stream = root->stream;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = create_seq_stream(&s->seq_stream);
    if (e != hipSuccess) {
        if (s->stream2) (void)hipStreamDestroy(s->stream2);
        if (own_stream && s->stream) (void)hipStreamDestroy(s->stream);
        delete s;
        fail(BBOCR_ERR_HIP, std::string("call slot: ") + hipGetErrorString(e));
    }
    s->cur = s->stream;
    return s;
}

// per-slot resources (the root is slot 0): everything but the weights and the root's compute stream
void slot_destroy(bbocr_ctx* c) {
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->seq_stream) (void)hipStreamSynchronize(c->seq_stream);
    if (c->feat_ev) (void)hipEventDestroy(c->feat_ev);
    for (hipEvent_t e : c->sub_events) (void)hipEventDestroy(e);
    if (c->det_t0) { (void)hipEventDestroy(c->det_t0); (void)hipEventDestroy(c->det_t1); }
    if (c->ccl_t0) { (void)hipEventDestroy(c->ccl_t0); (void)hipEventDestroy(c->ccl_t1); }
    if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
    if (c->seq_t1) { (void)hipEventDestroy(c->seq_t1); (void)hipEventDestroy(c->seq_t2); }
    for (auto& r : c->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&c->arena.buf, &c->heat, &c->gray, &c->resized, &c->ccl_label, &c->ccl_stat, &c->ccl_slot, &c->ccl_comps, &c->ccl_rowext,
                      &c->ccl_counters, &c->crop_desc, &c->crop_desc2, &c->crop_scratch, &c->crop_hscratch, &c->crop_wscratch, &c->crop_luts, &c->crop_hist,
                      &c->ctc_idx, &c->ctc_pmax, &c->ctc_out_idx, &c->ctc_out, &c->seq_v, &c->seq_xp, &c->seq_h, &c->seq_lin, &c->seq_logits,
                      &c->seq_tables, &c->pp_gray, &c->pp_a, &c->pp_b, &c->pp_c, &c->pp_tab, &c->ctc_probs};
    for (DevBuf* b : bufs) b->release();
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->seq_stream) (void)hipStreamDestroy(c->seq_stream);
    if (c->root != c) {
        if (c->stream && c->stream != c->root->stream) (void)hipStreamDestroy(c->stream);
        delete c;
    }
}

// stage times of the call a thread has just finished (bbocr_stage_times): per calling thread, because one context serves several
namespace {
struct ThreadTimes { const bbocr_ctx* root = nullptr; float ms[8] = {0}; };
thread_local ThreadTimes tl_times;
thread_local std::string tl_err;
}  // namespace
void publish_times(const bbocr_ctx* s) {
    tl_times.root = s->root;
    memcpy(tl_times.ms, s->times, sizeof(tl_times.ms));
}

extern "C" {

void bbocr_destroy(bbocr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dist_release(c);
    for (bbocr_ctx* s : c->slots) slot_destroy(s);
    c->slots.clear();
    free_weights(c);
    slot_destroy(c);
    if (c->zero_page) (void)hipFree(c->zero_page);
    if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* bbocr_last_error(bbocr_ctx* c) {
    if (!c) return "null context";
    std::lock_guard<std::mutex> lk(c->pool_mu);
    tl_err = c->err;                 // a copy per calling thread: another slot may fail while the caller reads
    return tl_err.c_str();
}

int bbocr_load_weights(bbocr_ctx* ctx, int which, const bbocr_tensor_desc* descs, int n) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!descs || n <= 0 || (which != 0 && which != 1)) fail(BBOCR_ERR_ARG, "bad weight descriptor table");
        if (which == 0 ? ctx->craft_loaded : ctx->crnn_loaded)
            fail(BBOCR_ERR_STATE, "this network is already loaded: the packed blocks (and the weight blob layout) are fixed per context -- create a new context");
        TensorMap tm(descs, n);
        if (which == 0) load_craft(ctx, tm);
        else load_crnn(ctx, tm);
    }, /*exclusive=*/true);
}

int bbocr_alloc_weights(bbocr_ctx* ctx, int which) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (which != 0 && which != 1) fail(BBOCR_ERR_ARG, "which: 0 = detector, 1 = recogniser");
        if (which == 0 ? ctx->craft_loaded : ctx->crnn_loaded)
            fail(BBOCR_ERR_STATE, "this network is already laid out in this context -- create a new context");
        TensorMap tm;                        // every tensor present, all zeros: lays the packed plans out, bbocr_weights_import fills them
        if (which == 0) load_craft(ctx, tm);
        else load_crnn(ctx, tm);
    }, /*exclusive=*/true);
}

int bbocr_weights_blob_size(bbocr_ctx* ctx, size_t* bytes) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!bytes) fail(BBOCR_ERR_ARG, "null pointer");
        *bytes = weights_blob_bytes(ctx);
    }, /*exclusive=*/true);
}

int bbocr_weights_export(bbocr_ctx* ctx, void* dev_blob, size_t bytes) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!dev_blob) fail(BBOCR_ERR_ARG, "null device pointer");
        weights_export(ctx, dev_blob, bytes);
    }, /*exclusive=*/true);
}

int bbocr_weights_import(bbocr_ctx* ctx, const void* dev_blob, size_t bytes) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        if (!dev_blob) fail(BBOCR_ERR_ARG, "null device pointer");
        weights_import(ctx, dev_blob, bytes);
    }, /*exclusive=*/true);
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
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        bbocr_params pp;
        bbocr_default_params(&pp);
        if (p) pp = *p;
        if (!dev_rgb || !dev_heat_out) fail(BBOCR_ERR_ARG, "null device pointer");
        memset(ctx->times, 0, sizeof(ctx->times));
        auto t0 = clk::now();
        detect_impl(ctx, dev_rgb, B, H, W, pp, dev_heat_out);
        slot_sync(ctx, ctx->stream);
        prof_collect(ctx);
        ctx->times[0] = ctx->times[7] = (float)ms_since(t0);
    });
}

int bbocr_boxes(bbocr_ctx* ctx, const float* dev_heat, int B, int h, int w, double ratio, const bbocr_params* p, bbocr_boxlist** out) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
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
    return guarded(ctx, [&](bbocr_ctx* ctx) {
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
    return guarded(ctx, [&](bbocr_ctx* ctx) {
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
        static const bool early_on = (diag_knob("BBOCR_REC_EARLY", 1) != 0);   // A/B knob
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

void bbocr_preproc_defaults(bbocr_preproc_params* p, int legacy) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->scale = 1.5; p->unsharp_radius = 1.0; p->unsharp_threshold = 3;
    if (legacy) { p->blur_sigma = 5.0; p->contrast = 1.3; p->brightness = 0.0; p->clahe_clip = 2.0; p->unsharp_percent = 20; }
    else        { p->blur_sigma = 3.0; p->contrast = 1.9; p->brightness = 1.2; p->clahe_clip = 2.5; p->unsharp_percent = 30; }
}

int bbocr_preprocess_chain(bbocr_ctx* ctx, const uint8_t* dev_bgr, int H, int W, const bbocr_preproc_params* p, uint8_t* dev_out, int* out_h,
                           int* out_w) {
    return guarded(ctx, [&](bbocr_ctx* ctx) {
        bbocr_preproc_params q;
        bbocr_preproc_defaults(&q, 0);
        if (p) q = *p;
        if (H <= 0 || W <= 0) fail(BBOCR_ERR_ARG, "bad image shape");
        if (q.scale < 0 || q.blur_sigma < 0 || q.contrast < 0 || q.brightness < 0 || q.clahe_clip < 0 || q.unsharp_radius < 0 || q.unsharp_percent < 0)
            fail(BBOCR_ERR_ARG, "negative pre-processing parameter");
        const int dh = q.scale > 0 ? (int)(H * q.scale) : H, dw = q.scale > 0 ? (int)(W * q.scale) : W;
        if (out_h) *out_h = dh;
        if (out_w) *out_w = dw;
        if (!dev_out) return;
        if (!dev_bgr) fail(BBOCR_ERR_ARG, "null device pointer");
        if (dh <= 0 || dw <= 0) fail(BBOCR_ERR_ARG, "image collapses to zero size");
        if (q.clahe_clip > 0 && (dh < 16 || dw < 16)) fail(BBOCR_ERR_ARG, "image too small for the 8x8 CLAHE tile grid");
        preprocess_chain_impl(ctx, dev_bgr, H, W, q, dev_out, dh, dw);
    });
}

int bbocr_preprocess_book_cover(bbocr_ctx* ctx, const uint8_t* dev_bgr, int H, int W, uint8_t* dev_out, int* out_h, int* out_w) {
    bbocr_preproc_params q;
    bbocr_preproc_defaults(&q, 0);
    return bbocr_preprocess_chain(ctx, dev_bgr, H, W, &q, dev_out, out_h, out_w);
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
    return guarded(ctx, [&](bbocr_ctx* c) {
        c->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
        std::lock_guard<std::mutex> lk(c->pool_mu);
        for (int g = 0; g < 2; ++g) { c->prof_ms[g] = 0; c->prof_flops[g] = 0; c->prof_launches[g] = 0; }
    }, /*exclusive=*/true);
}

int bbocr_conv_profile(bbocr_ctx* ctx, int group, double* ms, double* flops, long long* launches) {
    if (!ctx || group < 0 || group > 1) return BBOCR_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    if (ms) *ms = ctx->prof_ms[group];
    if (flops) *flops = ctx->prof_flops[group];
    if (launches) *launches = ctx->prof_launches[group];
    return BBOCR_OK;
}

int bbocr_stage_times(bbocr_ctx* ctx, float* ms, int n) {
    if (!ctx || !ms || n <= 0) return BBOCR_ERR_ARG;
    for (int i = 0; i < n && i < 8; ++i) ms[i] = tl_times.root == ctx ? tl_times.ms[i] : 0.f;
    return BBOCR_OK;
}

// ---------------------------------------------------------------------------------------- host-only geometry (no GPU needed)

}  // extern "C"
