// Recogniser orchestration: crops (utils.py::get_image_list), AlignCollate, CRNN forward, CTC, contrast retry, rotation variants (easyocr recognition.py::get_text).
#include "ctx.h"

// AlignCollate / get_image_list geometry of one horizontal box; false = skipped (degenerate)
bool plan_horizontal(const std::array<int, 4>& box, int img, int H, int W, BoxJob& j) {
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

bool plan_free(const std::array<double, 8>& fq, int img, BoxJob& j) {
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

// tensor mode of the recogniser (kernels.h REC_*) and stored 16-bit elements per logical channel
int rec_mode(const bbocr_ctx* c) { return rec_split(c) ? REC_SPLIT : (rec_el(c) ? REC_F16 : REC_BF16); }
static inline int rec_mul(const bbocr_ctx* c) { return rec_split(c) ? 2 : 1; }

// conv stack of the recogniser for n normalised crops of one padded width: 16-bit [n,64,imgW] -> v [n*T, 256] (x rec_mul)
void crnn_features(bbocr_ctx* c, const uint16_t* crops, int n, int imgW, uint16_t* v_out) {
    Arena& ar = c->arena;
    c->prof_group = 1;
    const int T = imgW / 4 - 1;
    const int m = rec_mul(c);
    Act c0{ar.alloc<uint16_t>((size_t)n * 32 * (imgW / 2) * 32 * m), n, 32, imgW / 2, 32 * m};
    if (!ar.dry) HIPCHK(launch_crnn_conv0(crops, c->r0_wb, c->r0_wb + 288, c0.p, n, imgW, rec_mode(c), c->cur, c->r0_afrag));
    Act q1 = conv_pool_act(c, c->r1, c0, false, true, 64, 1, false, nullptr);
    Act c2 = conv_act(c, c->r2, q1, false, nullptr, false, true, 128);
    Act q2 = conv_pool_act(c, c->r3, c2, false, true, 128, 2, false, nullptr);
    Act c4 = conv_act(c, c->r4, q2, false, nullptr, false, true, 256);
    Act q3 = conv_pool_act(c, c->r5, c4, false, true, 256, 2, false, nullptr);
    Act c6 = conv_act(c, c->r6, q3, false, nullptr, false, true, 256);   // [n,3,T,256]
    if (!ar.dry) HIPCHK(launch_rowmean3(c6.p, v_out, n, T, 256, rec_mode(c), c->cur));
}

// The same conv stack over the WIDE image of a recognition pass: every crop side by side with 4 zero columns between
// neighbours (CropDesc::slot = first column), so each layer is ONE launch over [H, Wt] whatever the mix of width buckets.  The
// separator columns are each layer's zero padding; convolutions write into them, so they are cleared on every layer output
// (4 >> shift columns per crop).  The 3-row mean is gathered straight into every crop's pooled rows (CropDesc::pad_).
static void crnn_features_wide(bbocr_ctx* c, const uint16_t* wide, int Wt, const CropDesc* descs, int first, int count, uint16_t* seq_v) {
    Arena& ar = c->arena;
    c->prof_group = 1;
    const int m = rec_mul(c);
    Act c0{ar.alloc<uint16_t>((size_t)32 * (Wt / 2) * 32 * m), 1, 32, Wt / 2, 32 * m};
    if (!ar.dry) HIPCHK(launch_crnn_conv0(wide, c->r0_wb, c->r0_wb + 288, c0.p, 1, Wt, rec_mode(c), c->cur, c->r0_afrag));
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
    if (!ar.dry) HIPCHK(launch_rowmean3_gather(c6.p, c6.W, 256, descs, first, count, seq_v, rec_mode(c), c->cur));
}

// Sequence half of the recogniser over the POOLED time steps of every bucket (rows = sum n_i*T_i, padded to x256):
// v bf16 [rows,256] (ctx->seq_v) -> logits fp32 [rows,112].  The two linear layers and both input projections are
// single GEMMs over all rows; each BiLSTM layer is ONE launch whose workgroups carry their own sequence length.
void crnn_sequence(bbocr_ctx* c, size_t rows_pad, const int* tiles_dev, int ntiles, float* logits) {
    c->prof_group = 1;
    c->arena.dry = false;
    const bool sp = rec_split(c);
    const int m = rec_mul(c);
    c->seq_xp.ensure(rows_pad * 2048 * (sp ? 4 : 2));            // exact mode: the input projection stays fp32
    c->seq_h.ensure(rows_pad * 512 * 2 * m);
    c->seq_lin.ensure(rows_pad * 256 * 2 * m);
    const int Hh = (int)(rows_pad / 256);
    Act cur{(uint16_t*)c->seq_v.p, 1, Hh, 256, 256 * m};
    for (int l = 0; l < 2; ++l) {
        run_conv(c, c->xproj[l], cur, false, nullptr, false, false, c->seq_xp.p, 2048, 2048, sp);
        HIPCHK(launch_lstm(c->seq_xp.p, c->whh[l], (uint16_t*)c->seq_h.p, tiles_dev, ntiles, rec_mode(c), c->whh_scale[l], c->cur));
        Act hh{(uint16_t*)c->seq_h.p, 1, Hh, 256, 512 * m};
        uint16_t* dst = (uint16_t*)(l == 0 ? c->seq_lin.p : c->seq_v.p);
        run_conv(c, c->lin[l], hh, false, nullptr, false, false, dst, 256 * m, 256, false);
        cur.p = dst;
    }
    run_conv(c, c->pred, cur, false, nullptr, false, false, logits, 112, 112, true);
}

// A recognition pass = feature PARTS + one sequence stage.  A part is a set of crops standing side by side in ONE wide image
// (CropDesc::slot = first column, 4 zero columns after every crop; ::pad_ = first pooled row), so each CRNN conv layer is a single
// launch per part whatever the mix of widths; every part gathers its pooled time steps into the pass's shared [rows,256] tensor,
// and the sequence stage (input projections, both BiLSTM layers, linear layers, class projection, CTC) then runs ONCE over all rows.
// readtext_batch uses two parts: the crops of the first detector pass's pages go through the conv stack while the last pass's boxes
// are still being extracted (CCL + host geometry), so that stretch no longer leaves the card idle.
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
    EnqLock enq(c);                           // one feature part = one contiguous block on the compute stream
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
                                (uint8_t*)c->crop_hscratch.p, (const uint8_t*)c->crop_luts.p, wide, 2, c->stream, Wt, REC_GAP, rec_mode(c)));
        crnn_features_wide(c, wide, Wt, dd, 0, n, (uint16_t*)c->seq_v.p);
        if (pass == 0) c->arena.buf.ensure(c->arena.off);
    }
    if (!c->feat_ev) HIPCHK(hipEventCreateWithFlags(&c->feat_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->feat_ev, c->stream));      // what the sequence stage (on seq_stream) waits for
}

static void rec_add_tables(RecRun& run, const RecPart& part, int tile_seqs) {
    for (const RecChunk& ch : part.chunks) {
        for (int s0 = 0; s0 < ch.n; s0 += tile_seqs) {
            run.tiles.push_back((int)(ch.row0 + (size_t)s0 * ch.T));
            run.tiles.push_back(std::min(tile_seqs, ch.n - s0));
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
    c->ctc_idx.ensure(rows * 4);
    c->ctc_pmax.ensure(rows * 4);
    c->ctc_out_idx.ensure(rows * 4);
    c->ctc_out.ensure((size_t)nseq * sizeof(CtcOut));
    const bool beam = c->beam_width > 0;
    if (beam) c->ctc_probs.ensure(rows * 112 * sizeof(float));
    const size_t oo_off = align_up(rows * 4, 16);
    c->ctc_pin.ensure(oo_off + (size_t)nseq * sizeof(CtcOut));
    const int* oidx = (const int*)c->ctc_pin.p;
    const CtcOut* oo = (const CtcOut*)((const char*)c->ctc_pin.p + oo_off);
    std::vector<float> probs(beam ? rows * 112 : 0);
    std::vector<std::vector<int>> beam_texts;
    if (!c->seq_t1) { HIPCHK(hipEventCreate(&c->seq_t1)); HIPCHK(hipEventCreate(&c->seq_t2)); }
    {
        // Sequence stage, CTC and the read-back of its results: one block, ONE host wait.  It runs on the slot's own seq_stream behind the
        // event of this slot's feature parts: its launches are latency chains of 100-300 workgroups (BiLSTM: 4.9 us per time step against
        // 1 us of MFMA work) that leave most of the card idle -- with a second call in flight that call's detector fills it, instead of
        // waiting in (or making this stage wait in) the compute stream's FIFO.
        static const bool own = (diag_knob("BBOCR_SEQ_STREAM", 1) != 0);      // A/B knob
        hipStream_t ss = (own && c->seq_stream && c->feat_ev) ? c->seq_stream : c->stream;
        std::unique_lock<std::mutex> enq;
        if (ss == c->stream) enq = std::unique_lock<std::mutex>(c->root->enq_mu);
        else HIPCHK(hipStreamWaitEvent(ss, c->feat_ev, 0));
        c->cur = ss;
        struct Restore { bbocr_ctx* c; ~Restore() { c->cur = c->stream; } } restore{c};
        HIPCHK(hipMemcpyAsync(tiles_dev, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, ss));
        HIPCHK(hipMemcpyAsync(seqs_dev, seqs.data(), seqs.size() * 4, hipMemcpyHostToDevice, ss));
        crnn_sequence(c, rows_pad, tiles_dev, ntiles, (float*)c->seq_logits.p);
        HIPCHK(hipEventRecord(c->seq_t1, ss));
        HIPCHK(launch_ctc((const float*)c->seq_logits.p, rows, 97, 112, seqs_dev, nseq, (int*)c->ctc_idx.p, (float*)c->ctc_pmax.p,
                          (int*)c->ctc_out_idx.p, (CtcOut*)c->ctc_out.p, ss, c->ignore_mask, beam ? (float*)c->ctc_probs.p : nullptr));
        HIPCHK(hipMemcpyAsync(c->ctc_pin.p, c->ctc_out_idx.p, rows * 4, hipMemcpyDeviceToHost, ss));
        HIPCHK(hipMemcpyAsync((char*)c->ctc_pin.p + oo_off, c->ctc_out.p, (size_t)nseq * sizeof(CtcOut), hipMemcpyDeviceToHost, ss));
        if (beam) HIPCHK(hipMemcpyAsync(probs.data(), c->ctc_probs.p, probs.size() * sizeof(float), hipMemcpyDeviceToHost, ss));
        HIPCHK(hipEventRecord(c->seq_t2, ss));
    }
    HIPCHK(hipEventSynchronize(c->seq_t2));
    float ctc_ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ctc_ms, c->seq_t1, c->seq_t2));
    c->times[4] += (float)ms_since(t0) - ctc_ms;     // host wait for the recogniser's device work (conv stack of the parts + sequence stage)
    c->times[5] += ctc_ms;                           // CTC kernels + read-back (device span) ...
    t0 = clk::now();                                 // ... + the host decode below
    if (beam) ctc_beam_search_batch(probs.data(), seqs.data(), nseq, 97, 112, c->beam_width, beam_texts, &host_pool(c));   // the confidence stays the greedy path's
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
        c->seq_v.ensure(align_up(part.rows, 256) * 256 * 2 * rec_mul(c));
        rec_launch_part(c, gray, H, W, part, c->crop_desc, stage_a);
        c->times[3] += (float)ms_since(t0);
        RecRun run;
        rec_add_tables(run, part, lstm_tile_seqs(rec_mode(c)));
        rec_finish(c, run, texts, confs);
    }
}

// np.percentile(img, q) (method 'linear') from a 256-bin histogram of n uint8 samples
double percentile_u8(const unsigned int* hist, size_t n, double q) {
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
void rec_early_begin(bbocr_ctx* c, const uint8_t* gray, int pages, int B, int H, int W, const HostBoxes& hb, const bbocr_params& p,
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
    c->seq_v.ensure(align_up((size_t)((double)e.part.rows * grow), 256) * 256 * 2 * rec_mul(c));
    auto t0 = clk::now();
    rec_launch_part(c, gray, H, W, e.part, c->crop_desc, true);
    c->times[3] += (float)ms_since(t0);
    e.pages = pages;
    e.active = true;
}

void recognize_impl(bbocr_ctx* c, const uint8_t* gray, int B, int H, int W, const HostBoxes& hb, const bbocr_params& p,
                           std::vector<BoxJob>& jobs, std::vector<int>& box_off, RecEarly* early) {
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
        rec_add_tables(run, early->part, lstm_tile_seqs(rec_mode(c)));
        if (early->part.rows + part2.rows <= rec_max_rows(c)) {
            c->seq_v.ensure_keep(align_up(early->part.rows + part2.rows, 256) * 256 * 2 * rec_mul(c), early->part.rows * 256 * 2 * rec_mul(c));
            auto t0 = clk::now();
            rec_launch_part(c, gray, H, W, part2, c->crop_desc2, true);
            c->times[3] += (float)ms_since(t0);
            rec_add_tables(run, part2, lstm_tile_seqs(rec_mode(c)));
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
    slot_sync(c, c->stream);
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

bbocr_result* export_result(int B, const std::vector<BoxJob>& jobs, const std::vector<int>& box_off) {
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
