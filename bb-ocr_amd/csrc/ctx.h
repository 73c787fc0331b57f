// Internal header of libbbocr: context, small host helpers and the functions the translation units share.
// Translation units: weights.cpp (BN folding, MFMA packing), detector.cpp (CRAFT forward, box extraction), recognizer.cpp (crops,
// CRNN, CTC), preprocess.cpp (the f2 chain), abi.cpp (context + pipeline entry points of include/bbocr.h), abi_ops.cpp (stage-level
// entry points used by the parity tests).
#pragma once
#include "../../include/bbocr.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "boxpost.h"
#include "hostpool.h"
#include "common.h"
#include "kernels.h"


using clk = std::chrono::steady_clock;
static inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct StatusError {
    int code;
    std::string msg;
};
[[noreturn]] inline void fail(int code, const std::string& m) { throw StatusError{code, m}; }
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



// Everything bbocr_load_weights leaves on the device, as the kernels read it.  A context's extra CALL SLOTS (below) hold a copy of this
// view -- pointers into the root context's blocks, which alone owns (and frees) them.
struct WeightView {
    // ---- detector
    bool craft_loaded = false;
    uint16_t* c11_wf = nullptr;               // conv1_1 weights in the layout of the producer fused into conv1_2
    float* c11_b = nullptr;
    float* c11_w32 = nullptr;                 // exact mode: conv1_1 folded fp32 [64][3][3][3] (craft_pair.hip::pair_conv1_1_kernel)
    float* cls6_w32 = nullptr;                // exact mode: conv_cls.6.weight fp32 [16][16] (pair_cls_tail_kernel)
    ConvPlan conv1_2, conv2_1, conv2_2, conv3_1, conv3_2, conv3_3, conv4_1, conv4_2, conv4_3, conv5_1, conv5_2, fc6, fc7;
    ConvPlan up1a, up1b, up2b, up3b, up4b, cls0, cls2, cls4;
    // U-net 1x1 layers over cat[up(y), skip], split by linearity: upNy = the columns of y (no bias, run at y's resolution),
    // upNs = the columns of the skip tensor (+ bias), whose epilogue adds the 2x bilinear up-sampling of upNy's output
    ConvPlan up2y, up2s, up3y, up3s, up4y, up4s;
    float* cls_tail = nullptr;   // b1[16] w2[32] b2[2]
    uint16_t* cls_tail_frag = nullptr;   // conv_cls.6 weight as an MFMA A fragment
    uint16_t* up4y_post = nullptr;       // upconv4's 1x1, y half (64 -> 64), as the A fragments of the 1x1 applied in upconv3.3x3's epilogue
    // ---- recogniser
    bool crnn_loaded = false;
    float* r0_wb = nullptr;      // w[9][32] (tap-major) b[32]
    uint16_t* r0_afrag = nullptr;  // taps 0..7 three-way split into element-type terms, as MFMA A fragments (crnn_conv0_mfma_kernel)
    ConvPlan r1, r2, r3, r4, r5, r6, xproj[2], lin[2], pred;
    uint16_t* whh[2] = {nullptr, nullptr};
    float whh_scale[2] = {1.f, 1.f};   // exact mode: 2^-s of the packed W_hh (pack_lstm_whh_split)
    void* zero_page = nullptr;   // 256 zero bytes (padding source of the LDS-DMA conv variant)
};

// One bbocr_ctx = the ROOT (what bbocr_create returns: configuration, weights, the compute stream, the call-slot pool) and, at the
// same time, call slot 0.  The reference shares ONE Reader between ThreadPoolExecutor workers (batch_processor_enhanced.py:215,
// default 2): a second call arriving while one is in flight takes a second slot -- a context of the same type that owns its own work
// buffers, side stream and events, shares the root's weights (WeightView copy) and queues its kernels on the SAME compute stream, so
// the card executes the two calls' kernels in issue order with no gap: call B's detector runs while call A's host thread is busy
// with the last pass's box geometry, CTC read-back and result export (the ~8 % of a step that left the card idle).
struct bbocr_ctx : WeightView {
    bbocr_config cfg{};
    bbocr_ctx* root = nullptr;                // slot -> its root; the root points at itself
    std::vector<bbocr_ctx*> slots;            // root only: extra call slots, created when a second call arrives while the first is running
    std::mutex pool_mu;                       // root only: guards slot_busy / slots / err / prof totals / last_times
    std::condition_variable pool_cv;
    std::mutex enq_mu;                        // root only: held while a slot queues ONE block of launches (a detector pass, a recogniser part, a
                                              // sequence stage) on the shared compute stream, so that blocks of concurrent calls do not interleave kernel
                                              // by kernel (the per-launch HIP events of the roofline leg would then time foreign kernels too)
    bool slot_busy = false;                   // this slot is running a call
    hipStream_t stream = nullptr;             // the compute stream (root's; slots share it: kernels of concurrent calls run in issue order)
    DevBuf pp_gray, pp_a, pp_b, pp_c, pp_tab;  // pre-processing chain (f2): planes and small tables
    DevBuf pp_cubic;                           // cubic-resize weight tables, kept on the device while (W, dw, H, dh) repeats
    int pp_cubic_key[4] = {0, 0, 0, 0};
    unsigned long long pp_cubic_K[2] = {0, 0};
    unsigned int ignore_mask[4] = {0, 0, 0, 0};   // recogniser class mask of the running call (bbocr_params::ignore_mask)
    int beam_width = 0;                           // > 0: decoder='beamsearch' for the running call (bbocr_params::decoder / beam_width)
    hipStream_t cur = nullptr;                // stream the layer helpers launch on
    hipStream_t upload_stream = nullptr;      // root only: bbocr_upload_pages copies here, outside the call slots (an upload never waits for a running call)
    std::mutex upload_mu;
    hipStream_t seq_stream = nullptr;         // per slot: the recogniser's SEQUENCE stage (projections, BiLSTMs, linears, CTC, read-back) -- latency-bound
                                              // launches of 100-300 workgroups that leave most of the card idle -- runs here, behind an event of this
                                              // slot's feature parts, so that it overlaps the OTHER call's detector instead of queueing in front of it
    hipEvent_t feat_ev = nullptr;             // end of this slot's latest feature part on the compute stream
    hipStream_t stream2 = nullptr;            // box extraction of detector sub-batch k runs here while sub-batch k+1 is on `stream` (per slot)
    std::vector<hipEvent_t> sub_events;       // one per detector sub-batch of a readtext_batch call
    hipEvent_t ccl_t0 = nullptr, ccl_t1 = nullptr;   // GPU span of the CCL kernels of one boxes_impl call
    hipEvent_t det_t0 = nullptr, det_t1 = nullptr;   // detector span on `stream` (the host is busy with boxes meanwhile)
    hipEvent_t sync_ev = nullptr;             // "this slot's work so far" on a stream (slot_sync)
    hipEvent_t seq_t1 = nullptr, seq_t2 = nullptr;   // end of the sequence stage / of CTC + read-back (rec_finish)
    std::string err;                          // root: message of the last failed call (any slot)
    float times[8] = {0};                     // the running call's stage times; published per calling thread when the call ends

    // ---- optional per-launch timing of the conv_mfma kernel (HIP events on the compute stream)
    struct ProfRec { hipEvent_t e0, e1; double flops; int group; };
    int profiling = 0;                        // root: 0 off, 1 = time the detector's conv launches (group 0), 2 = also the recogniser's
    int prof_group = 0;                     // 0 = detector, 1 = recogniser
    std::vector<ProfRec> prof_recs;           // per slot: launches recorded by the running call
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[2] = {0, 0}, prof_flops[2] = {0, 0};   // root: totals
    long long prof_launches[2] = {0, 0};

    void* dist_comm = nullptr;                // root: RCCL communicator of the bbocr_dist_* entry points (dist.cpp), or null
    int dist_rank = 0, dist_world = 0;
    std::unique_ptr<HostPool> pool;           // per slot, made on first use: host threads for box geometry / beam search (hostpool.h)
    std::vector<void*> owned;    // root: every hipMalloc'd weight block, in load order (the order of the weight blob, bbocr_weights_export)
    std::vector<size_t> owned_bytes;

    Arena arena;
    DevBuf heat, gray, resized;
    DevBuf ccl_label, ccl_stat, ccl_slot, ccl_comps, ccl_rowext, ccl_counters;
    DevBuf crop_desc, crop_scratch, crop_hscratch, crop_wscratch, crop_luts, crop_hist;
    DevBuf ctc_idx, ctc_pmax, ctc_out_idx, ctc_out, ctc_probs, crop_desc2;
    PinBuf desc_pin, desc_pin2;               // staging of crop_desc / crop_desc2 uploads
    PinBuf ctc_pin;                           // CTC results land here (pinned: the 1.7 MB D2H copy of a 64-page pass runs at link speed)
    DevBuf seq_v, seq_xp, seq_h, seq_lin, seq_logits, seq_tables;
};

// element type of a network's MFMA operands / stored activations (El<> in common.h), from bbocr_config::precision
inline int det_el(const bbocr_ctx* c) { return (c->cfg.precision == BBOCR_PREC_FP16 || c->cfg.precision == BBOCR_PREC_EXACT || c->cfg.precision == BBOCR_PREC_EXACT_REC) ? 1 : 0; }
inline int rec_el(const bbocr_ctx* c) { return c->cfg.precision != BBOCR_PREC_BF16 ? 1 : 0; }     // MIXED: bf16 detector, fp16 recogniser
inline bool rec_split(const bbocr_ctx* c) { return c->cfg.precision == BBOCR_PREC_EXACT || c->cfg.precision == BBOCR_PREC_EXACT_REC; }   // recogniser tensors are [hi | lo] fp16 pairs
inline bool det_split(const bbocr_ctx* c) { return c->cfg.precision == BBOCR_PREC_EXACT; }   // and so are the detector's (split-fp16 plans in every layer)

// ------------------------------------------------------------------------------------------------ shared types
struct TensorMap {
    std::unordered_map<std::string, const bbocr_tensor_desc*> m;
    bool zeros = false;                       // allocation-only load (bbocr_alloc_weights): every tensor exists and is all zeros
    std::vector<float> zero_buf;              // sized once (pointers handed out stay valid): the largest tensor is fc6, 1024 x 512 x 9
    TensorMap() : zeros(true), zero_buf((size_t)5 << 20, 0.f) {}
    TensorMap(const bbocr_tensor_desc* d, int n) {
        for (int i = 0; i < n; ++i) {
            std::string k = d[i].name ? d[i].name : "";
            if (k.rfind("module.", 0) == 0) k = k.substr(7);
            m[k] = &d[i];
        }
    }
    const float* get(const std::string& name, size_t numel, bool required = true) const {
        if (zeros) {
            if (zero_buf.size() < numel) fail(BBOCR_ERR_INTERNAL, "allocation-only load: tensor '" + name + "' larger than the zero buffer");
            return zero_buf.data();
        }
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

struct RgbSource { const uint8_t* rgb; int Himg, Wimg; };   // conv1_2 with conv1_1 fused in: a0 then only carries the canvas shape

struct DetDims {
    int H32, W32, h, w, th, tw;
    double ratio;
};

struct HostBoxes {
    std::vector<std::vector<std::array<int, 8>>> polys;
    std::vector<std::vector<std::array<int, 4>>> hori;
    std::vector<std::vector<std::array<double, 8>>> freeb;
};

struct BoxJob {            // one box to recognise
    int img;
    bool is_free;
    double quad[8];        // reported corners
    CropDesc d;
    std::vector<int> text;
    double conf = 0.0;
};

struct RecChunk {
    int imgW, T, first, n;   // descriptors [first, first+n) share the padded width imgW
    size_t row0;             // first pooled row of the chunk in the pass's sequence tensors
};

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

struct RecEarly {
    bool active = false;
    int pages = 0;
    std::vector<BoxJob> jobs;
    std::vector<int> box_off;        // [pages + 1]
    size_t a_total = 0, w_total = 0; // crop scratch consumed by those jobs
    RecPart part;
};

constexpr int REC_GAP = 4;

// ------------------------------------------------------------------------------------------------ shared functions
template <typename T> inline T* upload(bbocr_ctx* c, const std::vector<T>& v) {
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, v.size() * sizeof(T)));
    c->owned.push_back(d);
    c->owned_bytes.push_back(v.size() * sizeof(T));
    HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return (T*)d;
}

ConvPlan make_plan(int Cin, int Cout, int KH, int KW, int pad, int dil, int el = 0, int bn = 0);
void upload_plan(bbocr_ctx* c, ConvPlan& p, const std::vector<float>& w, const std::vector<float>& b);
void free_weights(bbocr_ctx* c);
size_t weights_blob_bytes(const bbocr_ctx* c);
void weights_export(bbocr_ctx* c, void* dev_dst, size_t bytes);
void weights_import(bbocr_ctx* c, const void* dev_src, size_t bytes);
void load_craft(bbocr_ctx* c, const TensorMap& tm);
void load_crnn(bbocr_ctx* c, const TensorMap& tm);
hipError_t launch_conv_profiled(bbocr_ctx* c, const ConvPlan& p, ConvArgs a, bool may_decline = false);   // may_decline: hipErrorNotSupported is returned, not thrown
void run_conv(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, const Act* a1, bool relu1, bool relu_out, void* out, int out_cs, int cout_store, bool out_f32, const Act* addup = nullptr);
void prof_collect(bbocr_ctx* c);
Act conv_act(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, const Act* a1, bool relu1, bool relu_out, int store);
Act conv_pool_act(bbocr_ctx* c, const ConvPlan& p, const Act& a0, bool relu0, bool relu_out, int store, int mode, bool pool_relu, Act* full, const RgbSource* rgb = nullptr);
Act pool_act(bbocr_ctx* c, const Act& a, int kh, int kw, int sh, int sw, int ph, int pw, bool relu_in);
DetDims det_dims(int H, int W, int canvas, double mag);
void detect_impl(bbocr_ctx* c, const uint8_t* rgb, int B, int H, int W, const bbocr_params& p, float* heat, const std::function<void(int, int)>& after_sub = nullptr);
void boxes_impl(bbocr_ctx* c, const float* heat, int B, int h, int w, double ratio, const bbocr_params& p, HostBoxes& hb, hipStream_t st);
bbocr_boxlist* export_boxes(const HostBoxes& hb);
void import_boxes(const bbocr_boxlist* bl, HostBoxes& hb);
bool plan_horizontal(const std::array<int, 4>& box, int img, int H, int W, BoxJob& j);
bool plan_free(const std::array<double, 8>& fq, int img, BoxJob& j);
int rec_mode(const bbocr_ctx* c);
void crnn_features(bbocr_ctx* c, const uint16_t* crops, int n, int imgW, uint16_t* v_out);
void crnn_sequence(bbocr_ctx* c, size_t rows_pad, const int* tiles_dev, int ntiles, float* logits);
double percentile_u8(const unsigned int* hist, size_t n, double q);
void rec_early_begin(bbocr_ctx* c, const uint8_t* gray, int pages, int B, int H, int W, const HostBoxes& hb, const bbocr_params& p, RecEarly& e);
void recognize_impl(bbocr_ctx* c, const uint8_t* gray, int B, int H, int W, const HostBoxes& hb, const bbocr_params& p, std::vector<BoxJob>& jobs, std::vector<int>& box_off, RecEarly* early = nullptr);
bbocr_result* export_result(int B, const std::vector<BoxJob>& jobs, const std::vector<int>& box_off);
void pp_resize(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw, bool src_bgr = false);   // enqueues; the caller waits
void pp_gauss(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, double sigma);
const uint8_t* pp_fold_lut(bbocr_ctx* c, size_t n, double contrast, double brightness);
void pp_clahe(bbocr_ctx* c, const uint8_t* src, int H, int W, const uint8_t* d_lut, uint8_t* dst, double clip_limit);
void pp_unsharp(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, uint8_t* tmp1, uint8_t* tmp2, float radius, int percent, int threshold);
void preprocess_chain_impl(bbocr_ctx* c, const uint8_t* bgr, int H, int W, const bbocr_preproc_params& q, uint8_t* out, int dh, int dw);

// worker threads of this slot: bbocr_config::host_threads, or min(16, the process's CPU share); the pool is made once and kept
inline HostPool& host_pool(bbocr_ctx* c) {
    if (!c->pool) {
        const int want = c->cfg.host_threads > 0 ? c->cfg.host_threads : std::min(16, host_cpu_share());
        c->pool.reset(new HostPool(std::max(1, std::min(want, 64)) - 1));
    }
    return *c->pool;
}

// is another pipeline call running on this context right now (another call slot busy)?
inline bool other_call_in_flight(bbocr_ctx* c) {
    bbocr_ctx* root = c->root;
    std::lock_guard<std::mutex> lk(root->pool_mu);
    if (root != c && root->slot_busy) return true;
    for (bbocr_ctx* s : root->slots)
        if (s != c && s->slot_busy) return true;
    return false;
}

// ---- call slots
struct EnqLock {       // see bbocr_ctx::enq_mu; never held across a host wait for the device
    std::unique_lock<std::mutex> lk;
    explicit EnqLock(bbocr_ctx* c) : lk(c->root->enq_mu) {}
};
constexpr int kMaxSlots = 2;       // calls in flight per context (the reference runs 2 ThreadPoolExecutor workers on one Reader)
bbocr_ctx* slot_create(bbocr_ctx* root);                 // abi.cpp
void slot_destroy(bbocr_ctx* s);                         // abi.cpp (everything but the shared compute stream and the weights)
void publish_times(const bbocr_ctx* s);
void dist_release(bbocr_ctx* root);                      // dist.cpp                  // abi.cpp: the finished call's stage times -> the calling thread's record

// wait for what THIS slot has queued on `st` so far (the compute stream is shared between slots: hipStreamSynchronize would also wait for
// the other call's kernels queued behind ours -- and the card would then run dry while both hosts wait)
inline void slot_sync(bbocr_ctx* c, hipStream_t st) {
    if (!c->sync_ev) HIPCHK(hipEventCreateWithFlags(&c->sync_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->sync_ev, st));
    HIPCHK(hipEventSynchronize(c->sync_ev));
}

// bbocr.h promises that a call returns with its own work finished -- also when it fails half-way: kernels already queued may still
// read caller-owned inputs / write caller-owned outputs, and host vectors that were async-copy targets die during unwinding
inline void guarded_drain(bbocr_ctx* ctx) {
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->seq_stream) (void)hipStreamSynchronize(ctx->seq_stream);
    ctx->cur = ctx->stream;
    (void)hipGetLastError();
    for (auto& r : ctx->prof_recs) { ctx->prof_pool.push_back(r.e0); ctx->prof_pool.push_back(r.e1); }
    ctx->prof_recs.clear();
}

// RAII lease of a call slot.  exclusive: every slot (weight loading / export / import, anything that changes what the slots share).
struct SlotLease {
    bbocr_ctx* root;
    bbocr_ctx* slot = nullptr;
    bool exclusive;
    SlotLease(bbocr_ctx* r, bool excl) : root(r), exclusive(excl) {
        std::unique_lock<std::mutex> lk(root->pool_mu);
        const int max_slots = root->cfg.call_slots == 1 ? 1 : kMaxSlots;
        if (exclusive) {
            root->pool_cv.wait(lk, [&] {
                if (root->slot_busy) return false;
                for (bbocr_ctx* s : root->slots) if (s->slot_busy) return false;
                return true;
            });
            root->slot_busy = true;
            for (bbocr_ctx* s : root->slots) s->slot_busy = true;
            slot = root;
            return;
        }
        for (;;) {
            if (!root->slot_busy) { slot = root; break; }
            for (bbocr_ctx* s : root->slots) if (!s->slot_busy) { slot = s; break; }
            if (slot) break;
            if ((int)root->slots.size() + 1 < max_slots) {
                slot = slot_create(root);              // may throw StatusError: nothing is leased then
                root->slots.push_back(slot);
                break;
            }
            root->pool_cv.wait(lk);
        }
        slot->slot_busy = true;
        if (slot != root) {
            static_cast<WeightView&>(*slot) = static_cast<const WeightView&>(*root);   // weights may have been loaded since the slot was made
            slot->cfg = root->cfg;
            slot->profiling = root->profiling;
        }
    }
    ~SlotLease() {
        {
            std::lock_guard<std::mutex> lk(root->pool_mu);
            if (exclusive) {
                root->slot_busy = false;
                for (bbocr_ctx* s : root->slots) s->slot_busy = false;
            } else if (slot) {
                slot->slot_busy = false;
            }
        }
        root->pool_cv.notify_all();
    }
};

// every ABI entry point runs inside guarded(): device selected, a call slot leased, exceptions mapped to status codes.  f(slot) does the work.
template <typename F> inline int guarded(bbocr_ctx* root, F&& f, bool exclusive = false) {
    if (!root) return BBOCR_ERR_ARG;
    auto set_err = [&](const std::string& m) {
        std::lock_guard<std::mutex> lk(root->pool_mu);
        root->err = m;
    };
    bbocr_ctx* ctx = nullptr;
    try {
        hipError_t e = hipSetDevice(root->cfg.device);
        if (e != hipSuccess) fail(BBOCR_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
        SlotLease lease(root, exclusive);
        ctx = lease.slot;
        try {
            ctx->cur = ctx->stream;
            // the library runs on its own non-blocking streams: what the caller queued on the default stream (torch's) -- fills of
            // output buffers, input copies -- must be complete before our kernels touch the same memory
            e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) fail(BBOCR_ERR_HIP, std::string("default stream: ") + hipGetErrorString(e));
            f(ctx);
            publish_times(ctx);
            return BBOCR_OK;
        } catch (...) {
            guarded_drain(ctx);                      // still inside the lease: nobody else touches this slot's buffers meanwhile
            throw;
        }
    } catch (const StatusError& se) {
        set_err(se.msg);
        return se.code;
    } catch (const std::exception& ex) {
        set_err(ex.what());
        return BBOCR_ERR_INTERNAL;
    } catch (...) {
        set_err("unknown failure");
        return BBOCR_ERR_INTERNAL;
    }
}
