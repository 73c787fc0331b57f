/* libbbocr -- C ABI of the MI355X-native OCR backend (detector + recogniser) for BB-OCR.
 *
 * Drop-in boundary: the L3 -> L1 edge of the reference, i.e. what
 *     easyocr.Reader(["en"], gpu=...)                     pipeline_demo/extractor/enhanced_extractor.py:153
 *     reader.readtext(path, paragraph=False, batch_size=1, workers=0)      ...enhanced_extractor.py:520
 * compute.  The reference has no FFI of its own (it is pure Python on the third-party easyocr==1.7.2,
 * pipeline_demo/requirements.txt:7); the Python binding a maintainer adds is the ctypes stub shown in
 * INTEGRATION.md and shipped as bb_ocr_amd/_lib.py.
 *
 * Conventions: every entry point returns 0 on success and a negative bbocr_status on failure, never throws and
 * never aborts (the reference relies on catching exceptions: enhanced_extractor.py:529-531, i2j_ui/app/main.py:631-644);
 * bbocr_last_error() gives the message.  Image / heat-map pointers are DEVICE pointers (HBM); weight descriptors and
 * results are HOST memory.  All work of one call is ordered on the context's own HIP stream and the call returns after
 * its own work has finished.  One context may be used from several threads: up to bbocr_config::call_slots (default 2) pipeline
 * calls run at once, each in its own call slot (work buffers, side stream), sharing the weights and the compute stream; further
 * callers wait for a free slot; weight loading / export / import wait for every slot.  bbocr_stage_times and bbocr_last_error
 * answer for the calling thread.  ctypes releases the GIL around each call.
 */
#ifndef BBOCR_H
#define BBOCR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bbocr_ctx bbocr_ctx;

enum bbocr_status {
    BBOCR_OK = 0,
    BBOCR_ERR_ARG = -1,      /* bad argument / shape */
    BBOCR_ERR_HIP = -2,      /* a HIP runtime call failed */
    BBOCR_ERR_WEIGHTS = -3,  /* missing or mis-shaped tensor in a state-dict */
    BBOCR_ERR_STATE = -4,    /* weights not loaded, context destroyed, ... */
    BBOCR_ERR_OVERFLOW = -5, /* a device work buffer was too small (reported, never silent) */
    BBOCR_ERR_INTERNAL = -6
};

/* Streams: every context owns non-blocking HIP streams.  On entry each call waits for the legacy default stream (where torch
 * queues its copies and fills by default); device buffers produced on OTHER streams must be complete before they are passed in.
 * Every call returns with its own work finished (results are host-visible / device buffers final). */
typedef struct bbocr_config {
    int device;         /* HIP device ordinal */
    int det_sub_batch;  /* pages per detector pass; 0 = auto (<= 64 pages / 96 GB of activations, short last pass) */
    int rec_max_cols;   /* pixel columns (4 per pooled time step) whose sequence stage runs as one pass; 0 = default (6,000,000) */
    int precision;      /* arithmetic of the two networks (fixed per context: the weights are packed for it at bbocr_load_weights):
                         *   BBOCR_PREC_BF16  (0) bf16 MFMA operands and stored activations, fp32 accumulation;
                         *   BBOCR_PREC_FP16  (1) the same kernels on IEEE fp16 operands (v_mfma_f32_16x16x32_f16): 8x finer rounding,
                         *                        range 6e-8 .. 65504 -- BASELINE.json configs[4] ("fp16 MFMA conv path"), and what the Python host
                         *                        (bb_ocr_amd.Reader) selects unless told otherwise: boxes and strings equal to the fp32 CPU path's
                         *                        on every input class measured;
                         *   BBOCR_PREC_EXACT (2) BOTH networks in split fp16: every activation and weight is a pair hi + lo of fp16 values
                         *                        (22 significand bits) and every product runs as the three MFMA terms hi*hi + lo*hi + hi*lo
                         *                        accumulated in fp32, LSTM state and gates in fp32 -- threshold decisions and boxes follow the
                         *                        fp32 CPU path on arbitrary heat-maps, CONFIDENCES to 1e-5 (3x the MFMA work: ~0.3x FP16's rate);
                         *   BBOCR_PREC_MIXED (3) detector as BF16, recogniser as FP16: 1.6 % faster than FP16 and as exact on binary-ink pages
                         *                        (2,051 of 2,051 boxes and strings of the 1280x960 workload equal the fp32 CPU path's,
                         *                        profiles/r03_text_parity.json): bf16 keeps the detector's clock (fp16 operands
                         *                        toggle more bits under the power limit); on continuous-tone images its bf16
                         *                        heat-map flips a few threshold decisions (3 of 110 boxes on the reference's images).
                         *   BBOCR_PREC_EXACT_REC (4) detector as FP16, recogniser as in EXACT: the fp32 path's confidences (1e-5) -- what upstream's
                         *                        contrast-retry decision and "keep the better pass" read -- at ~0.65x FP16's rate; boxes as FP16's.
                         * Any other value: bbocr_create returns BBOCR_ERR_ARG. */
    int call_slots;     /* calls that may be IN FLIGHT on this context at once: 0 = default (2), 1 = calls serialise (rounds 1-3), 2.
                         * The reference shares one Reader between ThreadPoolExecutor workers (batch_processor_enhanced.py:215, default 2).
                         * A call that arrives while another is running takes the second slot: own work buffers (allocated on first use,
                         * i.e. only if two calls ever overlap), same weights, same compute stream -- its detector runs on the card while the
                         * first call's host thread finishes box geometry, CTC read-back and result export.  Results are those of the
                         * serial path bit for bit (tests/test_gpu_pipeline.py::test_two_calls_in_flight_equal_the_serial_path). */
    int host_threads;   /* host worker threads per call slot (box geometry of a detector pass, beam search): 0 = auto = min(16, the CPUs this
                         * PROCESS may use: scheduler affinity mask and cgroup CPU quota -- not the machine's core count, which every one of
                         * the 8 ranks of a node would claim for itself).  Ranks that are not pinned pass their share (the Python host:
                         * share / LOCAL_WORLD_SIZE).  The pool is created once per slot and kept; no thread is spawned per call. */
    int reserved[2];
} bbocr_config;
enum { BBOCR_PREC_BF16 = 0, BBOCR_PREC_FP16 = 1, BBOCR_PREC_EXACT = 2, BBOCR_PREC_MIXED = 3, BBOCR_PREC_EXACT_REC = 4 };

/* One tensor of an upstream state-dict (easyocr/craft.py::CRAFT or easyocr/model/vgg_model.py::Model key names,
 * optional "module." prefix), fp32, host memory, C-contiguous.  Replaces torch.load + load_state_dict in
 * easyocr/detection.py::get_detector and easyocr/recognition.py::get_recognizer. */
typedef struct bbocr_tensor_desc {
    const char* name;
    int ndim;
    int64_t shape[4];
    const float* data;
} bbocr_tensor_desc;

/* keyword arguments of easyocr.Reader.readtext that affect this path (same names, same defaults) */
enum { BBOCR_DECODER_GREEDY = 0, BBOCR_DECODER_BEAMSEARCH = 1 };

typedef struct bbocr_params {
    double text_threshold; /* 0.7 */
    double low_text;       /* 0.4 */
    double link_threshold; /* 0.4 */
    double mag_ratio;      /* 1.0 */
    double slope_ths;      /* 0.1 */
    double ycenter_ths;    /* 0.5 */
    double height_ths;     /* 0.5 */
    double width_ths;      /* 0.5 */
    double add_margin;     /* 0.1 */
    double contrast_ths;   /* 0.1 */
    double adjust_contrast;/* 0.5 */
    int canvas_size;       /* 2560 */
    int min_size;          /* 20 */
    unsigned int ignore_mask[4]; /* recognizer_predict's ignore_idx as a bit mask over class indices 0..127 (allowlist / blocklist):
                                  * those classes are zeroed and the rest renormalised before the arg-max; 0 = none (english_g2 default) */
    int decoder;           /* BBOCR_DECODER_GREEDY (0, readtext's default, the reference's call) or BBOCR_DECODER_BEAMSEARCH (1):
                            * easyocr/utils.py::ctcBeamSearch without a language model, run on the host over the device's probabilities;
                            * the confidence is the greedy path's custom_mean for both, as upstream computes it */
    int beam_width;        /* 5 (readtext's beamWidth); used when decoder == BBOCR_DECODER_BEAMSEARCH */
    int rotation_info[4];  /* readtext's rotation_info: up to three angles out of {90, 180, 270}, zero-terminated (all 0 = None, the
                            * reference's call).  Non-empty: Reader.recognize's batched branch -- every crop of a page padded to the page's
                            * max_width, each also recognised as np.rot90(crop, angle/90), the most confident variant reported, results
                            * sorted by the boxes' top y */
} bbocr_params;

/* output of detection (easyocr.Reader.detect): per image horizontal_list / free_list, plus the ungrouped polygons */
typedef struct bbocr_boxlist {
    int n_images;
    int* poly_off;   /* [n_images+1] */
    int* polys;      /* [poly_off[n]][8]  int32 x,y x4 (detection.get_textbox output, image coordinates) */
    int* hori_off;   /* [n_images+1] */
    int* hori;       /* [hori_off[n]][4]  xmin, xmax, ymin, ymax */
    int* free_off;   /* [n_images+1] */
    double* free_q;  /* [free_off[n]][8]  x,y x4 */
} bbocr_boxlist;

/* output of readtext: boxes in upstream order (horizontal boxes, then free boxes), per image */
typedef struct bbocr_result {
    int n_images;
    int* box_off;     /* [n_images+1] */
    double* quads;    /* [n_boxes][8]; horizontal boxes are the clamped integer corners */
    int* is_free;     /* [n_boxes] */
    int* text_off;    /* [n_boxes+1] into text_idx */
    int* text_idx;    /* class indices 1..96: position in the english_g2 character list (0 = CTC blank, never emitted) */
    double* conf;     /* [n_boxes] custom_mean confidence */
} bbocr_result;

int bbocr_create(const bbocr_config* cfg, bbocr_ctx** out);
void bbocr_destroy(bbocr_ctx* ctx);
const char* bbocr_last_error(bbocr_ctx* ctx);
void bbocr_default_params(bbocr_params* p);

/* which: 0 = CRAFT detector (craft_mlt_25k layout), 1 = CRNN recogniser (english_g2 layout).  BatchNorm is folded,
 * weights are packed to the MFMA fragment layout in bf16 and uploaded. */
int bbocr_load_weights(bbocr_ctx* ctx, int which, const bbocr_tensor_desc* descs, int n);

/* ---- multi-GPU: the packed weights as one device blob ("RCCL broadcast of detector/recognizer weights") ----
 * What easyocr does per forward under nn.DataParallel (broadcast_coalesced of every parameter, easyocr.py::get_detector /
 * get_recognizer) happens ONCE here, on the tensors as the kernels read them: rank 0 loads the state-dicts
 * (bbocr_load_weights), exports the BN-folded, element-type-rounded, MFMA-packed images as ONE contiguous DEVICE buffer
 * (bbocr_weights_export; ~49 MB in bf16) and broadcasts it device-to-device (ncclBroadcast over xGMI; torch.distributed.broadcast
 * in the Python host); every other rank lays its plans out with bbocr_alloc_weights (no values) and fills them with
 * bbocr_weights_import.  Size and layout depend only on bbocr_config::precision and on which networks are present; a blob from a
 * context with another precision is refused (BBOCR_ERR_WEIGHTS). */
int bbocr_alloc_weights(bbocr_ctx* ctx, int which /* 0 = detector, 1 = recogniser */);
int bbocr_weights_blob_size(bbocr_ctx* ctx, size_t* bytes);
int bbocr_weights_export(bbocr_ctx* ctx, void* dev_blob, size_t bytes);
int bbocr_weights_import(bbocr_ctx* ctx, const void* dev_blob, size_t bytes);

/* ---- multi-GPU for hosts without torch.distributed: one process per GPU, RCCL over xGMI ----
 * The Python host reaches RCCL through torch.distributed (bb-ocr_amd/dist.py); these entry points do the same exchanges from C.  RCCL is
 * loaded with dlopen("librccl.so.1") at the first call, so single-GPU users need no RCCL.  Rank 0 makes the 128-byte id
 * (bbocr_dist_unique_id = ncclGetUniqueId) and hands it to the other ranks out of band (environment, file, socket); every rank then
 * calls bbocr_dist_init (ncclCommInitRank: collective).  All of them are collective over the ranks of the communicator and wait for every
 * call slot of the context.  There is no collective inside the OCR path itself: pages are independent.
 *   bbocr_bcast_weights   ONE ncclBroadcast of the packed weight blob (root: bbocr_load_weights; others: bbocr_alloc_weights, same
 *                         precision) -- replaces nn.DataParallel's per-forward broadcast_coalesced (easyocr.py::get_detector / get_recognizer);
 *   bbocr_scatter_images  the loader rank's units (pages of unit_bytes = H*W*3) -> each rank's contiguous block [first, first + count)
 *                         (block partition, uneven / empty blocks allowed) as one grouped batch of ncclSend: the root's links carry all
 *                         blocks at once; dev_local holds ceil(n_units / world) units;
 *   bbocr_gather_results  per-rank host bytes (bbocr_result_pack of the rank's results) -> on root one malloc'd concatenation in rank
 *                         order (*all, free with bbocr_free_bytes) + sizes[world]; bbocr_result_unpack rebuilds each bbocr_result. */
int bbocr_dist_unique_id(void* id128);
int bbocr_dist_init(bbocr_ctx* ctx, int rank, int world, const void* id128);
int bbocr_dist_finalize(bbocr_ctx* ctx);
int bbocr_bcast_weights(bbocr_ctx* ctx, int root);
int bbocr_scatter_images(bbocr_ctx* ctx, const uint8_t* dev_all, long long n_units, size_t unit_bytes, int root, uint8_t* dev_local, long long* first,
                         long long* count);
int bbocr_gather_results(bbocr_ctx* ctx, const void* local, size_t local_bytes, int root, void** all, size_t* sizes);
int bbocr_result_pack(const bbocr_result* r, void** bytes, size_t* n);       /* free *bytes with bbocr_free_bytes */
int bbocr_result_unpack(const void* bytes, size_t n, bbocr_result** out);    /* free *out with bbocr_free_result */
void bbocr_free_bytes(void* p);

/* geometry of the detector for an H x W page: network input H32 x W32 (after canvas_size scaling, padded to x32),
 * heat-map h x w = H32/2 x W32/2, ratio as returned by resize_aspect_ratio */
int bbocr_detect_dims(int H, int W, int canvas_size, double mag_ratio, int* H32, int* W32, int* rh, int* rw, double* ratio);

/* S2+S3: uint8 RGB pages [B,H,W,3] (device) -> region/affinity heat-map fp32 [B,h,w,2] (device).
 * Replaces detection.test_net's resize_aspect_ratio + normalizeMeanVariance + CRAFT.forward. */
int bbocr_detect(bbocr_ctx* ctx, const uint8_t* dev_rgb, int B, int H, int W, const bbocr_params* p, float* dev_heat_out);

/* S4+S5: heat-map -> boxes.  Replaces craft_utils.getDetBoxes + adjustResultCoordinates + get_textbox +
 * utils.group_text_box + the min_size filter of Reader.detect.  ratio is the one bbocr_detect_dims returned. */
int bbocr_boxes(bbocr_ctx* ctx, const float* dev_heat, int B, int h, int w, double ratio, const bbocr_params* p, bbocr_boxlist** out);

/* S6-S10: gray pages [B,H,W] uint8 (device) + boxes -> text.  Replaces Reader.recognize (per-box branch:
 * get_image_list, AlignCollate, CRNN forward, softmax, greedy CTC, contrast retry). */
int bbocr_recognize(bbocr_ctx* ctx, const uint8_t* dev_gray, int B, int H, int W, const bbocr_boxlist* boxes, const bbocr_params* p,
                    bbocr_result** out);

/* all stages in one call (== Reader.readtext_batched for equally sized pages).  dev_gray may be NULL: it is then
 * derived from dev_rgb with cv2's BGR2GRAY fixed-point formula, as upstream does for ndarray input. */
int bbocr_readtext_batch(bbocr_ctx* ctx, const uint8_t* dev_rgb, const uint8_t* dev_gray, int B, int H, int W, const bbocr_params* p,
                         bbocr_result** out);

void bbocr_free_boxlist(bbocr_boxlist* b);
void bbocr_free_result(bbocr_result* r);

/* milliseconds spent in the last readtext/detect/boxes/recognize call OF THE CALLING THREAD on this context, per stage:
 * [0] detector net (S2+S3), [1] CCL kernels (S4 device), [2] box geometry + grouping (S4/S5 host),
 * [3] crops (S6/S7), [4] recogniser net (S8), [5] CTC decode (S9), [6] contrast retry pass, [7] total */
int bbocr_stage_times(bbocr_ctx* ctx, float* ms, int n);

/* Per-launch timing of the dominant kernel (conv_mfma) with HIP events recorded on the context's stream.
 * group 0 = detector convs, 1 = recogniser convs/GEMMs.  on = 1 times group 0 only (54 launches per 64-page step: the roofline
 * leg of bench.py), on = 2 both groups (about ten times as many events: ~1.5 % of a step), 0 = off.  Totals accumulate from the call on:
 * ms = sum of launch durations, flops = sum of ALGORITHMIC flops (2*N*OH*OW*Cout*Cin*KH*KW, unpadded). */
int bbocr_set_profiling(bbocr_ctx* ctx, int on);
int bbocr_conv_profile(bbocr_ctx* ctx, int group, double* ms, double* flops, long long* launches);

/* ---- host-only halves of S4/S5 (no device work; callable without a GPU, used by the CPU test-suite) ----
 * comps: [n][7] = root, left, top, right, bottom, area, row_off (heat-map coordinates); rowext: per component row the
 * min/max x of TEXT pixels ([row_off + y - top][2], max < 0 = none) -- exactly what the CCL kernels emit.
 * Writes n polygons [n][8] (craft_utils.getDetBoxes_core tail + adjustResultCoordinates + get_textbox). */
/* CPUs this PROCESS may use (scheduler affinity mask, cgroup v2 cpu.max / v1 cfs quota): what bbocr_config::host_threads = 0 sizes the
 * per-slot host pools from (capped at 16) -- never the machine's core count */
int bbocr_host_cpu_share(void);
int bbocr_host_component_polys(const int* comps, const int* rowext, int n, int w, int h, double ratio, int* polys_out);
/* utils.group_text_box + Reader.detect's min_size filter on n polygons of one image */
int bbocr_host_group_boxes(const int* polys, int n, const bbocr_params* p, bbocr_boxlist** out);
/* easyocr/utils.py::ctcBeamSearch (CTCLabelConverter.decode_beamsearch, no language model) on host probabilities fp32 [n,T,cs]
 * (C <= cs classes, class 0 = blank): text_off [n+1], text_idx (<= n*T).  The host half of bbocr_params::decoder == BEAMSEARCH */
int bbocr_host_ctc_beam(const float* probs, int n, int T, int C, int cs, int beam_width, int* text_off, int* text_idx);

/* ---- single-operator entry points (used by the parity tests; same kernels the pipeline runs) ---- */
/* conv2d on device tensors: in bf16 NHWC [N,H,W,Cin] (as uint16 bits), weights fp32 OIHW on the host (+bias or NULL),
 * out bf16 (out_f32 = 0) or fp32 NHWC [N,OH,OW,Cout_store]; Cin % 32 == 0; Cout_store = roundup16(Cout).  With
 * bbocr_config::precision FP16 / EXACT "bf16" reads "fp16" throughout (the DETECTOR's element type of the context; MIXED: bf16).
 * pool_mode 1/2 fuses MaxPool2d(2,2) / MaxPool2d((2,1),(2,1)) (optionally after ReLU: pool_relu) into the epilogue and
 * writes bf16 [N,OH/2,OW/2 or OW,Cout_store] to dev_pool_out; dev_out may then be NULL (pooled output only). */
int bbocr_op_conv2d(bbocr_ctx* ctx, const uint16_t* dev_in, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                    int KH, int KW, int pad, int dil, int relu_in, int relu_out, int out_f32, void* dev_out, int pool_mode, int pool_relu,
                    uint16_t* dev_pool_out);
/* recogniser network only: crops [n,64,imgW] (device, 16-bit elements of the context's RECOGNISER type: bf16 (BF16) / fp16 (FP16, MIXED) values already
 * normalised to [-1, 1]; BBOCR_PREC_EXACT: CODES, 0 = padding zero, 1 + grey level otherwise, from which the first layer rebuilds the
 * fp32 input ((g/255 - 0.5)/0.5) exactly) -> logits fp32 [n,T,112] (device), T = imgW/4-1 */
int bbocr_crnn_logits(bbocr_ctx* ctx, const uint16_t* dev_crops, int n, int imgW, float* dev_logits);
/* CTC decode of logits fp32 [n,T,cs]: host outputs text_off [n+1], text_idx (<= n*T), conf [n]; ignore_mask: 4 x 32-bit class mask
 * (bbocr_params::ignore_mask) or NULL; beam_width <= 0: greedy, > 0: ctcBeamSearch with that width (bbocr_params::decoder) */
int bbocr_op_ctc(bbocr_ctx* ctx, const float* dev_logits, int n, int T, int C, int cs, int* text_off, int* text_idx, double* conf,
                 const unsigned int* ignore_mask, int beam_width);
/* cv2.resize(INTER_LINEAR) on uint8 [N,sh,sw,C] -> [N,dh,dw,C] (device) */
int bbocr_op_resize_u8(bbocr_ctx* ctx, const uint8_t* dev_src, int N, int sh, int sw, int C, uint8_t* dev_dst, int dh, int dw);
/* JPEG pages decoded ONCE on the host into libjpeg's YCbCr triples (out_color_space = JCS_YCbCr; PIL: draft("YCbCr")): uint8 [npix,3]
 * (device) -> the RGB image libjpeg's own ycc_rgb_convert yields (what skimage / cv2.imread hand easyocr.utils.reformat_input,
 * reader.readtext at enhanced_extractor.py:520) and, when dev_gray is not NULL, the Y plane = cv2.imread(IMREAD_GRAYSCALE)'s
 * plane; both bit for bit (tests/test_oracle_cpu.py pins the formula against the decoder) */
int bbocr_op_ycc_to_rgb(bbocr_ctx* ctx, const uint8_t* dev_ycc, size_t npix, int pixel_stride, uint8_t* dev_rgb, uint8_t* dev_gray);
/* n separately allocated host pages of bytes_each bytes (the arrays a decode pool returns) -> one device buffer [n][bytes_each], copied
 * inside ONE call: a Python host releases its interpreter lock once per batch instead of once per page, and needs no host-side
 * concatenation of the pages (236 MB for 64 pages of 1280x960).  pixel_stride above: 3 = tight triples, 4 = Pillow's own 4-byte pixel
 * storage (Y Cb Cr x) uploaded as it is. */
int bbocr_upload_pages(bbocr_ctx* ctx, const void* const* host_pages, int n, size_t bytes_each, void* dev_dst);
/* recogniser inputs for explicit boxes of ONE gray page: fills crops bf16 [n,64,imgW] (in box order); returns their count in *n_out.
 * contrast != 0 applies adjust_contrast_grey first.  mode 0: the boxes whose own padded width is imgW (Reader.recognize's per-box
 * branch); mode 1..4: EVERY box at the forced width imgW, rotated by np.rot90(crop, mode - 1) (the batched branch rotation_info takes) */
int bbocr_op_crops(bbocr_ctx* ctx, const uint8_t* dev_gray, int H, int W, const int* hori, int n_hori, const double* free_q, int n_free,
                   int imgW, float contrast, uint16_t* dev_out, int* n_out, int mode);

/* ---- OCR pre-processing chain of the reference ----
 * pipeline_demo/ocr_testing/preprocessing/image_preprocessor.py::preprocess_for_book_cover (:147-160) on ONE decoded page:
 * dev_bgr uint8 [H,W,3] in cv2.imread's channel order -> dev_out uint8 [int(H*1.5), int(W*1.5)] (gray): BGR2GRAY, x1.5
 * INTER_CUBIC, GaussianBlur 3x3 sigma 3, PIL Contrast 1.9, PIL Brightness 1.2, CLAHE (clip 2.5, 8x8 tiles),
 * PIL UnsharpMask(radius 1, 30 %, threshold 3).  out_h / out_w receive the output size (pass dev_out = NULL to query it). */
int bbocr_preprocess_book_cover(bbocr_ctx* ctx, const uint8_t* dev_bgr, int H, int W, uint8_t* dev_out, int* out_h, int* out_w);
/* The same ImagePreprocessor stage sequence with its parameters spelled out (a stage whose parameter is 0 is skipped).
 * bbocr_preproc_defaults(p, 0) = the pipeline_demo chain above; (p, 1) = the LEGACY chain of
 * pipeline_components/img_to_json/ocr_testing/preprocessing/image_preprocessor.py:221-252 (sigma 5, contrast 1.3, no brightness
 * step, CLAHE 2.0, unsharp 20 %) -- the one whose stored outputs (results/images/book*_preprocessed.png) pin the stages. */
typedef struct bbocr_preproc_params {
    double scale;           /* resize(scale_factor): new size int(H*s) x int(W*s), cv2.INTER_CUBIC */
    double blur_sigma;      /* denoise(strength): cv2.GaussianBlur 3x3, sigma */
    double contrast;        /* increase_contrast(factor): PIL ImageEnhance.Contrast */
    double brightness;      /* increase_brightness(factor): PIL ImageEnhance.Brightness */
    double clahe_clip;      /* clahe(clip_limit), 8x8 tiles */
    double unsharp_radius;  /* sharpen: PIL UnsharpMask radius (1.0) */
    int unsharp_percent;    /* int(amount * 100) */
    int unsharp_threshold;  /* 3 */
    int reserved[4];
} bbocr_preproc_params;
void bbocr_preproc_defaults(bbocr_preproc_params* p, int legacy);
int bbocr_preprocess_chain(bbocr_ctx* ctx, const uint8_t* dev_bgr, int H, int W, const bbocr_preproc_params* p, uint8_t* dev_out, int* out_h,
                           int* out_w);
/* the chain's stages one by one (parity tests): stage 0 = resize cubic of a gray plane to (dh, dw), 1 = GaussianBlur 3x3
 * sigma `param`, 2 = PIL Contrast `param`, 3 = PIL Brightness `param`, 4 = CLAHE clip `param` 8x8, 5 = PIL UnsharpMask
 * (radius `param`, 30 %, threshold 3), 6 = cv2 BGR2GRAY of an interleaved [H,W,3] plane (the gray plane reformat_input derives from
 * arrays), 7 = PIL UnsharpMask (radius 1, `param` %, threshold 3).  Stages 1-7 keep the size (dh = H, dw = W). */
int bbocr_op_preprocess_stage(bbocr_ctx* ctx, int stage, const uint8_t* dev_src, int H, int W, uint8_t* dev_dst, int dh, int dw, double param);

#ifdef __cplusplus
}
#endif
#endif
