// Internal launch interface between the C-ABI layer (api.cpp) and the HIP kernels.  Not public.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

// ------------------------------------------------------------------ implicit-GEMM convolution (conv_mfma.hip)
struct ConvArgs {
    const uint16_t* in0;   // bf16 NHWC source 0
    const uint16_t* in1;   // bf16 NHWC source 1 (virtual channel concat after source 0) or null
    const uint16_t* wpk;   // packed bf16 weights, see pack_conv_weights()
    const float* bias;     // fp32 [Cout_pad] (BN folded), natural cout order
    void* out;             // bf16 or fp32 NHWC
    int C0, C1;            // channels taken from each source (multiples of 32)
    int in0_cs, in1_cs;    // pixel strides (elements) of the sources
    int N, H, W, OH, OW;
    int KH, KW, pad_h, pad_w, dil;
    int TH, TW, tiles_x, tiles_y, ntiles_n;
    int PH, PW, NP;        // activation patch (TH+(KH-1)dil) x (TW+(KW-1)dil), NP = roundup16(PH*PW)
    int relu_in0, relu_in1, relu_out, out_f32;
    int out_cs, cout_store;  // output pixel stride (elements), couts actually stored (multiple of 16)
    int nchunks, ntaps;
    // fused max-pool in the epilogue: 0 none, 1 = MaxPool2d(2,2), 2 = MaxPool2d((2,1),(2,1)); bf16 output [N,OH/2,OW(/2),pool_cs]
    int pool_mode, pool_relu, store_full, pool_cs;
    void* pool_out;
    int dbg;               // timing-only ablation switches (BBOCR_CONV_DBG); 0 in production
    const float* tail;     // non-null: fuse the CRAFT classifier tail (two 1x1 convs on 16 channels) into the epilogue (BN=64 config,
                           // cout_store 16): {b1[16], w2[32], b2[2]}; tail_frag: W1 as a bf16 MFMA A fragment [64 lanes][8]
    const uint16_t* tail_frag;
    // post_w: a 1x1 conv 64 -> 64 (no bias, no ReLU) applied to the finished 64 output channels in the epilogue, packed by
    // pack_post1x1_weights; only the product is stored (CRAFT: z = W_y u3b behind upconv3's 3x3, launch_conv declines other shapes)
    const uint16_t* post_w;
    int sub;               // > 1: dilated 3x3 run as sub*sub plain convs on the phase sub-lattices (set by launch_conv)
    int stack;             // sub > 1: rows of one phase image; the sub*sub images of a page are stacked along y, one zero row apart
    const void* zero;      // >= 16 zero bytes in device memory (source of padding pixels for the LDS-DMA staged variant)
    // non-null: add the 2x bilinear up-sampling (align_corners=False) of this bf16 NHWC tensor [N, up_H/2, up_W/2, up_cs] to the
    // accumulators before bias/ReLU -- conv1x1(cat[up(y), s]) == up(conv1x1_y(y)) + conv1x1_s(s): the U-net 1x1 layers never
    // materialise up(y).  up_H/up_W: size of the (full-resolution) output image; plain store path only.
    const uint16_t* addup;
    int up_H, up_W, up_cs;
    // non-null: CRAFT conv1_2 with conv1_1 fused in -- in0 is the uint8 RGB batch [N, rgb_H, rgb_W, 3] on the H x W canvas,
    // c11_w the conv1_1 weights packed by pack_conv1_1_weights_fused, c11_b its 64 biases (C0 stays 64)
    const uint16_t* c11_w;
    const float* c11_b;
    int rgb_H, rgb_W;
    unsigned long long* stamps;   // diagnostic (BBOCR_CONV_STAMPS): per workgroup {t_start, t_prologue, t_mainloop, t_end} s_memtime; null in production
    // conv3x3_up4_kernel (CRAFT upconv4 as ONE launch): in0 = the skip tensor s1 [N,H,W,128], addup/up_* = z = W_y y at half resolution,
    // aux_w / aux_b = the packed 1x1 weights (64 couts x 128 cin, BN = 64 plan) and bias of upconv4.conv.0; wpk / bias = upconv4.conv.3
    const uint16_t* aux_w;
    const float* aux_b;
    float acc_scale;       // accumulators are multiplied by this before the bias (set from ConvPlan::acc_scale by launch_conv; 1 unless
                           // the packed weights carry a power-of-two scale, see the split-fp16 plans)
    int lean;              // conv3x3_dma_kernel: which epilogue (set by its launcher: 0 shared, 1 / 2 plain lean with whole / half-line stores, 3 pooled lean)
    int split_off;         // > 0 (fp16 element type only): every stored value v goes out as the pair hi = fp16(v) at its channel and
                           // lo = fp16(v - hi) at channel + split_off -- the [hi | lo] activation layout of the exact recogniser mode
};

struct ConvPlan {      // host-side description of one packed conv layer
    int Cin = 0, Cout = 0;     // logical sizes
    int Cin_pad = 0, Cout_pad = 0;
    int KH = 1, KW = 1, pad_h = 0, pad_w = 0, dil = 1;
    int BN = 64;       // cout tile of the launch config chosen for this layer (64/128/256)
    int el = 0;        // element type of the packed weights and of the activations this layer reads / writes: 0 bf16, 1 fp16
    float acc_scale = 1.f;   // 2^-s when the packed weights are w * 2^s (exact; split-fp16 plans keep w_lo out of fp16's subnormals)
    int split = 0;     // 1: split-fp16 plan (weights.cpp::upload_split_plan): Cin counts the three blocks [a_hi | a_lo | a_hi]
    uint16_t* d_w = nullptr;   // device packed weights
    float* d_b = nullptr;      // device bias [Cout_pad]
};

int conv_plan_bn(int Cout);   // cout tile (64/128/256) of the launch configuration used for a layer
size_t conv_packed_elems(const ConvPlan& p);
// w: fp32 [Cout][Cin][KH][KW] already BN-folded; out: bf16 bits, layout [ntile][chunk][tap][frag][lane][8]
void pack_conv_weights(const ConvPlan& p, const float* w, uint16_t* out);
hipError_t launch_conv(const ConvPlan& p, ConvArgs a, hipStream_t s);   // a.zero must be set (device zero page)
// upconv4 of CRAFT fused: u4b = relu(conv3x3(relu(up(z) + W_s s1 + b_s)) + b): p1 = the 1x1 plan over s1 (128 -> 64), p3 = the 3x3 plan (64 -> 32).
// a: in0 = s1 (in0_cs = 128), addup = z (up_cs = 64), N/H/W of the full-resolution image, out/out_cs/cout_store of u4b.  Returns
// hipErrorNotSupported when the shapes are not the ones the kernel is built for (the caller then runs the two launches).
hipError_t launch_up4_fused(const ConvPlan& p1, const ConvPlan& p3, ConvArgs a, hipStream_t s);

// ------------------------------------------------------------------ detector front/back (craft_misc.hip)
void pack_post1x1_weights(const float* w /*[64][64] cout x cin*/, uint16_t* out /*[2][4][64][8]*/, int el);
void pack_conv1_1_weights_fused(const float* w /*[64][3][3][3] folded*/, uint16_t* out /*[4][64][8]*/, int el);   // one k-step: taps 2g, 2g+1 per lane group, tap 8 in the pad slots; couts in the conv epilogue's run order
hipError_t launch_maxpool(const uint16_t* in, uint16_t* out, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                          int relu_in, hipStream_t s);
hipError_t launch_gray(const uint8_t* rgb, uint8_t* gray, size_t npix, hipStream_t s);
hipError_t launch_ycc_to_rgb_gray(const uint8_t* ycc, int stride, uint8_t* rgb, uint8_t* gray, size_t npix, hipStream_t s);
hipError_t launch_resize_u8(const uint8_t* src, int N, int sh, int sw, int C, uint8_t* dst, int dh, int dw, hipStream_t s);

// ------------------------------------------------------------------ exact detector, element-wise helpers on pair tensors (craft_pair.hip)
// pair tensor = [hi C | lo C] fp16 per pixel, value = hi + lo / 2048 (REC_SPLIT below); C % 8 == 0 everywhere
hipError_t launch_pair_conv1_1(const uint8_t* rgb, int N, int Hi, int Wi, int H, int W, const float* w /*[64][3][3][3] folded*/, const float* b, uint16_t* out,
                               hipStream_t s);
hipError_t launch_pair_relu(const uint16_t* in, uint16_t* out, size_t npix, int C, hipStream_t s);
hipError_t launch_pair_maxpool3x3s1(const uint16_t* in, uint16_t* out, int N, int H, int W, int C, hipStream_t s);
// out = cat([interpolate(y -> H x W, bilinear, align_corners=False), skip]) (yh x yw == H x W: plain concat)
hipError_t launch_pair_upcat(const uint16_t* y, int yh, int yw, int Cy, const uint16_t* skip, int Cs, uint16_t* out, int N, int H, int W, hipStream_t s);
hipError_t launch_pair_cls_tail(const uint16_t* in /*[npix, 16 | 16]*/, const float* w1 /*[16][16]*/, const float* tail /*b1[16] w2[32] b2[2]*/, float* heat,
                                size_t npix, hipStream_t s);

// ------------------------------------------------------------------ box extraction (ccl.hip)
struct CclOut {        // per accepted component, device-written, host-sorted by root
    int root, left, top, right, bottom, area, row_off, img;
};
// label/slot: [N*h*w]; stat: [N*h*w][6]; comps: [cap_comps] and rowext: [cap_rows][2] for the WHOLE batch; counters: [4] = ncomps, nrows, overflow
hipError_t launch_ccl(const float* heat, int N, int h, int w, float low_text, float link_thr, double text_thr, int* label, int* stat,
                      int* slot, CclOut* comps, int* rowext, int* counters, int cap_comps, int cap_rows, hipStream_t s);

// ------------------------------------------------------------------ recogniser (crnn_misc.hip, lstm.hip, ctc.hip)
struct CropDesc {      // one recogniser input, filled on the host
    int img;           // page index in the batch
    int sx0, sy0, sw, sh;   // source rectangle: in the gray page, or (0,0,ww,wh) of the warped crop when warp != 0
    int rw, rh;        // cv2.resize target; rh == 64 unless the box is taller than wide (then rw == 64)
    int fw;            // content width after AlignCollate (<= imgW)
    int imgW;          // padded width (bucket)
    int slot;          // row inside the bucket tensor
    int warp;          // 1: four_point_transform crop, uses Minv
    int warp_off;      // byte offset of the warped crop in the warp scratch
    int a_off;         // byte offset of the stage-A (cv2-resized) crop in the crop scratch
    int lut_off;       // >= 0: contrast LUT (256 bytes) offset, -1: none
    int pad_;          // wide recogniser image: first pooled row (time step) of this crop in the sequence tensors
    int rot;           // rotation_info variant: the stage-A image is np.rot90(resized crop, rot); rw x rh are its dimensions AFTER the rotation
    double Minv[9];    // dst -> src homography (already inverted)
};
// stage_mask bit0: gather (warp) + cv2 resize into scratch; bit1: (PIL bicubic) + LUT + normalise + pad into out_bucket
hipError_t launch_crops(const uint8_t* gray, int H, int W, const CropDesc* descs_dev, int first, int count, int imgW, int any_warp,
                        int any_tall, uint8_t* wscratch, uint8_t* scratch, uint8_t* hscratch, const uint8_t* luts, uint16_t* out_bucket,
                        int stage_mask, hipStream_t s, int wide_row_stride = 0, int gap = 0, int mode = 0);   // wide_row_stride > 0: ONE image [64][Wt], slot = first column
// Recogniser tensor modes (bbocr_config::precision): REC_BF16 / REC_F16 = 16-bit elements; REC_SPLIT = the exact mode: the crop image
// holds CODES (0 = padding zero, 1 + grey level otherwise -- conv0 rebuilds the fp32 input ((g/255 - 0.5)/0.5) exactly), every later
// activation is a pair of fp16 tensors [hi C | lo C] per pixel with value = hi + lo / 2048 (lo scaled so that it stays in fp16's
// normal range), the LSTM reads an fp32 input projection.
enum { REC_BF16 = 0, REC_F16 = 1, REC_SPLIT = 2 };
constexpr float SPLIT_LO_SCALE = 2048.f;
hipError_t launch_crop_hist(const uint8_t* scratch, const CropDesc* descs_dev, int first, int count, unsigned int* hist, hipStream_t s);
hipError_t launch_crnn_conv0(const uint16_t* in, const float* w /*[9][32] tap-major*/, const float* b, uint16_t* out, int n, int W, int mode, hipStream_t s,
                             const uint16_t* afrag = nullptr /*pack_crnn_conv0_mfma: the MFMA form (bf16 / fp16 modes, W % 4 == 0)*/);
void pack_crnn_conv0_mfma(const float* w_tap_major /*[9][32]*/, uint16_t* out /*[2][64][8]*/, int el);
hipError_t launch_rowmean3(const uint16_t* in, uint16_t* out, int n, int T, int C, int mode, hipStream_t s);   // C: logical channels
// wide recogniser image (all crops side by side, CropDesc::slot = first column, ::pad_ = first pooled row): clear the separator
// columns of a layer output [H][Wl][C] (shift = log2 horizontal down-scale), and the 3-row mean gathered into the pooled rows
hipError_t launch_crnn_zero_gaps(uint16_t* t, const CropDesc* descs_dev, int first, int count, int H, int Wl, int C, int shift, hipStream_t s);
hipError_t launch_rowmean3_gather(const uint16_t* in, int Wc, int C, const CropDesc* descs_dev, int first, int count, uint16_t* out, int mode,
                                  hipStream_t s);
// BiLSTM recurrence: xproj bf16 [n,T,2048] (permuted channels, see lstm8_xproj_channel), out bf16 [n,T,512] (fwd | bwd)
// tiles_dev: int4 per workgroup {first row, sequences (<=16), T, 0}; tensors are pooled over all buckets: [rows, C]
// mode REC_SPLIT: xproj is FP32 [rows, 2048], out is the pair [rows, 512 hi | 512 lo] with the lo half UNSCALED (fp16(h - hi): the linear
// layer behind it is packed with lo scale 1), whh_pk comes from pack_lstm_whh_split and acc_scale is the inverse of its weight scale;
// tiles hold up to lstm_tile_seqs(mode) sequences
hipError_t launch_lstm(const void* xproj, const uint16_t* whh_pk, uint16_t* out, const int* tiles_dev, int ntiles, int mode, float acc_scale,
                       hipStream_t s);
int lstm_tile_seqs(int mode);    // sequences per LSTM workgroup (tile table entries): 16, or 32 for REC_SPLIT
size_t lstm_whh_packed_elems();
void pack_lstm_whh8(const float* whh_fwd, const float* whh_bwd, uint16_t* out, int el);
size_t lstm_whh_split_packed_elems();
float pack_lstm_whh_split(const float* whh_fwd, const float* whh_bwd, uint16_t* out);   // -> acc_scale (2^-s)
int lstm8_xproj_channel(int dir, int gate, int unit);
struct CtcOut { int len; int cnt; float prod; int pad; };
// seqs_dev: int2 per sequence {first row, T}; logits fp32 [rows, cs]; out_idx is row-indexed like the pool
hipError_t launch_ctc(const float* logits, size_t rows, int C, int cs, const int* seqs_dev, int nseq, int* idx_tmp, float* pmax_tmp,
                      int* out_idx, CtcOut* out, hipStream_t s, const unsigned int* ignore = nullptr, float* probs_out = nullptr);   // ignore: 4 x 32-bit class mask or null
// decoder='beamsearch' (easyocr/utils.py::ctcBeamSearch, host): probs fp32 [rows, cs] as ctc_rows_kernel writes them; seqs = {first row, T}
void ctc_beam_search_host(const float* mat, int T, int C, int cs, int beam_width, std::vector<int>& text);
class HostPool;
void ctc_beam_search_batch(const float* probs, const int* seqs, int nseq, int C, int cs, int beam_width, std::vector<std::vector<int>>& texts,
                           HostPool* pool = nullptr);     // pool: the calling slot's workers (null: the calling thread alone)

// ------------------------------------------------------------------ OCR pre-processing chain (preproc.hip), SURVEY 8 row f2
int pp_resize_tile_rows(int H, int W, int dh, int dw);
hipError_t launch_pp_resize_cubic(const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw, const int* x0, const double* wx, const long long* nx,
                                  const int* y0, const double* wy, const long long* ny, unsigned long long KX, unsigned long long KY, hipStream_t s, int src_bgr = 0);
hipError_t launch_pp_gauss3(const uint8_t* src, int H, int W, uint8_t* dst, int k0, int k1, int k2, unsigned long long* sum, hipStream_t s);
hipError_t launch_pp_clahe_hist(const uint8_t* src, int H, int W, const uint8_t* lut, int tw, int th, int tx, int ty, unsigned int* hist,
                                hipStream_t s);
hipError_t launch_pp_clahe_apply(const uint8_t* src, int H, int W, const uint8_t* lut, const uint8_t* tile_luts, int tw, int th, int tx, int ty,
                                 uint8_t* dst, hipStream_t s);
hipError_t launch_pp_box_pass(const uint8_t* src, uint8_t* dst, int H, int W, int vertical, int r, unsigned int ww, unsigned int fw, hipStream_t s);
hipError_t launch_pp_fold_lut(const unsigned long long* sum, unsigned long long n, float contrast, float brightness, uint8_t* lut, hipStream_t s);
hipError_t launch_pp_clahe_luts(const unsigned int* hist, int tiles, int clip, float lut_scale, uint8_t* tile_luts, hipStream_t s);
hipError_t launch_pp_lut(const uint8_t* src, uint8_t* dst, const uint8_t* lut, size_t total, hipStream_t s);
bool pp_unsharp_fused_ok(int H, int W, int r, const uint8_t* a, const uint8_t* b, const uint8_t* c);
hipError_t launch_pp_unsharp_fused(const uint8_t* in, uint8_t* tmp, uint8_t* dst, int H, int W, unsigned int ww, unsigned int fw, int percent,
                                   int threshold, hipStream_t s);
hipError_t launch_pp_unsharp(const uint8_t* in, const uint8_t* blur, uint8_t* dst, size_t total, int percent, int threshold, hipStream_t s);
