// Recogniser-side kernels other than the MFMA convs and the LSTM: box crops (cv2 bilinear / warpPerspective / PIL
// bicubic, all integer-exact restatements), contrast histogram, normalise+pad, CRNN conv0, 3-row mean.
// Upstream stages restated: easyocr/utils.py::{get_image_list,four_point_transform,compute_ratio_and_resize},
// easyocr/recognition.py::{AlignCollate,NormalizePAD,adjust_contrast_grey}, model/modules.py::VGG_FeatureExtractor
// (first conv) and vgg_model.py::Model (AdaptiveAvgPool2d((None,1))); reference call site
// pipeline_demo/extractor/enhanced_extractor.py:520.
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ warpPerspective
// cv2.warpPerspective(gray, M, (ww, wh)), INTER_LINEAR, BORDER_CONSTANT 0: coordinates rounded to 1/32 px (cvRound of a
// double), weights (32-ax)(32-ay)*32 etc. (exact 15-bit table entries), result (sum + 2^14) >> 15.
__global__ void __launch_bounds__(256) crop_warp_kernel(const uint8_t* __restrict__ gray, int H, int W, const CropDesc* __restrict__ descs,
                                                        int first, uint8_t* __restrict__ wscratch) {
    const CropDesc d = descs[first + blockIdx.y];
    if (!d.warp) return;
    const uint8_t* src = gray + (size_t)d.img * H * W;
    uint8_t* dst = wscratch + d.warp_off;
    const int total = d.sw * d.sh;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int x = i % d.sw, y = i / d.sw;
        const double X0 = d.Minv[0] * (double)x + (d.Minv[1] * (double)y + d.Minv[2]);
        const double Y0 = d.Minv[3] * (double)x + (d.Minv[4] * (double)y + d.Minv[5]);
        const double W0 = d.Minv[6] * (double)x + (d.Minv[7] * (double)y + d.Minv[8]);
        const double Wi = W0 != 0.0 ? 32.0 / W0 : 0.0;
        double fX = X0 * Wi, fY = Y0 * Wi;
        fX = fX < -2147483648.0 ? -2147483648.0 : (fX > 2147483647.0 ? 2147483647.0 : fX);
        fY = fY < -2147483648.0 ? -2147483648.0 : (fY > 2147483647.0 ? 2147483647.0 : fY);
        const long long X = (long long)rint(fX), Y = (long long)rint(fY);
        long long sxl = X >> 5, syl = Y >> 5;
        sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
        syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
        const int sx = (int)sxl, sy = (int)syl;
        const int ax = (int)(X & 31), ay = (int)(Y & 31);
        auto tap = [&](int yy, int xx) -> int {
            return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (int)src[(size_t)yy * W + xx] : 0;
        };
        const int acc = tap(sy, sx) * ((32 - ax) * (32 - ay) * 32) + tap(sy, sx + 1) * (ax * (32 - ay) * 32) +
                        tap(sy + 1, sx) * ((32 - ax) * ay * 32) + tap(sy + 1, sx + 1) * (ax * ay * 32);
        int v = (acc + (1 << 14)) >> 15;
        dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// ------------------------------------------------------------------------------------------------ cv2 bilinear crop
__device__ __forceinline__ void cv_lin_coef_d(int d, int ssize, double scale, int& s0, int& s1, int& a0, int& a1) {
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    a0 = __float2int_rn((1.f - fx) * 2048.f);
    a1 = __float2int_rn(fx * 2048.f);
    s0 = sx;
    s1 = sx + 1 < ssize ? sx + 1 : ssize - 1;
}

__global__ void __launch_bounds__(256) crop_resize_kernel(const uint8_t* __restrict__ gray, int H, int W, const CropDesc* __restrict__ descs,
                                                          int first, const uint8_t* __restrict__ wscratch, uint8_t* __restrict__ scratch) {
    const CropDesc d = descs[first + blockIdx.y];
    const uint8_t* src;
    int stride;
    if (d.warp) { src = wscratch + d.warp_off; stride = d.sw; }
    else { src = gray + (size_t)d.img * H * W + (size_t)d.sy0 * W + d.sx0; stride = W; }
    uint8_t* dst = scratch + d.a_off;
    // d.rw x d.rh is the stage-A image AFTER np.rot90(crop, d.rot) (rotation_info variants); cv2.resize works on the unrotated crop
    const int sw = d.sw, sh = d.sh, dw = (d.rot & 1) ? d.rh : d.rw, dh = (d.rot & 1) ? d.rw : d.rh;
    const bool same = (dw == sw && dh == sh), area2 = (sw == 2 * dw && sh == 2 * dh);
    const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
    const int total = dw * dh;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int dx = i % dw, dy = i / dw;
        int v;
        if (same) {
            v = src[(size_t)dy * stride + dx];
        } else if (area2) {
            v = (src[(size_t)(2 * dy) * stride + 2 * dx] + src[(size_t)(2 * dy) * stride + 2 * dx + 1] +
                 src[(size_t)(2 * dy + 1) * stride + 2 * dx] + src[(size_t)(2 * dy + 1) * stride + 2 * dx + 1] + 2) >> 2;
        } else {
            int x0, x1, a0, a1, y0, y1, b0, b1;
            cv_lin_coef_d(dx, sw, scale_x, x0, x1, a0, a1);
            cv_lin_coef_d(dy, sh, scale_y, y0, y1, b0, b1);
            const int r0 = src[(size_t)y0 * stride + x0] * a0 + src[(size_t)y0 * stride + x1] * a1;
            const int r1 = src[(size_t)y1 * stride + x0] * a0 + src[(size_t)y1 * stride + x1] * a1;
            v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        // np.rot90(m, k)[r][c]: k=1 m[c][W-1-r], k=2 m[H-1-r][W-1-c], k=3 m[H-1-c][r]
        int o = i;
        if (d.rot == 1) o = (dw - 1 - dx) * dh + dy;
        else if (d.rot == 2) o = (dh - 1 - dy) * dw + (dw - 1 - dx);
        else if (d.rot == 3) o = dx * dh + (dh - 1 - dy);
        dst[o] = (uint8_t)v;
    }
}

// 256-bin histogram of each stage-A crop (for np.percentile in adjust_contrast_grey; finished on the host in double)
__global__ void __launch_bounds__(256) crop_hist_kernel(const uint8_t* __restrict__ scratch, const CropDesc* __restrict__ descs, int first,
                                                        unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    const CropDesc d = descs[first + blockIdx.x];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* p = scratch + d.a_off;
    const int total = d.rw * d.rh;
    for (int i = threadIdx.x; i < total; i += 256) atomicAdd(&h[p[i]], 1u);
    __syncthreads();
    hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ PIL bicubic (8 bpc)
#define PIL_PREC 22
__device__ __forceinline__ double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
// filter window of output sample xx (Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc).  The taps are evaluated on the
// fly (pil_tap), not stored: a rotation_info variant of a long line shrinks thousands of rows to 64, i.e. hundreds of taps per sample
struct PilWin { int x0, n; double center, ss, ww; };
__device__ PilWin pil_window(int in_size, int out_size, int xx) {
    double scale = (double)in_size / (double)out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    PilWin w;
    w.center = ((double)xx + 0.5) * scale;
    w.ss = 1.0 / filterscale;
    int x0 = (int)(w.center - support + 0.5);
    if (x0 < 0) x0 = 0;
    int x1 = (int)(w.center + support + 0.5);
    if (x1 > in_size) x1 = in_size;
    w.x0 = x0;
    w.n = x1 - x0;
    w.ww = 0.0;
    for (int x = 0; x < w.n; ++x) w.ww += pil_bicubic(((double)(x + x0) - w.center + 0.5) * w.ss);
    return w;
}
__device__ __forceinline__ int pil_tap(const PilWin& w, int x) {
    double k = pil_bicubic(((double)(x + w.x0) - w.center + 0.5) * w.ss);
    if (w.ww != 0.0) k = k / w.ww;
    return k < 0 ? (int)(-0.5 + k * (double)(1 << PIL_PREC)) : (int)(0.5 + k * (double)(1 << PIL_PREC));
}
__device__ __forceinline__ int pil_clip8(long long v) {
    const long long r = v >> PIL_PREC;
    return r < 0 ? 0 : (r > 255 ? 255 : (int)r);
}

// stage B0 (tall boxes only): horizontal PIL pass  [rh][rw] -> [rh][fw] uint8 into hscratch
__global__ void __launch_bounds__(256) crop_pil_h_kernel(const CropDesc* __restrict__ descs, int first, const uint8_t* __restrict__ scratch,
                                                         const uint8_t* __restrict__ luts, uint8_t* __restrict__ hscratch) {
    const CropDesc d = descs[first + blockIdx.y];
    if (d.fw == d.rw && d.rh == 64) return;
    const uint8_t* src = scratch + d.a_off;
    const uint8_t* lut = d.lut_off >= 0 ? luts + d.lut_off : nullptr;
    uint8_t* dst = hscratch + d.a_off;    // same offsets: fw <= rw so the region fits
    const int total = d.rh * d.fw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int xx = i % d.fw, y = i / d.fw;
        int v;
        if (d.fw == d.rw) {
            v = src[(size_t)y * d.rw + xx];
            if (lut) v = lut[v];
        } else {
            const PilWin win = pil_window(d.rw, d.fw, xx);
            long long acc = 1LL << (PIL_PREC - 1);
            for (int k = 0; k < win.n; ++k) {
                int p = src[(size_t)y * d.rw + win.x0 + k];
                if (lut) p = lut[p];
                acc += (long long)p * pil_tap(win, k);
            }
            v = pil_clip8(acc);
        }
        dst[i] = (uint8_t)v;
    }
}

// stage B: (vertical PIL pass for tall boxes |  plain copy) + contrast LUT + ToTensor/normalise + right edge-replicate pad
//   out bf16 [slot][64][imgW]
// Output addressing: bucket tensor [slot][64][imgW] (row_stride = imgW, slot_stride = 64*imgW, gap = 0) or ONE wide image
// [64][Wt] holding every crop side by side (row_stride = Wt, slot_stride = 1, slot = the crop's first column, gap zero columns
// after it: the conv layers' zero padding between neighbours).
template <int MODE>
__global__ void __launch_bounds__(256) crop_final_kernel(const CropDesc* __restrict__ descs, int first, const uint8_t* __restrict__ scratch,
                                                         const uint8_t* __restrict__ hscratch, const uint8_t* __restrict__ luts,
                                                         uint16_t* __restrict__ out, int row_stride, long long slot_stride, int gap) {
    const CropDesc d = descs[first + blockIdx.y];
    const bool tall = !(d.fw == d.rw && d.rh == 64);
    const uint8_t* src = (tall ? hscratch : scratch) + d.a_off;
    const uint8_t* lut = (!tall && d.lut_off >= 0) ? luts + d.lut_off : nullptr;
    uint16_t* dst = out + (size_t)d.slot * (size_t)slot_stride;
    const int imgW = d.imgW;
    // 16 x 256 threads per crop = 64 rows x 64 threads: thread -> (row y, column phase), columns strided by 64: no per-element
    // division, the vertical PIL coefficients (tall boxes) once per row
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int y = t >> 6, x0 = t & 63;
    if (y >= 64) return;
    const bool vert = tall && d.rh != 64;
    PilWin win;
    if (vert) win = pil_window(d.rh, 64, y);
    auto sample = [&](int xs) {
        int v;
        if (!tall) {
            v = src[(size_t)y * d.rw + xs];
            if (lut) v = lut[v];
        } else if (!vert) {
            v = src[(size_t)y * d.fw + xs];
        } else {
            long long acc = 1LL << (PIL_PREC - 1);
            for (int k = 0; k < win.n; ++k) acc += (long long)src[(size_t)(win.x0 + k) * d.fw + xs] * pil_tap(win, k);
            v = pil_clip8(acc);
        }
        if constexpr (MODE == REC_SPLIT) return (unsigned short)(v + 1);          // code: conv0 rebuilds the fp32 value exactly
        else return El<MODE == REC_F16 ? 1 : 0>::from_f32(((float)v / 255.0f - 0.5f) / 0.5f);
    };
    uint16_t* drow = dst + (size_t)y * row_stride;
    for (int x = imgW + x0; x < imgW + gap; x += 64) drow[x] = 0;     // zero separator columns (wide layout)
    int x = x0;
    for (; x < d.fw; x += 64) drow[x] = sample(x);
    if (x < imgW) {                                                  // NormalizePAD: the last content column, replicated
        const uint16_t edge = sample(d.fw - 1);
        for (; x < imgW; x += 64) drow[x] = edge;
    }
}

hipError_t launch_crops(const uint8_t* gray, int H, int W, const CropDesc* descs_dev, int first, int count, int imgW, int any_warp,
                        int any_tall, uint8_t* wscratch, uint8_t* scratch, uint8_t* hscratch, const uint8_t* luts, uint16_t* out_bucket,
                        int stage_mask, hipStream_t s, int wide_row_stride, int gap, int mode) {
    if (count <= 0) return hipSuccess;
    if (stage_mask & 1) {
        if (any_warp) hipLaunchKernelGGL(crop_warp_kernel, dim3(16, count), dim3(256), 0, s, gray, H, W, descs_dev, first, wscratch);
        hipLaunchKernelGGL(crop_resize_kernel, dim3(16, count), dim3(256), 0, s, gray, H, W, descs_dev, first, wscratch, scratch);
    }
    if (stage_mask & 2) {
        if (any_tall) hipLaunchKernelGGL(crop_pil_h_kernel, dim3(8, count), dim3(256), 0, s, descs_dev, first, scratch, luts, hscratch);
        const int rs = wide_row_stride > 0 ? wide_row_stride : imgW;
        const long long ss = wide_row_stride > 0 ? 1LL : (long long)64 * imgW;
        const int gp = wide_row_stride > 0 ? gap : 0;
        if (mode == REC_SPLIT)
            hipLaunchKernelGGL(crop_final_kernel<REC_SPLIT>, dim3(16, count), dim3(256), 0, s, descs_dev, first, scratch, hscratch, luts, out_bucket, rs, ss, gp);
        else if (mode == REC_F16)
            hipLaunchKernelGGL(crop_final_kernel<REC_F16>, dim3(16, count), dim3(256), 0, s, descs_dev, first, scratch, hscratch, luts, out_bucket, rs, ss, gp);
        else
            hipLaunchKernelGGL(crop_final_kernel<REC_BF16>, dim3(16, count), dim3(256), 0, s, descs_dev, first, scratch, hscratch, luts, out_bucket, rs, ss, gp);
    }
    return hipGetLastError();
}

hipError_t launch_crop_hist(const uint8_t* scratch, const CropDesc* descs_dev, int first, int count, unsigned int* hist, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(crop_hist_kernel, dim3(count), dim3(256), 0, s, scratch, descs_dev, first, hist);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ CRNN conv0 + pool
// FeatureExtraction.ConvNet.0 (1->32, 3x3, pad 1) + ReLU + MaxPool2d(2,2): one pooled pixel x 32 channels per thread.
template <int MODE>
__global__ void __launch_bounds__(256) crnn_conv0_kernel(const uint16_t* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ b, uint16_t* __restrict__ out, int n, int W) {
    const int OW = W / 2, OH = 32;
    const size_t total = (size_t)n * OH * OW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        const size_t r = i / OW;
        const int oy = (int)(r % OH);
        const int img = (int)(r / OH);
        const uint16_t* p = in + (size_t)img * 64 * W;
        float v[4][4];   // 4x4 input window around the 2x2 conv outputs
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int iy = 2 * oy - 1 + dy, ix = 2 * ox - 1 + dx;
                float x = 0.f;
                if (iy >= 0 && iy < 64 && ix >= 0 && ix < W) {
                    const unsigned short q = p[(size_t)iy * W + ix];
                    if constexpr (MODE == REC_SPLIT) x = q ? ((float)(q - 1) / 255.0f - 0.5f) / 0.5f : 0.f;      // ToTensor + sub_(0.5).div_(0.5) in fp32
                    else x = El<MODE == REC_F16 ? 1 : 0>::to_f32(q);
                }
                v[dy][dx] = x;
            }
        uint16_t* op = out + i * (MODE == REC_SPLIT ? 64 : 32);
#pragma unroll 1
        for (int c8 = 0; c8 < 32; c8 += 8) {        // 8 channels -> one 16-byte store
            u32x4 o, ol;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                // two channels at a time on the packed fp32 pipe (v_pk_fma_f32): w is tap-major [9][32], so a channel pair of
                // one tap is one 8-byte scalar load and the input value is broadcast to both halves.  (Keeping all 64 pair
                // accumulators live with tap-outer loops and LDS-broadcast weights was measured 1.6x SLOWER.)
                const int c = c8 + c2 * 2;
                const f32x2_t bias = {b[c], b[c + 1]};
                f32x2_t best = {0.f, 0.f};   // ReLU floor
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        f32x2_t a = bias;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const f32x2_t wv = *(const f32x2_t*)(w + (ky * 3 + kx) * 32 + c);
                                const float x = v[py + ky][px + kx];
                                a = __builtin_elementwise_fma(wv, (f32x2_t){x, x}, a);
                            }
                        best = __builtin_elementwise_max(best, a);
                    }
                if constexpr (MODE == REC_SPLIT) {
                    const float h0 = El<1>::to_f32(El<1>::from_f32(best[0])), h1 = El<1>::to_f32(El<1>::from_f32(best[1]));
                    o[c2] = El<1>::pack2(h0, h1);
                    ol[c2] = El<1>::pack2((best[0] - h0) * SPLIT_LO_SCALE, (best[1] - h1) * SPLIT_LO_SCALE);
                } else {
                    o[c2] = El<MODE == REC_F16 ? 1 : 0>::pack2(best[0], best[1]);
                }
            }
            *(u32x4*)(op + c8) = o;
            if constexpr (MODE == REC_SPLIT) *(u32x4*)(op + 32 + c8) = ol;
        }
    }
}

// The same layer on the MFMA pipe (bf16 / fp16 modes): the VALU kernel above runs at the packed-fp32 issue rate (v_pk_fma_f32 issues at half
// rate: 2,900 of them per wave), while the 9-tap x 32-channel product is a 16-cout x 16-pixel x K MFMA with K to spare.  The input pixels are
// element-type values already (crop_final rounds them); the fp32 weights are split THREE ways into element-type terms w = hi + lo + lo2
// (24+ significand bits), so every product is exact in fp32 and the result differs from the FMA chain only by the order of fp32 additions:
//   K = 32: lane group 0 / 1 / 2 = taps 0..7 x (hi / lo / lo2), group 3 unused (zero weights); tap 8 is one fp32 FMA per output value.
// B fragment of a 16-pixel row segment: 8 two-byte LDS reads per lane at compile-time offsets from the lane's pixel, the same for every lane.
// A wave owns row PAIRS (2r, 2r+1): the 2x2 pool is one v_max3 across the pair's two fragments (with the ReLU floor) and one DPP max across
// neighbouring pixels; a lane's 8 pooled values are 8 consecutive channels = one 16-byte store.
void pack_crnn_conv0_mfma(const float* w_tap_major, uint16_t* out, int el) {
    size_t o = 0;
    for (int j = 0; j < 2; ++j)
        for (int l = 0; l < 64; ++l) {
            const int m = l & 15, kg = l >> 4;
            const int cout = 8 * (m >> 2) + 4 * j + (m & 3);     // D row 4g + r of fragment j <-> channel 8g + 4j + r
            for (int t = 0; t < 8; ++t) {
                const float v = w_tap_major[t * 32 + cout];
                const float hi = el_to_f32_host(el, f32_to_el_host(el, v));
                const float lo = el_to_f32_host(el, f32_to_el_host(el, v - hi));
                const float lo2 = el_to_f32_host(el, f32_to_el_host(el, (v - hi) - lo));
                out[o++] = f32_to_el_host(el, kg == 0 ? hi : (kg == 1 ? lo : (kg == 2 ? lo2 : 0.f)));
            }
        }
}

template <int EL>
__global__ void __launch_bounds__(256) crnn_conv0_mfma_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ afrag, const float* __restrict__ w,
                                                              const float* __restrict__ b, uint16_t* __restrict__ out, int W) {
    constexpr int TW = 64, PITCH = TW + 16, ROWS = 66;            // LDS tile: rows -1 .. 64, columns x0 - 8 .. x0 + 71 (16-byte chunks)
    __shared__ __attribute__((aligned(16))) uint16_t tile[ROWS * PITCH];
    typedef typename El<EL>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int x0 = blockIdx.x * TW, img = blockIdx.y;
    const int OW = W >> 1;
    const uint16_t* src = in + (size_t)img * 64 * W;
    for (int idx = tid; idx < ROWS * (PITCH / 4); idx += 256) {
        const int row = idx / (PITCH / 4), ch = idx - row * (PITCH / 4);
        const int iy = row - 1, ix = x0 - 8 + ch * 4;
        u32x2 v = {0u, 0u};
        if (iy >= 0 && iy < 64 && ix >= 0 && ix + 4 <= W) v = *(const u32x2*)(src + (size_t)iy * W + ix);      // (W % 4 == 0: whole 8-byte chunks only)
        *(u32x2*)(tile + row * PITCH + ch * 4) = v;
    }
    const v8 a0 = *(const v8*)(afrag + (size_t)lane * 8), a1 = *(const v8*)(afrag + (size_t)(64 + lane) * 8);
    float w8[8], bi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        w8[i] = w[8 * 32 + 8 * g + i];
        bi[i] = b[8 * g + i];
    }
    __syncthreads();
    auto frag = [&](int y, int cf, f32x4& d0, f32x4& d1) {
        const uint16_t* p = tile + (y + 1) * PITCH + 8 + cf * 16 + n;       // the lane's pixel
        u32x4 bq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t0 = 2 * i, t1 = 2 * i + 1;
            const unsigned int lo = p[(t0 / 3 - 1) * PITCH + (t0 % 3 - 1)], hi = p[(t1 / 3 - 1) * PITCH + (t1 % 3 - 1)];
            bq[i] = lo | (hi << 16);
        }
        const float x8 = El<EL>::to_f32(p[PITCH + 1]);
        d0 = El<EL>::mfma(a0, __builtin_bit_cast(v8, bq), (f32x4){bi[0], bi[1], bi[2], bi[3]});
        d1 = El<EL>::mfma(a1, __builtin_bit_cast(v8, bq), (f32x4){bi[4], bi[5], bi[6], bi[7]});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d0[r] = fmaf(w8[r], x8, d0[r]);
            d1[r] = fmaf(w8[4 + r], x8, d1[r]);
        }
    };
#pragma unroll 1
    for (int rp = 0; rp < 8; ++rp) {
        const int y = 2 * (wave * 8 + rp);
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
            f32x4 u0, u1, l0, l1;
            frag(y, cf, u0, u1);
            frag(y + 1, cf, l0, l1);
            float m[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                m[r] = __builtin_fmaxf(__builtin_fmaxf(u0[r], l0[r]), 0.f);          // v_max3: the row pair and the ReLU floor
                m[4 + r] = __builtin_fmaxf(__builtin_fmaxf(u1[r], l1[r]), 0.f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                m[i] = fmaxf(m[i], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[i]), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true)));
            const int px = (x0 + cf * 16 + n) >> 1;
            if (!(n & 1) && px < OW) {
                const u32x4 o = {El<EL>::pack2(m[0], m[1]), El<EL>::pack2(m[2], m[3]), El<EL>::pack2(m[4], m[5]), El<EL>::pack2(m[6], m[7])};
                *(u32x4*)(out + (((size_t)img * 32 + (y >> 1)) * OW + px) * 32 + 8 * g) = o;
            }
        }
    }
}

hipError_t launch_crnn_conv0(const uint16_t* in, const float* w, const float* b, uint16_t* out, int n, int W, int mode, hipStream_t s,
                             const uint16_t* afrag) {
    const size_t total = (size_t)n * 32 * (W / 2);
    if (total == 0) return hipSuccess;
    if (afrag && mode != REC_SPLIT && (W & 3) == 0 && n <= 65535) {
        const dim3 grid((W + 63) / 64, n);
        if (mode == REC_F16) hipLaunchKernelGGL(crnn_conv0_mfma_kernel<1>, grid, dim3(256), 0, s, in, afrag, w, b, out, W);
        else hipLaunchKernelGGL(crnn_conv0_mfma_kernel<0>, grid, dim3(256), 0, s, in, afrag, w, b, out, W);
        return hipGetLastError();
    }
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (mode == REC_SPLIT) hipLaunchKernelGGL(crnn_conv0_kernel<REC_SPLIT>, dim3(grid), dim3(256), 0, s, in, w, b, out, n, W);
    else if (mode == REC_F16) hipLaunchKernelGGL(crnn_conv0_kernel<REC_F16>, dim3(grid), dim3(256), 0, s, in, w, b, out, n, W);
    else hipLaunchKernelGGL(crnn_conv0_kernel<REC_BF16>, dim3(grid), dim3(256), 0, s, in, w, b, out, n, W);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ mean over 3 rows
// AdaptiveAvgPool2d((None, 1)) over the 3 feature rows for 8 channels: p0 / p1 / p2 point at the 8 channels of the three rows.
// REC_SPLIT: the rows are [hi C | lo C] pairs (lo_off elements apart), the mean is taken on hi + lo/2048 in fp32 and split again.
template <int MODE>
__device__ __forceinline__ void rowmean3_8(const uint16_t* p0, const uint16_t* p1, const uint16_t* p2, uint16_t* o, int lo_off) {
    typedef El<MODE == REC_BF16 ? 0 : 1> E;
    const u32x4 a = *(const u32x4*)(p0), bq = *(const u32x4*)(p1), c = *(const u32x4*)(p2);
    u32x4 oh, ol;
    if constexpr (MODE == REC_SPLIT) {
        const u32x4 al = *(const u32x4*)(p0 + lo_off), bl = *(const u32x4*)(p1 + lo_off), cl = *(const u32x4*)(p2 + lo_off);
        const float inv = 1.0f / SPLIT_LO_SCALE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2_t m = ((E::unpack2(a[j]) + E::unpack2(al[j]) * inv) + (E::unpack2(bq[j]) + E::unpack2(bl[j]) * inv) +
                               (E::unpack2(c[j]) + E::unpack2(cl[j]) * inv)) / 3.0f;
            const f32x2_t h = E::unpack2(E::pack2(m[0], m[1]));
            oh[j] = E::pack2(h[0], h[1]);
            ol[j] = E::pack2((m[0] - h[0]) * SPLIT_LO_SCALE, (m[1] - h[1]) * SPLIT_LO_SCALE);
        }
        *(u32x4*)(o) = oh;
        *(u32x4*)(o + lo_off) = ol;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2_t m = (E::unpack2(a[j]) + E::unpack2(bq[j]) + E::unpack2(c[j])) / 3.0f;
            oh[j] = E::pack2(m[0], m[1]);
        }
        *(u32x4*)(o) = oh;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) rowmean3_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int n, int T, int C8) {
    constexpr int M = MODE == REC_SPLIT ? 2 : 1;             // stored channels per logical channel
    const size_t total = (size_t)n * T * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t per = (size_t)T * C8;
        const size_t img = i / per, rem = i - img * per;
        const size_t t = rem / C8, c8 = rem - t * C8;
        const size_t row = (size_t)T * C8 * 8 * M;           // elements of one feature row
        const uint16_t* p = in + img * 3 * row + (t * C8 * M + c8) * 8;
        rowmean3_8<MODE>(p, p + row, p + 2 * row, out + ((img * T + t) * C8 * M + c8) * 8, C8 * 8);
    }
}

hipError_t launch_rowmean3(const uint16_t* in, uint16_t* out, int n, int T, int C, int mode, hipStream_t s) {
    const size_t total = (size_t)n * T * (C / 8);
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (mode == REC_SPLIT) hipLaunchKernelGGL(rowmean3_kernel<REC_SPLIT>, dim3(grid), dim3(256), 0, s, in, out, n, T, C / 8);
    else if (mode == REC_F16) hipLaunchKernelGGL(rowmean3_kernel<REC_F16>, dim3(grid), dim3(256), 0, s, in, out, n, T, C / 8);
    else hipLaunchKernelGGL(rowmean3_kernel<REC_BF16>, dim3(grid), dim3(256), 0, s, in, out, n, T, C / 8);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ wide recogniser image helpers
// All crops of a recognition pass sit side by side in ONE image (crop_final's wide layout), so every CRNN layer is one launch
// over [H, Wt] instead of one per width bucket.  The columns between two crops are that layer's zero padding: a conv writes
// neighbour-dependent values there, so they are cleared again on every layer's output (a few columns per crop).
// shift: log2 of the layer's horizontal down-scale (1 after conv0's pool, 2 after r1's), gapw = 4 >> shift columns.
__global__ void __launch_bounds__(256) crnn_zero_gaps_kernel(uint16_t* __restrict__ t, const CropDesc* __restrict__ descs, int first, int H, int Wl,
                                                             int C8, int shift, int gapw) {
    const CropDesc d = descs[first + blockIdx.x];
    const int x0 = (d.slot + d.imgW) >> shift;
    const int total = H * gapw * C8;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c8 = i % C8, r = i / C8;
        const int gx = r % gapw, y = r / gapw;
        if (x0 + gx < Wl) *(u32x4*)(t + ((size_t)y * Wl + x0 + gx) * C8 * 8 + (size_t)c8 * 8) = (u32x4){0u, 0u, 0u, 0u};
    }
}
hipError_t launch_crnn_zero_gaps(uint16_t* t, const CropDesc* descs_dev, int first, int count, int H, int Wl, int C, int shift, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(crnn_zero_gaps_kernel, dim3(count), dim3(256), 0, s, t, descs_dev, first, H, Wl, C / 8, shift, 4 >> shift);
    return hipGetLastError();
}

// AdaptiveAvgPool over the 3 feature rows + gather: wide features [3][Wc][C] -> pooled rows [row0 + t][C] of every crop
// (columns slot/4 .. slot/4 + T - 1, T = imgW/4 - 1, row0 = CropDesc::pad_)
template <int MODE>
__global__ void __launch_bounds__(256) rowmean3_gather_kernel(const uint16_t* __restrict__ in, int Wc, int C8, const CropDesc* __restrict__ descs,
                                                              int first, uint16_t* __restrict__ out) {
    constexpr int M = MODE == REC_SPLIT ? 2 : 1;
    const CropDesc d = descs[first + blockIdx.y];
    const int T = d.imgW / 4 - 1, xs = d.slot >> 2;
    const size_t plane = (size_t)Wc * C8 * 8 * M;
    const int total = T * C8;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int t = i / C8, c8 = i - t * C8;
        const uint16_t* p = in + ((size_t)(xs + t) * C8 * M + c8) * 8;
        rowmean3_8<MODE>(p, p + plane, p + 2 * plane, out + ((size_t)(d.pad_ + t) * C8 * M + c8) * 8, C8 * 8);
    }
}
hipError_t launch_rowmean3_gather(const uint16_t* in, int Wc, int C, const CropDesc* descs_dev, int first, int count, uint16_t* out, int mode,
                                  hipStream_t s) {
    if (count <= 0) return hipSuccess;
    if (mode == REC_SPLIT) hipLaunchKernelGGL(rowmean3_gather_kernel<REC_SPLIT>, dim3(8, count), dim3(256), 0, s, in, Wc, C / 8, descs_dev, first, out);
    else if (mode == REC_F16) hipLaunchKernelGGL(rowmean3_gather_kernel<REC_F16>, dim3(8, count), dim3(256), 0, s, in, Wc, C / 8, descs_dev, first, out);
    else hipLaunchKernelGGL(rowmean3_gather_kernel<REC_BF16>, dim3(8, count), dim3(256), 0, s, in, Wc, C / 8, descs_dev, first, out);
    return hipGetLastError();
}
