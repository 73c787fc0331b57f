// Detector-side kernels that are not the generic implicit-GEMM conv: conv1_1 weight packing, max-pool, gray conversion and the
// cv2-style uint8 bilinear resize.
// Upstream stages restated: easyocr/imgproc.py::{resize_aspect_ratio,normalizeMeanVariance}, craft.py::CRAFT.forward,
// utils.py::reformat_input (reference call site pipeline_demo/extractor/enhanced_extractor.py:520).
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ conv1_1 weights
// conv1_1 (3 -> 64) lives inside conv1_2's prologue (conv_mfma.hip, FUSE1): K is laid out as tap*4 + channel (channel 3 = 0) and
// padded 36 -> 64 (two MFMA k-steps); fragment nf, row -> cout (nf>>1)*32 + (row>>2)*8 + (nf&1)*4 + (row&3), the run order of the
// conv epilogue.  el: element type of the packed fragments (0 bf16, 1 fp16).
void pack_conv1_1_weights_fused(const float* w, uint16_t* out, int el) {
    // ONE 32-deep k-step for the 27 (tap, channel) products: lane group g carries taps 2g and 2g+1 as [r g b X | r g b Y] -- the patch
    // pixels are stored as four 16-bit values r g b 0 -- and the ninth tap rides in the pad slots: X(g=0) = tap 8 r, Y(g=0) = tap 8 g,
    // X(g=1) = tap 8 b (conv3x3_dma_kernel<FUSE1> builds the matching B fragment)
    size_t o = 0;
    for (int nf = 0; nf < 4; ++nf)
        for (int l = 0; l < 64; ++l) {
            const int row = l & 15, g = l >> 4;
            const int cout = (nf >> 1) * 32 + (row >> 2) * 8 + (nf & 1) * 4 + (row & 3);
            for (int j = 0; j < 8; ++j) {
                int tap = 2 * g + (j >> 2), ch = j & 3;
                if (ch == 3) {
                    tap = 8;
                    ch = (g == 0) ? (j == 3 ? 0 : 1) : ((g == 1 && j == 3) ? 2 : -1);
                }
                const float v = ch >= 0 ? w[((size_t)cout * 3 + ch) * 9 + tap] : 0.f;
                out[o++] = f32_to_el_host(el, v);
            }
        }
}

// ------------------------------------------------------------------------------------------------ max-pool
// NHWC 16-bit floats (bf16 or fp16: both sign-magnitude, compared through sm16_key), 8 channels (16 B) per thread; optional ReLU on
// the input (max(relu(x)) == relu(max(x))).
__global__ void __launch_bounds__(256) maxpool_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int N, int H, int W,
                                                      int C8, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw,
                                                      int relu_in) {
    const size_t total = (size_t)N * OH * OW * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        size_t r = i / C8;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        const short lowest = relu_in ? (short)0 : (short)-32768;         // key of +0 / below every finite value
        s16x8 m = {lowest, lowest, lowest, lowest, lowest, lowest, lowest, lowest};
        for (int dy = 0; dy < kh; ++dy) {
            const int iy = oy * sh - ph + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < kw; ++dx) {
                const int ix = ox * sw - pw + dx;
                if (ix < 0 || ix >= W) continue;
                const s16x8 v = *(const s16x8*)(in + (((size_t)(n * H + iy) * W + ix) * C8 + c8) * 8);
                m = __builtin_elementwise_max(m, sm16_key(v));
            }
        }
        *(s16x8*)(out + i * 8) = sm16_key(m);
    }
}

// MaxPool2d(3, stride 1, padding 1) (CRAFT's pool5 in front of fc6): a thread owns a column strip of four output rows -- the 3-wide row
// maxima of its six input rows are computed once (18 loads for 4 outputs instead of 36)
__global__ void __launch_bounds__(256) maxpool3x3s1_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int N, int H, int W, int C8,
                                                           int relu_in) {
    const int HG = (H + 3) >> 2;
    const size_t total = (size_t)N * HG * W * C8;
    const short lowest = relu_in ? (short)0 : (short)-32768;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        size_t r = i / C8;
        const int ox = (int)(r % W);
        r /= W;
        const int oy0 = (int)(r % HG) * 4;
        const int n = (int)(r / HG);
        s16x8 rm[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int iy = oy0 - 1 + k;
            s16x8 m = {lowest, lowest, lowest, lowest, lowest, lowest, lowest, lowest};
            if (iy >= 0 && iy < H) {
                const uint16_t* row = in + (((size_t)(n * H + iy) * W) * C8 + c8) * 8;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int ix = ox + dx;
                    if (ix >= 0 && ix < W) m = __builtin_elementwise_max(m, sm16_key(*(const s16x8*)(row + (size_t)ix * C8 * 8)));
                }
            }
            rm[k] = m;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = oy0 + k;
            if (oy < H)
                *(s16x8*)(out + (((size_t)(n * H + oy) * W + ox) * C8 + c8) * 8) =
                    sm16_key(__builtin_elementwise_max(__builtin_elementwise_max(rm[k], rm[k + 1]), rm[k + 2]));
        }
    }
}

hipError_t launch_maxpool(const uint16_t* in, uint16_t* out, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                          int relu_in, hipStream_t s) {
    if (C & 7) return hipErrorInvalidValue;
    if (kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1) {
        const size_t tot = (size_t)N * ((H + 3) / 4) * W * (C / 8);
        const int g = (int)((tot + 255) / 256 < 16384 ? (tot + 255) / 256 : 16384);
        hipLaunchKernelGGL(maxpool3x3s1_kernel, dim3(g > 0 ? g : 1), dim3(256), 0, s, in, out, N, H, W, C / 8, relu_in);
        return hipGetLastError();
    }
    const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
    const size_t total = (size_t)N * OH * OW * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, in, out, N, H, W, C / 8, OH, OW, kh, kw, sh, sw, ph, pw,
                       relu_in);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ gray
// cv2 BGR2GRAY applied to channels as given (upstream applies it to whatever 3-channel array it holds): OpenCV 4's 15-bit fixed
// point (R 9798, G 19235, B 3735, round to nearest; pinned by the reference's stored pre-processing outputs, tests/golden/legacy_preprocess).
__global__ void __launch_bounds__(256) gray_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ gray, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const int c0 = rgb[i * 3], c1 = rgb[i * 3 + 1], c2 = rgb[i * 3 + 2];
        gray[i] = (uint8_t)((c2 * 9798 + c1 * 19235 + c0 * 3735 + (1 << 14)) >> 15);
    }
}
// ---- libjpeg's YCbCr -> RGB (jdcolor.c::ycc_rgb_convert, the conversion behind every RGB decode of a JFIF file: cv2.imread, skimage /
// PIL).  A JPEG decoded ONCE with out_color_space = JCS_YCbCr holds both planes upstream's path branch reads: its Y channel IS
// cv2.imread(IMREAD_GRAYSCALE)'s plane (libjpeg decodes component 0 alone for JCS_GRAYSCALE, same IDCT), and the RGB image is this
// pointwise integer map of the (fancy-upsampled) triple.  16-bit fixed point, FIX(x) = (int)(x * 65536 + 0.5), arithmetic shifts:
//   R = y + ((FIX(1.40200) * (cr - 128) + 2^15) >> 16)
//   G = y + ((-FIX(0.34414) * (cb - 128) + 2^15 - FIX(0.71414) * (cr - 128)) >> 16)
//   B = y + ((FIX(1.77200) * (cb - 128) + 2^15) >> 16), each clamped to 0..255 (range_limit).
// `stride` = bytes per source pixel: 3 (tight triples) or 4 (Pillow's own pixel storage, Y Cb Cr x, taken zero-copy: one dword load)
__global__ void __launch_bounds__(256) ycc_to_rgb_gray_kernel(const uint8_t* __restrict__ ycc, int stride, uint8_t* __restrict__ rgb,
                                                               uint8_t* __restrict__ gray, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        int y, cb, cr;
        if (stride == 4) {
            const unsigned int p = ((const unsigned int*)ycc)[i];
            y = (int)(p & 255u); cb = (int)((p >> 8) & 255u) - 128; cr = (int)((p >> 16) & 255u) - 128;
        } else {
            y = ycc[i * 3]; cb = (int)ycc[i * 3 + 1] - 128; cr = (int)ycc[i * 3 + 2] - 128;
        }
        int r = y + ((91881 * cr + 32768) >> 16);
        int g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        int b = y + ((116130 * cb + 32768) >> 16);
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        g = g < 0 ? 0 : (g > 255 ? 255 : g);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        rgb[i * 3] = (uint8_t)r;
        rgb[i * 3 + 1] = (uint8_t)g;
        rgb[i * 3 + 2] = (uint8_t)b;
        if (gray) gray[i] = (uint8_t)y;
    }
}
hipError_t launch_ycc_to_rgb_gray(const uint8_t* ycc, int stride, uint8_t* rgb, uint8_t* gray, size_t npix, hipStream_t s) {
    if ((stride != 3 && stride != 4) || (stride == 4 && ((size_t)ycc & 3))) return hipErrorInvalidValue;
    const int grid = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    hipLaunchKernelGGL(ycc_to_rgb_gray_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, ycc, stride, rgb, gray, npix);
    return hipGetLastError();
}
hipError_t launch_gray(const uint8_t* rgb, uint8_t* gray, size_t npix, hipStream_t s) {
    const int grid = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    hipLaunchKernelGGL(gray_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, rgb, gray, npix);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ cv2 bilinear (uint8)
// cv::resize INTER_LINEAR, 8-bit: 11-bit fixed-point coefficients (cvRound of float), horizontal pass in int32,
// vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2; exact 2x decimation takes the INTER_AREA path.
__device__ __forceinline__ void cv_lin_coef(int d, int ssize, double scale, int& s0, int& s1, int& a0, int& a1) {
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

__global__ void __launch_bounds__(256) resize_u8_kernel(const uint8_t* __restrict__ src, int N, int sh, int sw, int C,
                                                        uint8_t* __restrict__ dst, int dh, int dw) {
    const size_t total = (size_t)N * dh * dw;
    const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
    const bool same = (dh == sh && dw == sw), area2 = (sw == 2 * dw && sh == 2 * dh);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int dx = (int)(i % dw);
        const size_t r = i / dw;
        const int dy = (int)(r % dh);
        const int n = (int)(r / dh);
        const uint8_t* sp = src + (size_t)n * sh * sw * C;
        uint8_t* dp = dst + i * C;
        if (same) {
            for (int c = 0; c < C; ++c) dp[c] = sp[((size_t)dy * sw + dx) * C + c];
        } else if (area2) {
            for (int c = 0; c < C; ++c) {
                const int v = sp[((size_t)(2 * dy) * sw + 2 * dx) * C + c] + sp[((size_t)(2 * dy) * sw + 2 * dx + 1) * C + c] +
                              sp[((size_t)(2 * dy + 1) * sw + 2 * dx) * C + c] + sp[((size_t)(2 * dy + 1) * sw + 2 * dx + 1) * C + c];
                dp[c] = (uint8_t)((v + 2) >> 2);
            }
        } else {
            int x0, x1, a0, a1, y0, y1, b0, b1;
            cv_lin_coef(dx, sw, scale_x, x0, x1, a0, a1);
            cv_lin_coef(dy, sh, scale_y, y0, y1, b0, b1);
            for (int c = 0; c < C; ++c) {
                const int r0 = sp[((size_t)y0 * sw + x0) * C + c] * a0 + sp[((size_t)y0 * sw + x1) * C + c] * a1;
                const int r1 = sp[((size_t)y1 * sw + x0) * C + c] * a0 + sp[((size_t)y1 * sw + x1) * C + c] * a1;
                int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                dp[c] = (uint8_t)v;
            }
        }
    }
}

hipError_t launch_resize_u8(const uint8_t* src, int N, int sh, int sw, int C, uint8_t* dst, int dh, int dw, hipStream_t s) {
    const size_t total = (size_t)N * dh * dw;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, src, N, sh, sw, C, dst, dh, dw);
    return hipGetLastError();
}
