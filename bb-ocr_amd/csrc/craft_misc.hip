// Detector-side kernels that are not the generic implicit-GEMM conv: fused normalise+conv1_1, max-pool,
// bilinear x2 up-sampling, gray conversion and the cv2-style uint8 bilinear resize.
// Upstream stages restated: easyocr/imgproc.py::{resize_aspect_ratio,normalizeMeanVariance}, craft.py::CRAFT.forward,
// utils.py::reformat_input (reference call site pipeline_demo/extractor/enhanced_extractor.py:520).
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ conv1_1
// K is laid out as tap*4 + channel (channel 3 = 0) and padded 36 -> 64, so one pixel's B fragment is two 8-byte LDS
// reads (2 taps x 4 bf16) per k-step.  Weights live in registers (8 fragments per wave).
void pack_conv1_1_weights(const float* w, uint16_t* out) {
    size_t o = 0;
    for (int s = 0; s < 2; ++s)
        for (int nf = 0; nf < 4; ++nf)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 15;
                const int cout = (row >> 2) * 16 + nf * 4 + (row & 3);
                for (int j = 0; j < 8; ++j) {
                    const int k = s * 32 + 8 * (l >> 4) + j;
                    const int tap = k >> 2, ch = k & 3;
                    float v = 0.f;
                    if (tap < 9 && ch < 3) v = w[((size_t)cout * 3 + ch) * 9 + tap];
                    out[o++] = f32_to_bf16_host(v);
                }
            }
}

// variant for the producer fused into conv1_2 (conv_mfma.hip): fragment nf, row -> cout (nf>>1)*32 + (row>>2)*8 + (nf&1)*4 + (row&3)
void pack_conv1_1_weights_fused(const float* w, uint16_t* out) {
    size_t o = 0;
    for (int s = 0; s < 2; ++s)
        for (int nf = 0; nf < 4; ++nf)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 15;
                const int cout = (nf >> 1) * 32 + (row >> 2) * 8 + (nf & 1) * 4 + (row & 3);
                for (int j = 0; j < 8; ++j) {
                    const int k = s * 32 + 8 * (l >> 4) + j;
                    const int tap = k >> 2, ch = k & 3;
                    float v = 0.f;
                    if (tap < 9 && ch < 3) v = w[((size_t)cout * 3 + ch) * 9 + tap];
                    out[o++] = f32_to_bf16_host(v);
                }
            }
}

__global__ void __launch_bounds__(256) conv1_1_kernel(const uint8_t* __restrict__ rgb, int Himg, int Wimg, int H32, int W32,
                                                      const uint16_t* __restrict__ wpk, const float* __restrict__ bias,
                                                      uint16_t* __restrict__ out, int tiles_x, int tiles_y) {
    constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2;
    __shared__ __attribute__((aligned(16))) u32x2 patch[PH * PW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const float m0 = 0.485f * 255.0f, m1 = 0.456f * 255.0f, m2 = 0.406f * 255.0f;
    const float s0 = 0.229f * 255.0f, s1 = 0.224f * 255.0f, s2 = 0.225f * 255.0f;
    for (int p = tid; p < PH * PW; p += 256) {
        const int py = p / PW, px = p - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        u32x2 v = {0u, 0u};
        if (iy >= 0 && iy < H32 && ix >= 0 && ix < W32) {
            float r = 0.f, g = 0.f, b = 0.f;
            if (iy < Himg && ix < Wimg) {
                const uint8_t* q = rgb + ((size_t)(n * Himg + iy) * Wimg + ix) * 3;
                r = (float)q[0]; g = (float)q[1]; b = (float)q[2];
            }
            v[0] = pack_bf16x2((r - m0) / s0, (g - m1) / s1);
            v[1] = pack_bf16x2((b - m2) / s2, 0.f);
        }
        patch[p] = v;
    }
    bf16x8 af[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) af[s][j] = *(const bf16x8*)(wpk + ((size_t)(s * 4 + j) * 64 + lane) * 8);
    const int g = lane >> 4, pl = lane & 15;
    float bs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bs[j] = bias[g * 16 + j];
    __syncthreads();
#pragma unroll 1
    for (int f = 0; f < 8; ++f) {
        const int fr = wave * 4 + (f >> 1), fc = f & 1;
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int t0 = s * 8 + 2 * g, t1 = t0 + 1;
            t0 = t0 > 8 ? 8 : t0;
            t1 = t1 > 8 ? 8 : t1;
            const u32x2 a0 = patch[(fr + t0 / 3) * PW + fc * 16 + pl + (t0 % 3)];
            const u32x2 a1 = patch[(fr + t1 / 3) * PW + fc * 16 + pl + (t1 % 3)];
            const u32x4 bb = {a0[0], a0[1], a1[0], a1[1]};
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, bb);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s][j], bfr, acc[j], 0, 0, 0);
        }
        const int oy = oy0 + fr, ox = ox0 + fc * 16 + pl;
        if (oy < H32 && ox < W32) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j * 4 + r] = fmaxf(acc[j][r] + bs[j * 4 + r], 0.f);
            uint16_t* op = out + ((size_t)(n * H32 + oy) * W32 + ox) * 64 + g * 16;
            *(u32x4*)(op) = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            *(u32x4*)(op + 8) = (u32x4){pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])};
        }
    }
}

hipError_t launch_conv1_1(const uint8_t* rgb, int N, int Himg, int Wimg, int H32, int W32, const uint16_t* wpk, const float* bias,
                          uint16_t* out, hipStream_t s) {
    const int tiles_x = (W32 + 31) / 32, tiles_y = (H32 + 15) / 16;
    hipLaunchKernelGGL(conv1_1_kernel, dim3(N * tiles_x * tiles_y), dim3(256), 0, s, rgb, Himg, Wimg, H32, W32, wpk, bias, out, tiles_x,
                       tiles_y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ max-pool
// NHWC bf16, 8 channels (16 B) per thread; optional ReLU on the input (max(relu(x)) == relu(max(x))).
__device__ __forceinline__ float bf16lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }

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
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = relu_in ? 0.f : -3.0e38f;
        for (int dy = 0; dy < kh; ++dy) {
            const int iy = oy * sh - ph + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < kw; ++dx) {
                const int ix = ox * sw - pw + dx;
                if (ix < 0 || ix >= W) continue;
                const u32x4 v = *(const u32x4*)(in + (((size_t)(n * H + iy) * W + ix) * C8 + c8) * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m[2 * j] = fmaxf(m[2 * j], bf16lo(v[j]));
                    m[2 * j + 1] = fmaxf(m[2 * j + 1], bf16hi(v[j]));
                }
            }
        }
        // values are exact bf16, so re-packing by truncation is lossless
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__float_as_uint(m[2 * j]) >> 16) | (__float_as_uint(m[2 * j + 1]) & 0xffff0000u);
        *(u32x4*)(out + i * 8) = o;
    }
}

hipError_t launch_maxpool(const uint16_t* in, uint16_t* out, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                          int relu_in, hipStream_t s) {
    if (C & 7) return hipErrorInvalidValue;
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
