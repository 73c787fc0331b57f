// Element-wise helpers of the EXACT detector (bbocr_config::precision = BBOCR_PREC_EXACT): every activation of the CRAFT pass is a PAIR of
// fp16 tensors [hi C | lo C] per pixel with value = hi + lo / 2048 (22 significand bits; kernels.h REC_SPLIT), every conv runs as a split-fp16
// plan (weights.cpp::upload_split_plan: three MFMA product terms per layer in one launch, fp32 accumulation).  What is not a convolution
// runs here, on the pair values rebuilt in fp32, in the operation order of the fp32 reference (easyocr/craft.py::CRAFT.forward,
// easyocr/imgproc.py::normalizeMeanVariance): conv1_1 from the uint8 page, ReLU, pool5, F.interpolate + torch.cat, the classifier tail.
// None of it is on the fast modes' path; these kernels are plain grid-stride loops over 8-channel (16-byte) groups.
#include "common.h"
#include "kernels.h"

namespace {
constexpr float kInvLo = 1.0f / SPLIT_LO_SCALE;

// 8 channels of a pair tensor -> fp32
__device__ __forceinline__ void pair_load8(const uint16_t* hi, const uint16_t* lo, float (&v)[8]) {
    const u32x4 h = *(const u32x4*)hi, l = *(const u32x4*)lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2_t a = El<1>::unpack2(h[j]), b = El<1>::unpack2(l[j]);
        v[2 * j] = a[0] + b[0] * kInvLo;
        v[2 * j + 1] = a[1] + b[1] * kInvLo;
    }
}
// fp32 -> 8 channels of a pair tensor: hi = fp16(v), lo = fp16((v - hi) * 2048)
__device__ __forceinline__ void pair_store8(uint16_t* hi, uint16_t* lo, const float (&v)[8]) {
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int q = El<1>::pack2(v[2 * j], v[2 * j + 1]);
        const f32x2_t r = El<1>::unpack2(q);
        h[j] = q;
        l[j] = El<1>::pack2((v[2 * j] - r[0]) * SPLIT_LO_SCALE, (v[2 * j + 1] - r[1]) * SPLIT_LO_SCALE);
    }
    *(u32x4*)hi = h;
    *(u32x4*)lo = l;
}
inline int grid_for(size_t total, int cap = 16384) {
    const size_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g < (size_t)cap ? g : (size_t)cap));
}
}  // namespace

// ------------------------------------------------------------------------------------------------ conv1_1 (3 -> 64) + BN + ReLU, fp32
// uint8 RGB pages [N, Hi, Wi, 3] on the zero canvas H x W -> pair [N, H, W, 64 | 64].  normalizeMeanVariance in fp32 exactly as numpy does
// it ((x - mean * 255) / (std * 255)); pixels of the canvas beyond the page are raw zeros (normalised like any pixel), pixels beyond the
// canvas are the convolution's zero padding.  w: folded fp32 [64][3][3][3] (cout, cin, ky, kx), b: [64].
// One thread = one pixel x all 64 couts: its 27 normalised inputs live in registers, the weights are read from LDS as [27][64] -- every
// lane of a wave reads the same address (a broadcast: no bank conflicts), 4 weights per ds_read_b128 -- and the 1,728 FMAs per pixel run at
// the fp32 issue rate.  (The first version gave each thread 8 couts and read the weights from global memory inside the tap loop: 38 ms per
// 11-page pass, a third of the exact mode's step; this one ~1 ms.)  The pair leaves through an LDS stage as contiguous lines (below).
__global__ void __launch_bounds__(256) pair_conv1_1_kernel(const uint8_t* __restrict__ rgb, int N, int Hi, int Wi, int H, int W, const float* __restrict__ w,
                                                           const float* __restrict__ b, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 64 + 64];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) {
        const int k = i >> 6, co = i & 63;                    // k = (ky * 3 + kx) * 3 + ch
        const int tap = k / 3, ch = k - tap * 3;
        wl[i] = w[((size_t)co * 3 + ch) * 9 + tap];
    }
    if (threadIdx.x < 64) wl[27 * 64 + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    const float mean[3] = {0.485f * 255.0f, 0.456f * 255.0f, 0.406f * 255.0f};
    const float sd[3] = {0.229f * 255.0f, 0.224f * 255.0f, 0.225f * 255.0f};
    // a pixel's 256 output bytes ([hi 64 | lo 64]) are staged in LDS (row pitch 272 B: the lanes' 16-byte pieces fall on different
    // banks) and leave as one contiguous 64-KB block per 256 pixels, 16 B per lane: written straight from the registers every store
    // instruction touched 64 different 256-byte segments, and the kernel ran at a third of the rate its stores allow
    __shared__ __attribute__((aligned(16))) unsigned char stage[256 * 272];
    const size_t total = (size_t)N * H * W;
    for (size_t base = (size_t)blockIdx.x * 256; base < total; base += (size_t)gridDim.x * 256) {
        const size_t i = base + threadIdx.x;
        if (i < total) {
            const int x = (int)(i % W);
            const size_t r = i / W;
            const int y = (int)(r % H);
            const int n = (int)(r / H);
            float v[27];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = y - 1 + ky, ix = x - 1 + kx;
                    const bool in_canvas = iy >= 0 && iy < H && ix >= 0 && ix < W;       // else: zero padding of the conv
                    const bool on_page = in_canvas && iy < Hi && ix < Wi;
                    const uint8_t* q = rgb + ((size_t)(n * Hi + (on_page ? iy : 0)) * Wi + (on_page ? ix : 0)) * 3;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) v[(ky * 3 + kx) * 3 + ch] = in_canvas ? ((on_page ? (float)q[ch] : 0.f) - mean[ch]) / sd[ch] : 0.f;
                }
            uint16_t* op = (uint16_t*)(stage + threadIdx.x * 272);
#pragma unroll 1
            for (int c8 = 0; c8 < 8; ++c8) {
                float acc[8];
                {
                    const f32x4 b0 = *(const f32x4*)(wl + 27 * 64 + c8 * 8), b1 = *(const f32x4*)(wl + 27 * 64 + c8 * 8 + 4);
                    acc[0] = b0[0]; acc[1] = b0[1]; acc[2] = b0[2]; acc[3] = b0[3]; acc[4] = b1[0]; acc[5] = b1[1]; acc[6] = b1[2]; acc[7] = b1[3];
                }
#pragma unroll
                for (int k = 0; k < 27; ++k) {
                    const f32x4 w0 = *(const f32x4*)(wl + k * 64 + c8 * 8), w1 = *(const f32x4*)(wl + k * 64 + c8 * 8 + 4);
                    acc[0] = fmaf(w0[0], v[k], acc[0]); acc[1] = fmaf(w0[1], v[k], acc[1]); acc[2] = fmaf(w0[2], v[k], acc[2]); acc[3] = fmaf(w0[3], v[k], acc[3]);
                    acc[4] = fmaf(w1[0], v[k], acc[4]); acc[5] = fmaf(w1[1], v[k], acc[5]); acc[6] = fmaf(w1[2], v[k], acc[6]); acc[7] = fmaf(w1[3], v[k], acc[7]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
                pair_store8(op + c8 * 8, op + 64 + c8 * 8, acc);
            }
        }
        __syncthreads();
        const size_t npx = total - base < 256 ? total - base : 256;
        unsigned char* gout = (unsigned char*)out + base * 256;
        for (unsigned int u = threadIdx.x; u < (unsigned int)npx * 16; u += 256)
            *(u32x4*)(gout + (size_t)u * 16) = *(const u32x4*)(stage + (u >> 4) * 272 + (u & 15) * 16);
        __syncthreads();
    }
}
hipError_t launch_pair_conv1_1(const uint8_t* rgb, int N, int Hi, int Wi, int H, int W, const float* w, const float* b, uint16_t* out, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    hipLaunchKernelGGL(pair_conv1_1_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, s, rgb, N, Hi, Wi, H, W, w, b, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ ReLU on a pair tensor [npix, C | C]
// (the fast modes apply ReLU-on-load as an integer max on the 16-bit operands; on a pair that would treat hi and lo separately)
__global__ void __launch_bounds__(256) pair_relu_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, size_t npix, int C8) {
    const size_t total = npix * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / C8, c8 = i - p * C8;
        const size_t o = p * C8 * 16 + c8 * 8;
        float v[8];
        pair_load8(in + o, in + o + (size_t)C8 * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        pair_store8(out + o, out + o + (size_t)C8 * 8, v);
    }
}
hipError_t launch_pair_relu(const uint16_t* in, uint16_t* out, size_t npix, int C, hipStream_t s) {
    if (C & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pair_relu_kernel, dim3(grid_for(npix * (C / 8))), dim3(256), 0, s, in, out, npix, C / 8);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ MaxPool2d(3, 1, 1) on a pair tensor
__global__ void __launch_bounds__(256) pair_maxpool3x3s1_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int N, int H, int W, int C8) {
    const size_t total = (size_t)N * H * W * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        size_t r = i / C8;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int n = (int)(r / H);
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -__builtin_inff();
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int iy = y + dy, ix = x + dx;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const size_t o = ((size_t)(n * H + iy) * W + ix) * C8 * 16 + (size_t)c8 * 8;
                float v[8];
                pair_load8(in + o, in + o + (size_t)C8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
            }
        const size_t o = (i / C8) * C8 * 16 + (size_t)c8 * 8;
        pair_store8(out + o, out + o + (size_t)C8 * 8, m);
    }
}
hipError_t launch_pair_maxpool3x3s1(const uint16_t* in, uint16_t* out, int N, int H, int W, int C, hipStream_t s) {
    if (C & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pair_maxpool3x3s1_kernel, dim3(grid_for((size_t)N * H * W * (C / 8))), dim3(256), 0, s, in, out, N, H, W, C / 8);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ torch.cat([F.interpolate(y), skip], dim = 1)
// y pair [N, yh, yw, Cy | Cy] -> bilinear, align_corners = False, to the skip's size H x W (up = 1: exactly 2x; up = 0: same size, plain
// copy), then the channel concat with the skip pair [N, H, W, Cs | Cs] -> out pair [N, H, W, (Cy + Cs) | (Cy + Cs)].  The blend runs in
// fp32 like torch's upsample_bilinear2d: source index (dst + 0.5) * 0.5 - 0.5 clamped at 0, neighbour clamped at the far edge.
__global__ void __launch_bounds__(256) pair_upcat_kernel(const uint16_t* __restrict__ y, int yh, int yw, int Cy8, const uint16_t* __restrict__ skip, int Cs8,
                                                         uint16_t* __restrict__ out, int N, int H, int W, int up) {
    const int Co8 = Cy8 + Cs8;
    const size_t total = (size_t)N * H * W * Co8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % Co8);
        size_t r = i / Co8;
        const int x = (int)(r % W);
        r /= W;
        const int yy = (int)(r % H);
        const int n = (int)(r / H);
        float v[8];
        if (c8 >= Cy8) {
            const size_t o = ((size_t)(n * H + yy) * W + x) * Cs8 * 16 + (size_t)(c8 - Cy8) * 8;
            pair_load8(skip + o, skip + o + (size_t)Cs8 * 8, v);
        } else if (!up) {
            const size_t o = ((size_t)(n * yh + yy) * yw + x) * Cy8 * 16 + (size_t)c8 * 8;
            pair_load8(y + o, y + o + (size_t)Cy8 * 8, v);
        } else {
            float sy = ((float)yy + 0.5f) * 0.5f - 0.5f, sx = ((float)x + 0.5f) * 0.5f - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < yh - 1 ? 1 : 0), x1 = x0 + (x0 < yw - 1 ? 1 : 0);
            const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
            float a[8], b[8], c[8], d[8];
            auto at = [&](int py, int px, float (&t)[8]) {
                const size_t o = ((size_t)(n * yh + py) * yw + px) * Cy8 * 16 + (size_t)c8 * 8;
                pair_load8(y + o, y + o + (size_t)Cy8 * 8, t);
            };
            at(y0, x0, a); at(y0, x1, b); at(y1, x0, c); at(y1, x1, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = hy * (hx * a[j] + lx * b[j]) + ly * (hx * c[j] + lx * d[j]);      // torch's grouping
        }
        const size_t o = (i / Co8) * Co8 * 16 + (size_t)c8 * 8;
        pair_store8(out + o, out + o + (size_t)Co8 * 8, v);
    }
}
hipError_t launch_pair_upcat(const uint16_t* y, int yh, int yw, int Cy, const uint16_t* skip, int Cs, uint16_t* out, int N, int H, int W, hipStream_t s) {
    if ((Cy & 7) || (Cs & 7)) return hipErrorInvalidValue;
    const int up = (yh == H && yw == W) ? 0 : 1;
    if (up && (H != 2 * yh || W != 2 * yw)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pair_upcat_kernel, dim3(grid_for((size_t)N * H * W * ((Cy + Cs) / 8), 65536)), dim3(256), 0, s, y, yh, yw, Cy / 8, skip, Cs / 8, out, N,
                       H, W, up);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ classifier tail in fp32
// conv_cls.6 (1x1 16 -> 16 + ReLU) and conv_cls.8 (1x1 16 -> 2) on the pair [npix, 16 | 16] conv_cls.4 stored -> heat fp32 [npix, 2].
// tail = {b1[16], w2[2][16], b2[2]} (ctx cls_tail), w1 = conv_cls.6.weight [16][16] fp32.
__global__ void __launch_bounds__(256) pair_cls_tail_kernel(const uint16_t* __restrict__ in, const float* __restrict__ w1, const float* __restrict__ tail,
                                                            float* __restrict__ heat, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        float v[16];
        {
            float a[8], b[8];
            pair_load8(in + i * 32, in + i * 32 + 16, a);
            pair_load8(in + i * 32 + 8, in + i * 32 + 24, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = a[j]; v[8 + j] = b[j]; }
        }
        float p0 = tail[48], p1 = tail[49];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            float h = tail[o];
#pragma unroll
            for (int k = 0; k < 16; ++k) h = fmaf(w1[o * 16 + k], v[k], h);
            h = fmaxf(h, 0.f);
            p0 = fmaf(tail[16 + o], h, p0);
            p1 = fmaf(tail[32 + o], h, p1);
        }
        *(float2*)(heat + i * 2) = make_float2(p0, p1);
    }
}
hipError_t launch_pair_cls_tail(const uint16_t* in, const float* w1, const float* tail, float* heat, size_t npix, hipStream_t s) {
    hipLaunchKernelGGL(pair_cls_tail_kernel, dim3(grid_for(npix)), dim3(256), 0, s, in, w1, tail, heat, npix);
    return hipGetLastError();
}
