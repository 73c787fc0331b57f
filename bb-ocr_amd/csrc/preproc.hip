// Device kernels for the reference's OCR pre-processing chain:
// pipeline_demo/ocr_testing/preprocessing/image_preprocessor.py::preprocess_for_book_cover (:147-160) =
//   cv2 BGR2GRAY -> cv2.resize x1.5 INTER_CUBIC -> cv2.GaussianBlur 3x3 sigma 3 -> PIL Contrast 1.9 -> PIL Brightness 1.2 ->
//   cv2 CLAHE (clip 2.5, 8x8 tiles) -> PIL UnsharpMask(radius 1, 30 %, threshold 3).
// All stages are 8-bit integer / byte work on one channel, HBM-bound (one read + one write of the 1.5x-upscaled plane each);
// the integer arithmetic is the CPU restatement's (oracle/preprocess.py) operation for operation, float steps are evaluated
// in the same order without contraction (-ffp-contract=off).  Coefficient tables, the contrast/brightness LUT and the CLAHE
// tile LUTs are tiny and are built on the host (api.cpp) from device-side sums / histograms.
#include "common.h"
#include "kernels.h"

// ---- cv2.resize INTER_CUBIC, 8u, as the IPP-backed opencv-python wheels compute it (ippiResizeCubic_8u, B = 0, C = 0.75): the exact
// bicubic value (A = -0.75, replicated borders) rounded half to even -- pinned by the reference's stored pre-processing outputs
// (oracle/preprocess.py::resize_cubic_u8, tests/golden/legacy_preprocess).  Per axis the host supplies the first tap, the four
// weights as doubles AND as exact integers over K = 4 (2 dst)^3.  A pixel is evaluated in float64 (error < 1e-12); only when that
// lands within 1e-9 of a rounding boundary is it decided exactly: 2 * sum_j ny_j (sum_i nx_i p_ij) <> (2 n + 1) KX KY in 128-bit
// integers (flat regions make exact x.5 ties common: ~0.1 % of a scanned cover's pixels).
__global__ void __launch_bounds__(256) pp_resize_cubic_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst, int dh, int dw,
                                                               const int* __restrict__ x0, const double* __restrict__ wx,
                                                               const long long* __restrict__ nx, const int* __restrict__ y0,
                                                               const double* __restrict__ wy, const long long* __restrict__ ny,
                                                               unsigned long long KX, unsigned long long KY) {
    const size_t total = (size_t)dh * dw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int dy = (int)(i / dw), dx = (int)(i - (size_t)dy * dw);
        const int sx = x0[dx], sy = y0[dy];
        int xs[4], p[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = sx + k;
            xs[k] = x < 0 ? 0 : (x >= W ? W - 1 : x);
        }
        double val = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int y = sy + j;
            y = y < 0 ? 0 : (y >= H ? H - 1 : y);
            const uint8_t* row = src + (size_t)y * W;
            double hor = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p[j][k] = row[xs[k]];
                hor += (double)p[j][k] * wx[dx * 4 + k];
            }
            val += hor * wy[dy * 4 + j];
        }
        const double fl = floor(val);
        double r = rint(val);                                   // half to even (only used away from the boundary)
        if (fabs(val - fl - 0.5) < 1e-9) {
            __int128 ex = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                long long hor = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) hor += nx[dx * 4 + k] * (long long)p[j][k];
                ex += (__int128)ny[dy * 4 + j] * (__int128)hor;
            }
            const long long n = (long long)fl;
            const __int128 c = 2 * ex - (__int128)(2 * n + 1) * ((__int128)KX * (__int128)KY);
            r = (double)(c > 0 ? n + 1 : (c < 0 ? n : n + (n & 1)));
        }
        dst[i] = (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
    }
}

// ---- cv2.GaussianBlur 3x3 (fixed-point 8.8 taps k0,k1,k2), BORDER_REFLECT_101; also accumulates the sum of the OUTPUT pixels
// (ImageEnhance.Contrast needs the mean of the blurred image)
__global__ void __launch_bounds__(256) pp_gauss3_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst, int k0, int k1, int k2,
                                                         unsigned long long* __restrict__ sum) {
    const size_t total = (size_t)H * W;
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const int xm = x > 0 ? x - 1 : (W > 1 ? 1 : 0), xp = x + 1 < W ? x + 1 : (W > 1 ? W - 2 : 0);
        const int ym = y > 0 ? y - 1 : (H > 1 ? 1 : 0), yp = y + 1 < H ? y + 1 : (H > 1 ? H - 2 : 0);
        const uint8_t *r0 = src + (size_t)ym * W, *r1 = src + (size_t)y * W, *r2 = src + (size_t)yp * W;
        const int h0 = k0 * r0[xm] + k1 * r0[x] + k2 * r0[xp];
        const int h1 = k0 * r1[xm] + k1 * r1[x] + k2 * r1[xp];
        const int h2 = k0 * r2[xm] + k1 * r2[x] + k2 * r2[xp];
        int v = (k0 * h0 + k1 * h1 + k2 * h2 + (1 << 15)) >> 16;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        dst[i] = (uint8_t)v;
        local += (unsigned)v;
    }
    // one atomic per workgroup (a per-wave atomic on the single sum address serialised 860k updates for a 55-MP plane: 3 ms)
    __shared__ unsigned long long part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(sum, t);
    }
}

// ---- the same, four pixels per thread (W % 4 == 0): three dword row loads + the two columns next to them
__global__ void __launch_bounds__(256) pp_gauss3_x4_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst, int k0, int k1, int k2,
                                                            unsigned long long* __restrict__ sum) {
    const int W4 = W >> 2;
    const size_t total = (size_t)H * W4;
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W4), x = (int)(i - (size_t)y * W4) << 2;
        const int xm = x > 0 ? x - 1 : (W > 1 ? 1 : 0), xp = x + 4 < W ? x + 4 : (W > 1 ? W - 2 : 0);
        const int ym = y > 0 ? y - 1 : (H > 1 ? 1 : 0), yp = y + 1 < H ? y + 1 : (H > 1 ? H - 2 : 0);
        int hrow[3][4];
        const int ys[3] = {ym, y, yp};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint8_t* row = src + (size_t)ys[r] * W;
            const unsigned int c = *(const unsigned int*)(row + x);
            const int p[6] = {row[xm], (int)(c & 255u), (int)((c >> 8) & 255u), (int)((c >> 16) & 255u), (int)(c >> 24), row[xp]};
#pragma unroll
            for (int k = 0; k < 4; ++k) hrow[r][k] = k0 * p[k] + k1 * p[k + 1] + k2 * p[k + 2];
        }
        unsigned int o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int v = (k0 * hrow[0][k] + k1 * hrow[1][k] + k2 * hrow[2][k] + (1 << 15)) >> 16;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            o |= (unsigned)v << (8 * k);
            local += (unsigned)v;
        }
        *(unsigned int*)(dst + (size_t)y * W + x) = o;
    }
    __shared__ unsigned long long part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(sum, t);
    }
}

// ---- CLAHE pass 1: per-tile histograms of lut[src] over the reflect-101 extended image (tiles tw x th, grid tx x ty)
__global__ void __launch_bounds__(256) pp_clahe_hist_kernel(const uint8_t* __restrict__ src, int H, int W, const uint8_t* __restrict__ lut, int tw, int th,
                                                             int tx, int ty, unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    const int tile = blockIdx.x;                                // one tile per blockIdx.x, blockIdx.y splits its rows
    const int tj = tile / tx, ti = tile - tj * tx;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int rows_per = (th + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(th, r0 + rows_per);
    for (int r = r0; r < r1; ++r) {
        int y = tj * th + r;
        if (y >= H) y = 2 * (H - 1) - y;                        // BORDER_REFLECT_101 (only the bottom / right are padded)
        const uint8_t* row = src + (size_t)y * W;
        for (int c = threadIdx.x; c < tw; c += 256) {
            int x = ti * tw + c;
            if (x >= W) x = 2 * (W - 1) - x;
            atomicAdd(&h[lut[row[x]]], 1u);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[tile * 256 + threadIdx.x], h[threadIdx.x]);
}

// ---- CLAHE pass 2: dst = cvRound((L11*xa1 + L12*xa)*ya1 + (L21*xa1 + L22*xa)*ya), Lij = tile LUTs of v = lut[src]
__global__ void __launch_bounds__(256) pp_clahe_apply_kernel(const uint8_t* __restrict__ src, int H, int W, const uint8_t* __restrict__ lut,
                                                              const uint8_t* __restrict__ tile_luts, int tw, int th, int tx, int ty,
                                                              uint8_t* __restrict__ dst, int vec) {
    const float inv_th = 1.0f / (float)th, inv_tw = 1.0f / (float)tw;
    auto pixel = [&](int y, int x, int raw) {
        const float tyf = (float)y * inv_th - 0.5f, txf = (float)x * inv_tw - 0.5f;
        int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
        const float ya = tyf - (float)ty1, xa = txf - (float)tx1;
        const float ya1 = 1.0f - ya, xa1 = 1.0f - xa;
        int ty2 = ty1 + 1, tx2 = tx1 + 1;
        ty1 = ty1 < 0 ? 0 : ty1; tx1 = tx1 < 0 ? 0 : tx1;
        ty2 = ty2 > ty - 1 ? ty - 1 : ty2; tx2 = tx2 > tx - 1 ? tx - 1 : tx2;
        const int v = lut[raw];
        const float l11 = (float)tile_luts[(ty1 * tx + tx1) * 256 + v], l12 = (float)tile_luts[(ty1 * tx + tx2) * 256 + v];
        const float l21 = (float)tile_luts[(ty2 * tx + tx1) * 256 + v], l22 = (float)tile_luts[(ty2 * tx + tx2) * 256 + v];
        const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
        const int r = __float2int_rn(res);                       // cvRound: round half to even
        return (unsigned)(r < 0 ? 0 : (r > 255 ? 255 : r));
    };
    if (vec) {                                                   // W % 4 == 0: dword load / store, four pixels per thread
        const int W4 = W >> 2;
        const size_t total = (size_t)H * W4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int y = (int)(i / W4), x = (int)(i - (size_t)y * W4) << 2;
            const unsigned int c = *(const unsigned int*)(src + (size_t)y * W + x);
            unsigned int o = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) o |= pixel(y, x + k, (c >> (8 * k)) & 255u) << (8 * k);
            *(unsigned int*)(dst + (size_t)y * W + x) = o;
        }
        return;
    }
    const size_t total = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        dst[i] = (uint8_t)pixel(y, x, src[i]);
    }
}

// ---- PIL box blur, one pass along rows (stride_x = 1) or columns (stride_x = W): radius r, weights ww / fw, edge replication
__global__ void __launch_bounds__(256) pp_box_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int vertical, int r,
                                                           unsigned int ww, unsigned int fw) {
    const size_t total = (size_t)H * W;
    const int n = vertical ? H : W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const int p = vertical ? y : x;
        const uint8_t* line = vertical ? src + x : src + (size_t)y * W;
        const size_t st = vertical ? (size_t)W : 1;
        unsigned int acc = 0;
        for (int d = -r; d <= r; ++d) {
            int q = p + d;
            q = q < 0 ? 0 : (q >= n ? n - 1 : q);
            acc += line[(size_t)q * st];
        }
        int qa = p - r - 1, qb = p + r + 1;
        qa = qa < 0 ? 0 : qa;
        qb = qb >= n ? n - 1 : qb;
        const unsigned int far = (unsigned)line[(size_t)qa * st] + (unsigned)line[(size_t)qb * st];
        const unsigned int bulk = acc * ww + far * fw;           // UINT32 arithmetic like libImaging/BoxBlur.c
        dst[i] = (uint8_t)((bulk + (1u << 23)) >> 24);
    }
}

// ---- the same for box radius 0 (PIL GaussianBlur radius 1: box radius 0.25), four pixels per thread, W % 4 == 0:
// out = (v*ww + (left + right)*fw + 2^23) >> 24 with edge replication, dword loads / stores
__global__ void __launch_bounds__(256) pp_box0_x4_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int vertical,
                                                          unsigned int ww, unsigned int fw) {
    const int W4 = W >> 2;
    const size_t total = (size_t)H * W4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W4), x = (int)(i - (size_t)y * W4) << 2;
        const uint8_t* row = src + (size_t)y * W;
        const unsigned int c = *(const unsigned int*)(row + x);
        unsigned int v[4] = {c & 255u, (c >> 8) & 255u, (c >> 16) & 255u, c >> 24}, a[4], b[4];
        if (vertical) {
            const unsigned int up = *(const unsigned int*)(src + (size_t)(y > 0 ? y - 1 : 0) * W + x);
            const unsigned int dn = *(const unsigned int*)(src + (size_t)(y + 1 < H ? y + 1 : H - 1) * W + x);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = (up >> (8 * k)) & 255u; b[k] = (dn >> (8 * k)) & 255u; }
        } else {
            a[0] = row[x > 0 ? x - 1 : 0]; a[1] = v[0]; a[2] = v[1]; a[3] = v[2];
            b[0] = v[1]; b[1] = v[2]; b[2] = v[3]; b[3] = row[x + 4 < W ? x + 4 : W - 1];
        }
        unsigned int o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o |= (((v[k] * ww + (a[k] + b[k]) * fw) + (1u << 23)) >> 24) << (8 * k);
        *(unsigned int*)(dst + (size_t)y * W + x) = o;
    }
}

// ---- PIL UnsharpMask combine: diff = in - blur; |diff| > threshold ? clip8(in + diff * percent / 100) : in
__global__ void __launch_bounds__(256) pp_unsharp_kernel(const uint8_t* __restrict__ in, const uint8_t* __restrict__ blur, uint8_t* __restrict__ dst,
                                                          size_t total, int percent, int threshold) {
    const size_t t4 = total >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < t4; i += (size_t)gridDim.x * 256) {
        const unsigned int av = ((const unsigned int*)in)[i], bv = ((const unsigned int*)blur)[i];
        unsigned int ov = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = (av >> (8 * k)) & 255, diff = a - (int)((bv >> (8 * k)) & 255);
            int o = a;
            if (abs(diff) > threshold) {
                o = a + diff * percent / 100;                    // C integer division: truncates towards zero
                o = o < 0 ? 0 : (o > 255 ? 255 : o);
            }
            ov |= (unsigned)o << (8 * k);
        }
        ((unsigned int*)dst)[i] = ov;
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {          // tail pixels
        const size_t i = (t4 << 2) + threadIdx.x;
        const int a = in[i], diff = a - (int)blur[i];
        int o = a;
        if (abs(diff) > threshold) {
            o = a + diff * percent / 100;
            o = o < 0 ? 0 : (o > 255 ? 255 : o);
        }
        dst[i] = (uint8_t)o;
    }
}

// ---- pointwise lookup (the PIL enhancers on their own; inside the chain they ride in front of CLAHE)
__global__ void __launch_bounds__(256) pp_lut_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint8_t* __restrict__ lut, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) dst[i] = lut[src[i]];
}

static inline int pp_grid(size_t total) {
    const size_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

hipError_t launch_pp_resize_cubic(const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw, const int* x0, const double* wx, const long long* nx,
                                  const int* y0, const double* wy, const long long* ny, unsigned long long KX, unsigned long long KY, hipStream_t s) {
    hipLaunchKernelGGL(pp_resize_cubic_kernel, dim3(pp_grid((size_t)dh * dw)), dim3(256), 0, s, src, H, W, dst, dh, dw, x0, wx, nx, y0, wy, ny, KX, KY);
    return hipGetLastError();
}
hipError_t launch_pp_gauss3(const uint8_t* src, int H, int W, uint8_t* dst, int k0, int k1, int k2, unsigned long long* sum, hipStream_t s) {
    if ((W & 3) == 0 && W >= 8 && (((size_t)src | (size_t)dst) & 3) == 0) {
        const int g4 = pp_grid((size_t)H * (W >> 2)) > 4096 ? 4096 : pp_grid((size_t)H * (W >> 2));
        hipLaunchKernelGGL(pp_gauss3_x4_kernel, dim3(g4), dim3(256), 0, s, src, H, W, dst, k0, k1, k2, sum);
        return hipGetLastError();
    }
    const int grid = pp_grid((size_t)H * W) > 4096 ? 4096 : pp_grid((size_t)H * W);      // grid-stride: at most 4096 atomics on `sum`
    hipLaunchKernelGGL(pp_gauss3_kernel, dim3(grid), dim3(256), 0, s, src, H, W, dst, k0, k1, k2, sum);
    return hipGetLastError();
}
hipError_t launch_pp_clahe_hist(const uint8_t* src, int H, int W, const uint8_t* lut, int tw, int th, int tx, int ty, unsigned int* hist,
                                hipStream_t s) {
    hipLaunchKernelGGL(pp_clahe_hist_kernel, dim3(tx * ty, 16), dim3(256), 0, s, src, H, W, lut, tw, th, tx, ty, hist);
    return hipGetLastError();
}
hipError_t launch_pp_clahe_apply(const uint8_t* src, int H, int W, const uint8_t* lut, const uint8_t* tile_luts, int tw, int th, int tx, int ty,
                                 uint8_t* dst, hipStream_t s) {
    const int vec = (W & 3) == 0 && (((size_t)src | (size_t)dst) & 3) == 0;
    hipLaunchKernelGGL(pp_clahe_apply_kernel, dim3(pp_grid(vec ? (size_t)H * (W >> 2) : (size_t)H * W)), dim3(256), 0, s, src, H, W, lut, tile_luts, tw,
                       th, tx, ty, dst, vec);
    return hipGetLastError();
}
hipError_t launch_pp_box_pass(const uint8_t* src, uint8_t* dst, int H, int W, int vertical, int r, unsigned int ww, unsigned int fw, hipStream_t s) {
    if (r == 0 && (W & 3) == 0 && (((size_t)src | (size_t)dst) & 3) == 0) {
        hipLaunchKernelGGL(pp_box0_x4_kernel, dim3(pp_grid((size_t)H * (W >> 2))), dim3(256), 0, s, src, dst, H, W, vertical, ww, fw);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pp_box_pass_kernel, dim3(pp_grid((size_t)H * W)), dim3(256), 0, s, src, dst, H, W, vertical, r, ww, fw);
    return hipGetLastError();
}
hipError_t launch_pp_unsharp(const uint8_t* in, const uint8_t* blur, uint8_t* dst, size_t total, int percent, int threshold, hipStream_t s) {
    hipLaunchKernelGGL(pp_unsharp_kernel, dim3(pp_grid((total >> 2) + 1)), dim3(256), 0, s, in, blur, dst, total, percent, threshold);
    return hipGetLastError();
}
hipError_t launch_pp_lut(const uint8_t* src, uint8_t* dst, const uint8_t* lut, size_t total, hipStream_t s) {
    hipLaunchKernelGGL(pp_lut_kernel, dim3(pp_grid(total)), dim3(256), 0, s, src, dst, lut, total);
    return hipGetLastError();
}
