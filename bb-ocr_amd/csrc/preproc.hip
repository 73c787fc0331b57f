// Device kernels for the reference's OCR pre-processing chain:
// pipeline_demo/ocr_testing/preprocessing/image_preprocessor.py::preprocess_for_book_cover (:147-160) =
//   cv2 BGR2GRAY -> cv2.resize x1.5 INTER_CUBIC -> cv2.GaussianBlur 3x3 sigma 3 -> PIL Contrast 1.9 -> PIL Brightness 1.2 ->
//   cv2 CLAHE (clip 2.5, 8x8 tiles) -> PIL UnsharpMask(radius 1, 30 %, threshold 3).
// All stages are 8-bit integer / byte work on one channel, HBM-bound (one read + one write of the 1.5x-upscaled plane each);
// the integer arithmetic is the CPU restatement's (oracle/preprocess.py) operation for operation, float steps are evaluated
// in the same order without contraction (-ffp-contract=off).  The cubic coefficient tables are built on the host once per page
// geometry (preprocess.cpp) and stay on the device; the contrast / brightness LUT and the CLAHE tile LUTs are built by one-block device
// kernels from the device-resident sum / histograms, so the chain runs without a host round trip.
#include "common.h"
#include "kernels.h"

// row of flat index i in a plane of row length W: 32-bit division whenever the index allows (a 64-bit one costs ~100 instructions)
__device__ __forceinline__ int pp_row(size_t i, int W) { return (i >> 32) ? (int)(i / (size_t)W) : (int)((unsigned)i / (unsigned)W); }

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
        const int dy = pp_row(i, dw), dx = (int)(i - (size_t)dy * dw);
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

// The same resize, one 64 x 16 output tile per workgroup: the source window of the tile (replicated borders applied while loading) is
// staged in LDS once, the horizontal sums hor[source row][dx] are formed once per source row -- every source row feeds ~4 dh/H output
// rows, so this is (window rows * 4 + 16 * 4) / (16 * 20) of the per-pixel kernel's float64 work at 1.5x: 0.39 -- and the results leave
// as whole dwords.  Float64 estimate (error < 1e-12) and the exact 128-bit decision within 1e-9 of a tie, like pp_resize_cubic_kernel:
// the byte is fully determined by the mathematics, not by the order or fusing of the float operations.  Launched only when the window fits PP_RS_SW x PP_RS_SH (any scale >= ~0.55).
#define PP_RS_TW 64
#define PP_RS_SW 128
#define PP_RS_SH 36
// TH = output rows per tile: 32 where the window still fits PP_RS_SH rows (scales >= ~1.04: twice the work per workgroup behind the same
// chain of table loads -> window loads -> two barriers, which is what a tile costs), else 16
// BGR: `src` is the interleaved 3-channel page and the window is filled with its cv2 BGR2GRAY values (craft_misc.hip::gray_kernel's formula):
// the chain's first stage rides in the window load, the gray plane of the source size is never written.
template <int TH, bool BGR>
__global__ void __launch_bounds__(256) pp_resize_cubic_tiled_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst, int dh,
                                                                     int dw, const int* __restrict__ x0, const double* __restrict__ wx,
                                                                     const long long* __restrict__ nx, const int* __restrict__ y0,
                                                                     const double* __restrict__ wy, const long long* __restrict__ ny,
                                                                     unsigned long long KX, unsigned long long KY) {
    __shared__ uint8_t win[PP_RS_SH][PP_RS_SW];
    __shared__ double hor[PP_RS_SH][PP_RS_TW];
    constexpr int NIT = TH / 4;
    __shared__ uint8_t res[TH][PP_RS_TW];
    const int ox = blockIdx.x * PP_RS_TW, oy = blockIdx.y * TH;
    const int nxo = min(PP_RS_TW, dw - ox), nyo = min(TH, dh - oy);
    // a thread keeps one output column (lx) and the output rows wv, wv+4, wv+8, ... of the tile through both phases: its table entries
    // are fetched up front, next to the window's corner taps, so the window load is the only dependent global access of the block
    const int lx = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dx = min(ox + lx, dw - 1);
    const int myx = x0[dx];
    const double w0 = wx[dx * 4], w1 = wx[dx * 4 + 1], w2 = wx[dx * 4 + 2], w3 = wx[dx * 4 + 3];
    int myy[NIT];
    double v[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int dy = min(oy + wv + 4 * it, dh - 1);
        myy[it] = y0[dy];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[it][j] = wy[dy * 4 + j];
    }
    const int sx0 = x0[ox], sy0 = y0[oy];                          // first taps are non-decreasing along an axis
    const int sw = x0[ox + nxo - 1] + 4 - sx0, sh = y0[oy + nyo - 1] + 4 - sy0;
    for (int ry = wv; ry < sh; ry += 4) {                          // window: a wave per row, lanes along it (no index division)
        int y = sy0 + ry;
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);
        const uint8_t* row = src + (size_t)y * W * (BGR ? 3 : 1);
        for (int rx = lx; rx < sw; rx += 64) {
            int x = sx0 + rx;
            x = x < 0 ? 0 : (x >= W ? W - 1 : x);
            if constexpr (BGR) {
                const int c0 = row[x * 3], c1 = row[x * 3 + 1], c2 = row[x * 3 + 2];
                win[ry][rx] = (uint8_t)((c2 * 9798 + c1 * 19235 + c0 * 3735 + (1 << 14)) >> 15);
            } else {
                win[ry][rx] = row[x];
            }
        }
    }
    __syncthreads();
    if (lx < nxo) {   // horizontal sums: thread -> output column, striding over the window rows
        const int rx = myx - sx0;
        for (int ry = wv; ry < sh; ry += 4) {
            // fused multiply-adds: the byte written does not depend on the rounding of these sums (error < 1e-12 either way, ties exact)
            double h = (double)win[ry][rx] * w0;
            h = __builtin_fma((double)win[ry][rx + 1], w1, h);
            h = __builtin_fma((double)win[ry][rx + 2], w2, h);
            h = __builtin_fma((double)win[ry][rx + 3], w3, h);
            hor[ry][lx] = h;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int ly = wv + 4 * it;
        if (ly >= nyo || lx >= nxo) continue;
        const int dy = oy + ly, ry = myy[it] - sy0;
        double val = hor[ry][lx] * v[it][0];
#pragma unroll
        for (int j = 1; j < 4; ++j) val = __builtin_fma(hor[ry + j][lx], v[it][j], val);
        double r = rint(val);
        if (fabs(fabs(val - r) - 0.5) < 1e-9) {                  // within 1e-9 of n + 1/2: decided exactly
            const double fl = floor(val);
            const int rx = myx - sx0;
            __int128 ex = 0;
            for (int j = 0; j < 4; ++j) {
                long long hs = 0;
                for (int k = 0; k < 4; ++k) hs += nx[dx * 4 + k] * (long long)win[ry + j][rx + k];
                ex += (__int128)ny[dy * 4 + j] * (__int128)hs;
            }
            const long long n = (long long)fl;
            const __int128 c = 2 * ex - (__int128)(2 * n + 1) * ((__int128)KX * (__int128)KY);
            r = (double)(c > 0 ? n + 1 : (c < 0 ? n : n + (n & 1)));
        }
        res[ly][lx] = (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
    }
    __syncthreads();
    if (nxo == PP_RS_TW && (dw & 3) == 0 && ((size_t)dst & 3) == 0) {   // whole dwords, 16 lanes per output row
        const int q = threadIdx.x & 15;
        for (int ly = threadIdx.x >> 4; ly < nyo; ly += 16) *(unsigned int*)(dst + (size_t)(oy + ly) * dw + ox + q * 4) = *(const unsigned int*)&res[ly][q * 4];
    } else {
        for (int i = threadIdx.x; i < TH * PP_RS_TW; i += 256) {
            const int ly = i >> 6, lx = i & 63;
            if (ly < nyo && lx < nxo) dst[(size_t)(oy + ly) * dw + ox + lx] = res[ly][lx];
        }
    }
}

// ---- cv2.GaussianBlur 3x3 (fixed-point 8.8 taps k0,k1,k2), BORDER_REFLECT_101; also accumulates the sum of the OUTPUT pixels
// (ImageEnhance.Contrast needs the mean of the blurred image)
__global__ void __launch_bounds__(256) pp_gauss3_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst, int k0, int k1, int k2,
                                                         unsigned long long* __restrict__ sum) {
    const size_t total = (size_t)H * W;
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = pp_row(i, W), x = (int)(i - (size_t)y * W);
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
        const int y = pp_row(i, W4), x = (int)(i - (size_t)y * W4) << 2;
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

// ---- CLAHE pass 1 on a W % 4 == 0 plane: the same histograms from dword loads, the enhancer LUT staged in LDS.  A tile's columns
// need not start on a dword, so a thread tests each of its four pixels against the tile's span; the reflected columns (x >= W, at most
// the grid's padding) are read byte-wise.
__global__ void __launch_bounds__(256) pp_clahe_hist_x4_kernel(const uint8_t* __restrict__ src, int H, int W, const uint8_t* __restrict__ lut, int tw,
                                                                int th, int tx, int ty, unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    __shared__ uint8_t lt[256];
    const int tile = blockIdx.x;
    const int tj = tile / tx, ti = tile - tj * tx;
    h[threadIdx.x] = 0;
    lt[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int rows_per = (th + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(th, r0 + rows_per);
    const int xlo = ti * tw, xend = xlo + tw, xhi = max(xlo, min(xend, W));   // [xlo, xhi) real columns, [xhi, xend) reflected ones
    const int g0 = xlo >> 2, ng = xlo < W ? ((xhi + 3) >> 2) - g0 : 0, nrefl = xend - xhi;
    for (int r = r0; r < r1; ++r) {
        int y = tj * th + r;
        if (y >= H) y = 2 * (H - 1) - y;                        // BORDER_REFLECT_101 (only the bottom / right are padded)
        const uint8_t* row = src + (size_t)y * W;
        for (int g = threadIdx.x; g < ng; g += 256) {
            const int x = (g0 + g) << 2;
            const unsigned int c = *(const unsigned int*)(row + x);
            if (x >= xlo && x + 4 <= xhi) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&h[lt[(c >> (8 * k)) & 255u]], 1u);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k >= xlo && x + k < xhi) atomicAdd(&h[lt[(c >> (8 * k)) & 255u]], 1u);
            }
        }
        if ((int)threadIdx.x < nrefl) atomicAdd(&h[lt[row[2 * (W - 1) - (xhi + (int)threadIdx.x)]]], 1u);
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
            const int y = pp_row(i, W4), x = (int)(i - (size_t)y * W4) << 2;
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
        const int y = pp_row(i, W), x = (int)(i - (size_t)y * W);
        dst[i] = (uint8_t)pixel(y, x, src[i]);
    }
}

// ---- PIL box blur, one pass along rows (stride_x = 1) or columns (stride_x = W): radius r, weights ww / fw, edge replication
__global__ void __launch_bounds__(256) pp_box_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int vertical, int r,
                                                           unsigned int ww, unsigned int fw) {
    const size_t total = (size_t)H * W;
    const int n = vertical ? H : W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = pp_row(i, W), x = (int)(i - (size_t)y * W);
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
        const int y = pp_row(i, W4), x = (int)(i - (size_t)y * W4) << 2;
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

// ---- the three row passes of box radius 0 in ONE launch (W % 4 == 0): a thread owns four pixels, reads the dwords left and right
// of them and carries the ten-pixel neighbourhood through the three levels in registers.  Every level rounds to 8 bits exactly as
// its stand-alone pass does; PIL replicates the edge pixel of EACH pass's input, so a level's values at positions outside the line
// are that level's own edge value (the fix-ups below), not a filter of replicated inputs.
__device__ __forceinline__ unsigned int pp_box0(unsigned int v, unsigned int a, unsigned int b, unsigned int ww, unsigned int fw) {
    return ((v * ww + (a + b) * fw) + (1u << 23)) >> 24;
}
// the four pixels row[x .. x+3] after the three row passes (dword-aligned x, W % 4 == 0)
__device__ __forceinline__ unsigned int pp_box0_h3_dw(const uint8_t* __restrict__ row, int x, int W, unsigned int ww, unsigned int fw) {
    const bool first = x == 0, last = x + 4 == W;
    const unsigned int c = *(const unsigned int*)(row + x);
    const unsigned int l = first ? (c & 255u) * 0x01010101u : *(const unsigned int*)(row + x - 4);
    const unsigned int r = last ? (c >> 24) * 0x01010101u : *(const unsigned int*)(row + x + 4);
    unsigned int in[10], y1[8], y2[6];                           // positions x-3.., x-2.., x-1..
#pragma unroll
    for (int k = 0; k < 3; ++k) { in[k] = (l >> (8 * (k + 1))) & 255u; in[7 + k] = (r >> (8 * k)) & 255u; }
#pragma unroll
    for (int k = 0; k < 4; ++k) in[3 + k] = (c >> (8 * k)) & 255u;
#pragma unroll
    for (int k = 0; k < 8; ++k) y1[k] = pp_box0(in[k + 1], in[k], in[k + 2], ww, fw);
    if (first) y1[0] = y1[1] = y1[2];
    if (last) y1[6] = y1[7] = y1[5];
#pragma unroll
    for (int k = 0; k < 6; ++k) y2[k] = pp_box0(y1[k + 1], y1[k], y1[k + 2], ww, fw);
    if (first) y2[0] = y2[1];
    if (last) y2[5] = y2[4];
    unsigned int o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) o |= pp_box0(y2[k + 1], y2[k], y2[k + 2], ww, fw) << (8 * k);
    return o;
}
__global__ void __launch_bounds__(256) pp_box0_h3_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, unsigned int ww,
                                                          unsigned int fw) {
    const int W4 = W >> 2;
    const size_t total = (size_t)H * W4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = pp_row(i, W4), x = (int)(i - (size_t)y * W4) << 2;
        *(unsigned int*)(dst + (size_t)y * W + x) = pp_box0_h3_dw(src + (size_t)y * W, x, W, ww, fw);
    }
}
// ---- the three column passes of box radius 0 AND the UnsharpMask combine in one launch: a thread owns a dword column (four pixels
// wide) of a strip of R rows, loads the R + 6 rows the three levels reach and writes only the final plane.  Level values above the
// first / below the last row are that level's own edge value, as in the row kernel.  `blur_of` = the plane after the row passes,
// `in` = the unsharp mask's input (the combine's first operand).
__device__ __forceinline__ unsigned int pp_box0_dw(unsigned int v, unsigned int a, unsigned int b, unsigned int ww, unsigned int fw) {
    unsigned int o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) o |= pp_box0((v >> (8 * k)) & 255u, (a >> (8 * k)) & 255u, (b >> (8 * k)) & 255u, ww, fw) << (8 * k);
    return o;
}
__device__ __forceinline__ unsigned int pp_unsharp_dw(unsigned int av, unsigned int bv, int percent, int threshold) {
    unsigned int ov = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = (av >> (8 * k)) & 255, diff = a - (int)((bv >> (8 * k)) & 255);
        int o = a;
        if (abs(diff) > threshold) {
            o = a + diff * percent / 100;                        // C integer division: truncates towards zero
            o = o < 0 ? 0 : (o > 255 ? 255 : o);
        }
        ov |= (unsigned)o << (8 * k);
    }
    return ov;
}
// HROWS: `blur_of` is the mask's INPUT and the three row passes are applied to each of the R + 6 rows as it is loaded (three dwords per
// row, the neighbours' from the L1) -- the whole UnsharpMask in one launch, no intermediate plane.
template <int R, bool HROWS>
__global__ void __launch_bounds__(256) pp_box0_v3_unsharp_kernel(const uint8_t* __restrict__ blur_of, const uint8_t* __restrict__ in,
                                                                  uint8_t* __restrict__ dst, int H, int W, unsigned int ww, unsigned int fw,
                                                                  int percent, int threshold) {
    const int W4 = W >> 2;
    const int x4 = blockIdx.x * 256 + threadIdx.x;
    if (x4 >= W4) return;
    const int y0 = blockIdx.y * R, x = x4 << 2;
    unsigned int a[R + 6], b[R + 4];                             // rows y0-3.. (inputs), then y0-2.. (level 1), reused for level 2 / 3
#pragma unroll
    for (int k = 0; k < R + 6; ++k) {
        int y = y0 - 3 + k;
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);
        if constexpr (HROWS) a[k] = pp_box0_h3_dw(blur_of + (size_t)y * W, x, W, ww, fw);
        else a[k] = *(const unsigned int*)(blur_of + (size_t)y * W + x);
    }
    const bool top = y0 == 0, bottom = y0 + R + 2 > H - 1;       // block-uniform
    // level 1 at rows y0-2+k
#pragma unroll
    for (int k = 0; k < R + 4; ++k) b[k] = pp_box0_dw(a[k + 1], a[k], a[k + 2], ww, fw);
    if (top) b[0] = b[1] = b[2];
    if (bottom) {
        unsigned int e = b[0];
#pragma unroll
        for (int k = 1; k < R + 4; ++k) e = (y0 - 2 + k <= H - 1) ? b[k] : e;
#pragma unroll
        for (int k = 1; k < R + 4; ++k) b[k] = (y0 - 2 + k > H - 1) ? e : b[k];
    }
    // level 2 at rows y0-1+k (into a[0 .. R+1])
#pragma unroll
    for (int k = 0; k < R + 2; ++k) a[k] = pp_box0_dw(b[k + 1], b[k], b[k + 2], ww, fw);
    if (top) a[0] = a[1];
    if (bottom) {
        unsigned int e = a[0];
#pragma unroll
        for (int k = 1; k < R + 2; ++k) e = (y0 - 1 + k <= H - 1) ? a[k] : e;
#pragma unroll
        for (int k = 1; k < R + 2; ++k) a[k] = (y0 - 1 + k > H - 1) ? e : a[k];
    }
    // level 3 at rows y0+k, combined with the mask's input
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int y = y0 + k;
        if (y < H) {
            const unsigned int bl = pp_box0_dw(a[k + 1], a[k], a[k + 2], ww, fw);
            const size_t off = (size_t)y * W + x;
            *(unsigned int*)(dst + off) = pp_unsharp_dw(*(const unsigned int*)(in + off), bl, percent, threshold);
        }
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

// ---- the two PIL enhancers folded into one LUT on the device (libImaging/Blend.c with a constant first image; same float steps as
// preprocess.cpp::pil_blend_lut).  Contrast blends against the rounded mean of the plane whose pixel sum `sum` holds; a step with
// factor <= 0 is skipped.  One block of 256 threads, one table entry each.
__device__ __forceinline__ int pp_blend1(int in1, float alpha, int v) {
    const float t = (float)in1 + alpha * (float)(v - in1);
    if (alpha >= 0.f && alpha <= 1.f) return (int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__global__ void __launch_bounds__(256) pp_fold_lut_kernel(const unsigned long long* __restrict__ sum, unsigned long long n, float contrast,
                                                           float brightness, uint8_t* __restrict__ lut) {
    int v = threadIdx.x;
    if (contrast > 0.f) v = pp_blend1((int)((double)sum[0] / (double)n + 0.5), contrast, v);
    if (brightness > 0.f) v = pp_blend1(0, brightness, v);
    lut[threadIdx.x] = (uint8_t)v;
}
// ---- CLAHE tile LUTs on the device (imgproc clahe.cpp: clip, redistribute the excess -- an equal batch to every bin and the residual one
// by one at stride 256 / residual --, LUT = cvRound(cumsum * 255 / tile_area) in float).  One block per tile, one bin per thread.
__global__ void __launch_bounds__(256) pp_clahe_luts_kernel(const unsigned int* __restrict__ hist, int clip, float lut_scale,
                                                             uint8_t* __restrict__ tile_luts) {
    __shared__ unsigned int sc[256];
    const int i = threadIdx.x;
    unsigned int h = hist[blockIdx.x * 256 + i];
    if (clip > 0) {
        const unsigned int over = h > (unsigned)clip ? h - (unsigned)clip : 0u;
        sc[i] = over;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (i < s) sc[i] += sc[i + s];
            __syncthreads();
        }
        const unsigned int clipped = sc[0];
        __syncthreads();
        if (over) h = (unsigned)clip;
        const unsigned int batch = clipped >> 8, residual = clipped & 255u;
        h += batch;
        if (residual) {
            const unsigned int step = 256u / residual;          // >= 1 since residual <= 255
            if (i % step == 0 && i / step < residual) h += 1;
        }
    }
    sc[i] = h;                                                   // inclusive scan over the bins
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned int add = i >= d ? sc[i - d] : 0u;
        __syncthreads();
        sc[i] += add;
        __syncthreads();
    }
    const int v = __float2int_rn((float)sc[i] * lut_scale);
    tile_luts[blockIdx.x * 256 + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// ---- CLAHE pass 2 on a W % 4 == 0 plane, one interpolation cell per workgroup column: between the centres of neighbouring tiles the four
// tile LUTs a pixel blends are fixed, so a workgroup stages those four (and the enhancer LUT) in LDS and its lookups never leave the
// CU -- the per-pixel kernel above spends its time on five dependent byte gathers per pixel through the vector L1.  Cells are cut by the
// SAME float expressions the per-pixel kernel evaluates: a workgroup scans a slightly wider integer range and keeps the pixels whose
// floor(x / tw - 0.5), floor(y / th - 0.5) name its cell, so every pixel is written exactly once with the identical arithmetic.
__global__ void __launch_bounds__(256) pp_clahe_apply_cell_kernel(const uint8_t* __restrict__ src, int H, int W, const uint8_t* __restrict__ lut,
                                                                   const uint8_t* __restrict__ tile_luts, int tw, int th, int tx, int ty,
                                                                   uint8_t* __restrict__ dst) {
    __shared__ uint8_t L[4][256];
    __shared__ uint8_t lt[256];
    const int cy = blockIdx.x / (tx + 1), cx = blockIdx.x - cy * (tx + 1);     // cell = (floor(tyf) + 1, floor(txf) + 1)
    const int ty1 = max(cy - 1, 0), ty2 = min(cy, ty - 1), tx1 = max(cx - 1, 0), tx2 = min(cx, tx - 1);
    L[0][threadIdx.x] = tile_luts[(ty1 * tx + tx1) * 256 + threadIdx.x];
    L[1][threadIdx.x] = tile_luts[(ty1 * tx + tx2) * 256 + threadIdx.x];
    L[2][threadIdx.x] = tile_luts[(ty2 * tx + tx1) * 256 + threadIdx.x];
    L[3][threadIdx.x] = tile_luts[(ty2 * tx + tx2) * 256 + threadIdx.x];
    lt[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const float inv_th = 1.0f / (float)th, inv_tw = 1.0f / (float)tw;
    const int xa = max(0, (int)(((float)cx - 0.5f) * (float)tw) - 2), xb = min(W, (int)(((float)cx + 0.5f) * (float)tw) + 3);
    const int ya = max(0, (int)(((float)cy - 0.5f) * (float)th) - 2), yb = min(H, (int)(((float)cy + 0.5f) * (float)th) + 3);
    if (xa >= xb || ya >= yb) return;
    const int rows_per = (yb - ya + gridDim.y - 1) / gridDim.y;
    const int r0 = ya + blockIdx.y * rows_per, r1 = min(yb, r0 + rows_per);
    const int g0 = xa >> 2, ng = ((xb + 3) >> 2) - g0;
    for (int i = threadIdx.x; i < (r1 - r0) * ng; i += 256) {
        const int rr = i / ng, y = r0 + rr, x = (g0 + (i - rr * ng)) << 2;
        const float tyf = (float)y * inv_th - 0.5f;
        const int fy = (int)floorf(tyf);
        if (fy != cy - 1) continue;
        const float ya_ = tyf - (float)fy, ya1 = 1.0f - ya_;
        const unsigned int c = *(const unsigned int*)(src + (size_t)y * W + x);
        unsigned int o = 0, own = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float txf = (float)(x + k) * inv_tw - 0.5f;
            const int fx = (int)floorf(txf);
            const float xa_ = txf - (float)fx, xa1 = 1.0f - xa_;
            const int v = lt[(c >> (8 * k)) & 255u];
            const float res = ((float)L[0][v] * xa1 + (float)L[1][v] * xa_) * ya1 + ((float)L[2][v] * xa1 + (float)L[3][v] * xa_) * ya_;
            const int r = __float2int_rn(res);                   // cvRound: round half to even
            o |= (unsigned)(r < 0 ? 0 : (r > 255 ? 255 : r)) << (8 * k);
            own |= (fx == cx - 1 ? 1u : 0u) << k;
        }
        uint8_t* out = dst + (size_t)y * W + x;
        if (own == 15u) *(unsigned int*)out = o;
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((own >> k) & 1u) out[k] = (uint8_t)(o >> (8 * k));
        }
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

// which tile height the LDS-tiled kernel takes for this geometry (0: none fits, the per-pixel kernel runs)
int pp_resize_tile_rows(int H, int W, int dh, int dw) {
    const long long sw = ((long long)PP_RS_TW * W + dw - 1) / dw + 5;
    const long long sh32 = (32LL * H + dh - 1) / dh + 5, sh16 = (16LL * H + dh - 1) / dh + 5;
    if (sw <= PP_RS_SW && sh32 <= PP_RS_SH && (dh + 31) / 32 <= 65535) return 32;
    if (sw <= PP_RS_SW && sh16 <= PP_RS_SH && (dh + 15) / 16 <= 65535) return 16;
    return 0;
}
// src_bgr != 0: src is the interleaved 3-channel page (only where pp_resize_tile_rows() != 0)
hipError_t launch_pp_resize_cubic(const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw, const int* x0, const double* wx, const long long* nx,
                                  const int* y0, const double* wy, const long long* ny, unsigned long long KX, unsigned long long KY, hipStream_t s,
                                  int src_bgr) {
    // window of a 64 x 16 output tile: at most ceil(tile * src / dst) + 4 source columns / rows
    const int th = pp_resize_tile_rows(H, W, dh, dw);
    const dim3 grid((dw + PP_RS_TW - 1) / PP_RS_TW, th ? (dh + th - 1) / th : 1);
#define PP_RS_LAUNCH(TH_, BGR_) hipLaunchKernelGGL((pp_resize_cubic_tiled_kernel<TH_, BGR_>), grid, dim3(256), 0, s, src, H, W, dst, dh, dw, x0, wx, nx, y0, wy, ny, KX, KY)
    if (th == 32) {
        if (src_bgr) PP_RS_LAUNCH(32, true); else PP_RS_LAUNCH(32, false);
        return hipGetLastError();
    }
    if (th == 16) {
        if (src_bgr) PP_RS_LAUNCH(16, true); else PP_RS_LAUNCH(16, false);
        return hipGetLastError();
    }
#undef PP_RS_LAUNCH
    if (src_bgr) return hipErrorInvalidValue;
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
    if ((W & 3) == 0 && ((size_t)src & 3) == 0) {
        hipLaunchKernelGGL(pp_clahe_hist_x4_kernel, dim3(tx * ty, 32), dim3(256), 0, s, src, H, W, lut, tw, th, tx, ty, hist);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pp_clahe_hist_kernel, dim3(tx * ty, 16), dim3(256), 0, s, src, H, W, lut, tw, th, tx, ty, hist);
    return hipGetLastError();
}
hipError_t launch_pp_clahe_apply(const uint8_t* src, int H, int W, const uint8_t* lut, const uint8_t* tile_luts, int tw, int th, int tx, int ty,
                                 uint8_t* dst, hipStream_t s) {
    const int vec = (W & 3) == 0 && (((size_t)src | (size_t)dst) & 3) == 0;
    if (vec && (size_t)H * W >= (size_t)1 << 14) {            // small planes: too few pixels per cell to pay for staging the tables
        const int chunks = th >= 512 ? 32 : (th >= 64 ? 8 : 2);
        hipLaunchKernelGGL(pp_clahe_apply_cell_kernel, dim3((tx + 1) * (ty + 1), chunks), dim3(256), 0, s, src, H, W, lut, tile_luts, tw, th, tx, ty,
                           dst);
        return hipGetLastError();
    }
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
// UnsharpMask with box radius 0 on a W % 4 == 0 plane: three row passes, then three column passes + combine (two launches for seven)
bool pp_unsharp_fused_ok(int H, int W, int r, const uint8_t* a, const uint8_t* b, const uint8_t* c) {
    return r == 0 && (W & 3) == 0 && W >= 4 && H >= 1 && ((((size_t)a | (size_t)b | (size_t)c)) & 3) == 0 && (H + 15) / 16 <= 65535;
}
hipError_t launch_pp_unsharp_fused(const uint8_t* in, uint8_t* tmp, uint8_t* dst, int H, int W, unsigned int ww, unsigned int fw, int percent,
                                   int threshold, hipStream_t s) {
    if (!tmp) {                                                  // one launch: row passes applied on the fly
        hipLaunchKernelGGL((pp_box0_v3_unsharp_kernel<16, true>), dim3(((W >> 2) + 255) / 256, (H + 15) / 16), dim3(256), 0, s, in, in, dst, H, W, ww, fw,
                           percent, threshold);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pp_box0_h3_kernel, dim3(pp_grid((size_t)H * (W >> 2))), dim3(256), 0, s, in, tmp, H, W, ww, fw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pp_box0_v3_unsharp_kernel<16, false>), dim3(((W >> 2) + 255) / 256, (H + 15) / 16), dim3(256), 0, s, tmp, in, dst, H, W, ww, fw,
                       percent, threshold);
    return hipGetLastError();
}
hipError_t launch_pp_unsharp(const uint8_t* in, const uint8_t* blur, uint8_t* dst, size_t total, int percent, int threshold, hipStream_t s) {
    hipLaunchKernelGGL(pp_unsharp_kernel, dim3(pp_grid((total >> 2) + 1)), dim3(256), 0, s, in, blur, dst, total, percent, threshold);
    return hipGetLastError();
}
hipError_t launch_pp_fold_lut(const unsigned long long* sum, unsigned long long n, float contrast, float brightness, uint8_t* lut, hipStream_t s) {
    hipLaunchKernelGGL(pp_fold_lut_kernel, dim3(1), dim3(256), 0, s, sum, n, contrast, brightness, lut);
    return hipGetLastError();
}
hipError_t launch_pp_clahe_luts(const unsigned int* hist, int tiles, int clip, float lut_scale, uint8_t* tile_luts, hipStream_t s) {
    hipLaunchKernelGGL(pp_clahe_luts_kernel, dim3(tiles), dim3(256), 0, s, hist, clip, lut_scale, tile_luts);
    return hipGetLastError();
}
hipError_t launch_pp_lut(const uint8_t* src, uint8_t* dst, const uint8_t* lut, size_t total, hipStream_t s) {
    hipLaunchKernelGGL(pp_lut_kernel, dim3(pp_grid(total)), dim3(256), 0, s, src, dst, lut, total);
    return hipGetLastError();
}
