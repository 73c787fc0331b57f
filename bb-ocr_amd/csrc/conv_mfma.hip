// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4), NHWC bf16 in, fp32 accumulate, bf16/fp32 out.
//
// Replaces, for the detector and recogniser, the ATen conv2d+batch_norm+relu sequence that
// easyocr's CRAFT / VGG_FeatureExtractor run under `reader.readtext` (reference call site
// pipeline_demo/extractor/enhanced_extractor.py:520; upstream easyocr/craft.py, model/modules.py).
//
// Mapping (D = A x B per v_mfma_f32_16x16x32_bf16):
//   A = weights   [16 couts][32 k]   lane l: row l&15, k = 8(l>>4)+j      (LDS image == global packed image, glds copy)
//   B = activation[32 k][16 pixels]  lane l: pixel l&15, k = 8(l>>4)+j    (LDS patch [4 channel-groups][NP pixels] x 16 B)
//   D             [16 couts][16 px]  lane l: pixel l&15, couts 4(l>>4)+r  -> cout permutation in the packed weights makes
//                                     each lane own 16 CONTIGUOUS couts of one pixel: 2 x 16-B NHWC stores per fragment.
// The activation patch (tile + halo) of one 32-channel chunk is staged once and re-read for all KHxKW taps; weights of
// one (chunk, tap) k-step stream through a 2-deep LDS ring via LDS-DMA.  One workgroup = WM x WN waves, wave tile =
// (MF*16 pixels) x 64 couts.  Launch grid is XCD-remapped so the cout tiles of one pixel tile share an L2.
#include "common.h"
#include "kernels.h"
#include <string.h>

template <int WM, int WN, int MF, int PITER>
__global__ void __launch_bounds__(WM * WN * 64) conv_mfma_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN, NT = NW * 64, BN = WN * 64;
    constexpr int WBUF = BN * 64;          // bytes of one weight k-step slice (BN couts x 32 k x 2 B)
    constexpr int WPIECES = WBUF / 16;     // 16-B pieces per slice
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wbuf = smem;                  // [2][WBUF]
    unsigned char* const pbuf = smem + 2 * WBUF;       // [2][NP*64]
    const int patch_bytes = a.NP * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % a.ntiles_n;
    bid /= a.ntiles_n;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int oy0 = ty * a.TH, ox0 = tx * a.TW;
    const int iy0 = oy0 - a.pad_h, ix0 = ox0 - a.pad_w;
    const int nk = a.nchunks * a.ntaps;

    // ---- patch loader: one wave-instruction = 16 patch pixels x 4 channel groups (64 B contiguous per pixel).
    // lane -> (pixel, group) chosen so each 8-lane ds_write group hits 8 consecutive 16-B slots of one group.
    const int n_wi = a.NP >> 4;
    const int l_pix = (lane & 7) + ((lane >> 5) << 3);
    const int l_kg = (lane >> 3) & 3;
    int src_pix[PITER];   // input pixel index (n,iy,ix flattened) or -1 when the slot is zero padding
#pragma unroll
    for (int it = 0; it < PITER; ++it) {
        const int wi = wave + it * NW;
        int sp = -1;
        if (wi < n_wi) {
            const int pix = wi * 16 + l_pix;
            const int py = pix / a.PW, px = pix - py * a.PW;
            const int iy = iy0 + py, ix = ix0 + px;
            if (py < a.PH && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) sp = (n * a.H + iy) * a.W + ix;
        }
        src_pix[it] = sp;
    }
    u32x4 pre[PITER];
    auto load_patch = [&](int chunk) {
        const int c = chunk * 32;
        const bool s0 = c < a.C0;
        const uint16_t* src = s0 ? a.in0 : a.in1;
        const int cs = s0 ? a.in0_cs : a.in1_cs;
        const int cb = (s0 ? c : c - a.C0) + l_kg * 8;
        const bool relu = s0 ? a.relu_in0 : a.relu_in1;
#pragma unroll
        for (int it = 0; it < PITER; ++it) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (src_pix[it] >= 0) {
                v = *(const u32x4*)(src + (size_t)src_pix[it] * cs + cb);
                if (relu) {
                    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    v = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
                }
            }
            pre[it] = v;
        }
    };
    auto store_patch = [&](int buf) {
        unsigned char* pb = pbuf + buf * patch_bytes;
#pragma unroll
        for (int it = 0; it < PITER; ++it) {
            const int wi = wave + it * NW;
            if (wi < n_wi) *(u32x4*)(pb + (size_t)(l_kg * a.NP + wi * 16 + l_pix) * 16) = pre[it];
        }
    };
    // ---- weight slice: LDS-DMA, LDS image == global image (fragment order), lane-linear
    const unsigned char* wsrc = (const unsigned char*)a.wpk + (size_t)nt * nk * WBUF;
    auto issue_w = [&](int ks, int buf) {
#pragma unroll
        for (int p0 = 0; p0 < WPIECES; p0 += NT) {
            if (p0 + wave * 64 < WPIECES) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wsrc + (size_t)ks * WBUF + (size_t)(p0 + tid) * 16),
                    (__attribute__((address_space(3))) void*)(wbuf + buf * WBUF + (p0 + wave * 64) * 16), 16, 0, 0);
            }
        }
    };

    // ---- per-wave fragment geometry
    const int fpr = a.TW >> 4;   // 16-pixel fragments per tile row
    int frag_off[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int F = wm * MF + f;
        const int fr = F / fpr, fc = F - fr * fpr;
        frag_off[f] = (fr * a.PW + fc * 16) * 16;
    }
    const int lane_patch_off = ((lane >> 4) * a.NP + (lane & 15)) * 16;
    const int lane_w_off = wn * 4 * 1024 + lane * 16;

    f32x4 acc[MF][4];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_patch(0);
    issue_w(0, 0);
    store_patch(0);
    __syncthreads();

    int ks = 0;
    for (int c = 0; c < a.nchunks; ++c) {
        const bool more = (c + 1 < a.nchunks);
        if (more) load_patch(c + 1);
        const unsigned char* pbase = pbuf + (c & 1) * patch_bytes + lane_patch_off;
        int ky = 0, kx = 0;
        for (int tap = 0; tap < a.ntaps; ++tap, ++ks) {
            if (ks + 1 < nk) issue_w(ks + 1, (ks + 1) & 1);
            const unsigned char* wb = wbuf + (ks & 1) * WBUF + lane_w_off;
            bf16x8 af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *(const bf16x8*)(wb + j * 1024);
            const unsigned char* pb = pbase + ((ky * a.PW + kx) * a.dil) * 16;
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const bf16x8 bfr = *(const bf16x8*)(pb + frag_off[f]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bfr, acc[f][j], 0, 0, 0);
            }
            if (more && tap == a.ntaps - 1) store_patch((c + 1) & 1);
            __syncthreads();
            if (++kx == a.KW) { kx = 0; ++ky; }
        }
    }

    // ---- epilogue: bias (+ReLU), each lane owns 16 contiguous couts of its pixel per fragment
    const int g = lane >> 4, pl = lane & 15;
    const int cout0 = nt * BN + wn * 64 + g * 16;
    if (cout0 < a.cout_store) {
        float bs[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = *(const f32x4*)(a.bias + cout0 + j * 4);
            bs[j * 4 + 0] = b4[0]; bs[j * 4 + 1] = b4[1]; bs[j * 4 + 2] = b4[2]; bs[j * 4 + 3] = b4[3];
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int F = wm * MF + f;
            const int fr = F / fpr, fc = F - fr * fpr;
            const int oy = oy0 + fr, ox = ox0 + fc * 16 + pl;
            if (oy < a.OH && ox < a.OW) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[f][j][r] + bs[j * 4 + r];
                        if (a.relu_out) x = fmaxf(x, 0.f);
                        v[j * 4 + r] = x;
                    }
                const size_t o = ((size_t)(n * a.OH + oy) * a.OW + ox) * a.out_cs + cout0;
                if (a.out_f32) {
                    float* op = (float*)a.out + o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) *(f32x4*)(op + j * 4) = (f32x4){v[j * 4], v[j * 4 + 1], v[j * 4 + 2], v[j * 4 + 3]};
                } else {
                    uint16_t* op = (uint16_t*)a.out + o;
                    u32x4 lo = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                    u32x4 hi = {pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])};
                    *(u32x4*)(op) = lo;
                    *(u32x4*)(op + 8) = hi;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

size_t conv_packed_elems(const ConvPlan& p) {
    return (size_t)p.Cout_pad * p.Cin_pad * p.KH * p.KW;
}

void pack_conv_weights(const ConvPlan& p, const float* w, uint16_t* out) {
    const int BN = p.BN, ntn = p.Cout_pad / BN, nch = p.Cin_pad / 32, ntaps = p.KH * p.KW, nfr = BN / 16;
    size_t o = 0;
    for (int nt = 0; nt < ntn; ++nt)
        for (int c = 0; c < nch; ++c)
            for (int tap = 0; tap < ntaps; ++tap)
                for (int fr = 0; fr < nfr; ++fr) {
                    const int wn = fr >> 2, nf = fr & 3;
                    for (int l = 0; l < 64; ++l) {
                        const int row = l & 15;
                        const int cout = nt * BN + wn * 64 + (row >> 2) * 16 + nf * 4 + (row & 3);
                        for (int j = 0; j < 8; ++j) {
                            const int cin = c * 32 + 8 * (l >> 4) + j;
                            float v = 0.f;
                            if (cout < p.Cout && cin < p.Cin) v = w[((size_t)cout * p.Cin + cin) * ntaps + tap];
                            out[o++] = f32_to_bf16_host(v);
                        }
                    }
                }
}

template <int WM, int WN, int MF, int PITER>
static hipError_t launch_cfg(const ConvArgs& a, size_t smem, int grid, hipStream_t s) {
    auto k = conv_mfma_kernel<WM, WN, MF, PITER>;
    static size_t cur = 0;   // per-instantiation high-water mark of the opt-in dynamic LDS size
    if (smem > cur) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        cur = smem;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_conv(const ConvPlan& p, ConvArgs a, hipStream_t s) {
    const int BN = p.BN;
    const int BM = (BN == 256) ? 256 : 512;
    a.KH = p.KH; a.KW = p.KW; a.pad_h = p.pad_h; a.pad_w = p.pad_w; a.dil = p.dil;
    a.OH = a.H + 2 * p.pad_h - (p.KH - 1) * p.dil;
    a.OW = a.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    if (a.OH <= 0 || a.OW <= 0) return hipErrorInvalidValue;
    a.ntaps = p.KH * p.KW;
    // tile shape: the TH x (BM/TH) rectangle with the least (MFMA work on partial tiles + patch staging) per layer
    {
        long long best = -1;
        for (int th = 16; th >= 4; th >>= 1) {
            const int tw = BM / th;
            const int ph = th + (p.KH - 1) * p.dil, pw = tw + (p.KW - 1) * p.dil;
            const int np = cdiv(ph * pw, 16) * 16;
            if (cdiv(np / 16, 8) > 8) continue;
            const long long tiles = (long long)cdiv(a.OH, th) * cdiv(a.OW, tw);
            const long long cost = tiles * ((long long)BM * a.ntaps + 2LL * np);
            if (best < 0 || cost < best) { best = cost; a.TH = th; a.TW = tw; a.PH = ph; a.PW = pw; a.NP = np; }
        }
        if (best < 0) return hipErrorInvalidValue;
    }
    a.tiles_x = cdiv(a.OW, a.TW);
    a.tiles_y = cdiv(a.OH, a.TH);
    a.ntiles_n = p.Cout_pad / BN;
    a.nchunks = p.Cin_pad / 32;
    if (a.C0 + a.C1 != p.Cin_pad || (a.C0 & 31) || (a.C1 & 31)) return hipErrorInvalidValue;
    if ((a.in0_cs & 7) || (a.C1 && (a.in1_cs & 7)) || (a.out_cs & (a.out_f32 ? 3 : 7)) || (a.cout_store & 15)) return hipErrorInvalidValue;
    a.wpk = p.d_w;
    a.bias = p.d_b;
    const size_t smem = (size_t)2 * BN * 64 + (size_t)2 * a.NP * 64;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    const long long grid_ll = (long long)a.N * a.tiles_x * a.tiles_y * a.ntiles_n;
    if (grid_ll <= 0 || grid_ll > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)grid_ll;
    const int piter = cdiv(a.NP / 16, 8);
    if (piter > 8) return hipErrorInvalidValue;
    if (BN == 256) return piter <= 4 ? launch_cfg<2, 4, 8, 4>(a, smem, grid, s) : launch_cfg<2, 4, 8, 8>(a, smem, grid, s);
    if (BN == 128) return piter <= 4 ? launch_cfg<4, 2, 8, 4>(a, smem, grid, s) : launch_cfg<4, 2, 8, 8>(a, smem, grid, s);
    if (BN == 64) return piter <= 4 ? launch_cfg<8, 1, 4, 4>(a, smem, grid, s) : launch_cfg<8, 1, 4, 8>(a, smem, grid, s);
    return hipErrorInvalidValue;
}
