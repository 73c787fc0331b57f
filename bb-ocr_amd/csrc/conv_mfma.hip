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
//                                     each lane own two runs of 8 contiguous couts: per store instruction the four lane
//                                     groups of a pixel write 64 contiguous bytes (see conv_epilogue).
// The activation patch (tile + halo) of one 32-channel chunk is staged once and re-read for all KHxKW taps; weights of
// one (chunk, tap) k-step stream through a 2-deep LDS ring via LDS-DMA.  One workgroup = WM x WN waves, wave tile =
// (MF*16 pixels) x 64 couts.  Launch grid is XCD-remapped so the cout tiles of one pixel tile share an L2.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

// Timing-only ablations (BBOCR_CONV_DBG) and the per-tile s_memtime timeline (BBOCR_CONV_STAMPS) produce garbage results / extra
// syncs: they exist in diagnostic builds only (make DIAG=1 -> -DBBOCR_DIAG), the shipped library has neither the branches nor the knobs.
#ifdef BBOCR_DIAG
#define CONV_DBG(a, bit) ((a).dbg & (bit))
#else
#define CONV_DBG(a, bit) false
#endif

// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4: lane l's 16 bytes land at lds + 16 l), issued as inline assembly.  The builtin
// (__builtin_amdgcn_global_load_lds) is modelled by hipcc's waitcnt pass as a FLAT access that touches LDS: while one is outstanding -- in
// these kernels always, their vmcnt waits are counted, never 0 -- every LDS dependency is waited for with a full `s_waitcnt lgkmcnt(0)`,
// so ds_reads issued ahead of their use could never stay in flight across an MFMA group.  Behind the asm the compiler tracks only the
// ds_reads (counted lgkmcnt); ordering against the DMA'd bytes is what the kernels' explicit `s_waitcnt vmcnt(N)` + s_barrier do anyway.
// `lds` must be wave-uniform.  M0 is written here and by nothing else in these kernels.
__device__ __forceinline__ void lds_dma16(const void* gptr, const void* lds) {
    const unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(la) : "memory");
}

// ---- shared epilogue: bias (+ReLU) (+fused max-pool).  Lane (pixel pl, group g) holds, per 16-pixel fragment, 16 outputs
// acc[j][r]; the cout permutation of pack_conv_weights maps them to cout = tile + wn*64 + (j>>1)*32 + g*8 + (j&1)*4 + r:
// TWO runs of 8 contiguous couts, 32 apart, so that one store instruction writes, for every pixel, 64 contiguous bytes
// (4 lane groups x 16 B) instead of four 16-byte pieces with gaps.  v[i], i = j*4+r: run h = i>>3, position i&7.
template <int EL, int MF, bool ADDUP = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MF][4], int n, int nt, int oy0, int ox0, int wm, int wn, int fpr,
                                              int lane, int BN, int sub = 1, int sph = 0, int spw = 0) {
    const int g = lane >> 4, pl = lane & 15;
    const int cout0 = nt * BN + wn * 64 + g * 8;        // first cout of run 0; run 1 starts at cout0 + 32
    if (a.tail) {
        // CRAFT classifier tail fused behind conv_cls.4 (3x3 32->16 + ReLU): conv_cls.6 (1x1 16->16 + ReLU) as ONE MFMA per
        // fragment, conv_cls.8 (1x1 16->2) as 8 FMAs + a cross-group shuffle reduction; fp32 heat-map [N,h,w,2] out.
        // Run 0 of lane groups 0 and 1 holds channels 8g..8g+7 of the pixel: exactly the B fragment (k = 8g+i) of the next
        // MFMA.  a.tail = {b1[16], w2[32], b2[2]} fp32, a.tail_frag = W1 as an MFMA A fragment (row = out channel, k = in
        // channel, zero beyond 16).
        const float* tw = a.tail;
        const typename El<EL>::v8 w1f = *(const typename El<EL>::v8*)(a.tail_frag + lane * 8);
        float cb[8], b1r[4], w2a[4], w2b[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) cb[i] = a.bias[(g & 1) * 8 + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) { b1r[r] = tw[4 * g + r]; w2a[r] = tw[16 + 4 * g + r]; w2b[r] = tw[32 + 4 * g + r]; }
        const float b20 = tw[48], b21 = tw[49];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int F = wm * MF + f;
            const int fr = F / fpr, fc = F - fr * fpr;
            const int oy = (oy0 + fr) * sub + sph, ox = (ox0 + fc * 16 + pl) * sub + spw;
            u32x4 bb;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float x0 = fmaxf(acc[f][j][0] + cb[j * 4 + 0], 0.f), x1 = fmaxf(acc[f][j][1] + cb[j * 4 + 1], 0.f);
                const float x2 = fmaxf(acc[f][j][2] + cb[j * 4 + 2], 0.f), x3 = fmaxf(acc[f][j][3] + cb[j * 4 + 3], 0.f);
                bb[2 * j] = g < 2 ? El<EL>::pack2(x0, x1) : 0u;
                bb[2 * j + 1] = g < 2 ? El<EL>::pack2(x2, x3) : 0u;
            }
            const f32x4 d = El<EL>::mfma(w1f, __builtin_bit_cast(typename El<EL>::v8, bb), (f32x4){0.f, 0.f, 0.f, 0.f});
            float p0 = 0.f, p1 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hv = fmaxf(d[r] + b1r[r], 0.f);
                p0 = fmaf(w2a[r], hv, p0);
                p1 = fmaf(w2b[r], hv, p1);
            }
            p0 += __shfl_xor(p0, 16); p1 += __shfl_xor(p1, 16);
            p0 += __shfl_xor(p0, 32); p1 += __shfl_xor(p1, 32);
            if (g == 0 && oy < a.OH && ox < a.OW) *(float2*)((float*)a.out + ((size_t)(n * a.OH + oy) * a.OW + ox) * 2) = make_float2(p0 + b20, p1 + b21);
        }
        return;
    }
    const bool run0 = cout0 < a.cout_store, run1 = cout0 + 32 < a.cout_store;
    if (!run0) return;
    const float sc = a.acc_scale;       // 1 unless the packed weights carry a power-of-two scale (fma(x, 1, b) == x + b)
    float bs[16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 b4 = *(const f32x4*)(a.bias + cout0 + h * 32 + q * 4);
            bs[h * 8 + q * 4 + 0] = b4[0]; bs[h * 8 + q * 4 + 1] = b4[1]; bs[h * 8 + q * 4 + 2] = b4[2]; bs[h * 8 + q * 4 + 3] = b4[3];
        }
    // Store the 16 pixels of one fragment row: lane (pl, g) owns v[16] of pixel x(pl) = (xb + pl) * sub + spw of the row whose
    // first element is `row` (elements).  When the wave's 64 couts are all stored (`fullw`) the two 32-byte runs of a pixel
    // are regrouped across lanes pl <-> pl^8 (two DPP row shifts per dword) so that every store instruction writes WHOLE
    // 128-byte lines (8 pixels x 128 B) instead of 16 half lines: the CU's store path is priced per line touched
    // (tools/micro/store_patterns.hip: 64-KB tile burst 5.9k -> 3.9k cycles).
    const bool fullw = !a.out_f32 && nt * BN + wn * 64 + 64 <= a.cout_store;   // wave-uniform
    int spw_cur = spw;                                                         // x phase of the fragment row being stored
    auto pack_runs = [&](const float (&v)[16], u32x4& lo, u32x4& hi) {
        lo = (u32x4){El<EL>::pack2(v[0], v[1]), El<EL>::pack2(v[2], v[3]), El<EL>::pack2(v[4], v[5]), El<EL>::pack2(v[6], v[7])};
        hi = (u32x4){El<EL>::pack2(v[8], v[9]), El<EL>::pack2(v[10], v[11]), El<EL>::pack2(v[12], v[13]), El<EL>::pack2(v[14], v[15])};
    };
    auto store_frag_at = [&](void* base, size_t row, int cs, int xb, int xlim, bool row_ok, const float (&v)[16], bool f32, int coff) {
        if (fullw) {
            u32x4 lo, hi, dA, dB;
            pack_runs(v, lo, hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // A: lanes pl < 8 keep their run 0, lanes pl >= 8 take run 1 of lane pl-8  -> pixels 0..7 of the fragment
                dA[i] = (unsigned)__builtin_amdgcn_update_dpp((int)lo[i], (int)hi[i], 0x118 /*row_shr:8*/, 0xF, 0xC, false);
                // B: lanes pl >= 8 keep their run 1, lanes pl < 8 take run 0 of lane pl+8  -> pixels 8..15
                dB[i] = (unsigned)__builtin_amdgcn_update_dpp((int)hi[i], (int)lo[i], 0x108 /*row_shl:8*/, 0xF, 0x3, false);
            }
            const int co = cout0 + (pl >> 3) * 32 + coff;
            const int xA = (xb + (pl & 7)) * sub + spw_cur, xB = (xb + 8 + (pl & 7)) * sub + spw_cur;
            uint16_t* op = (uint16_t*)base + row + co;
            if (row_ok && xA < xlim) *(u32x4*)(op + (size_t)xA * cs) = dA;
            if (row_ok && xB < xlim) *(u32x4*)(op + (size_t)xB * cs) = dB;
            return;
        }
        const int x = (xb + pl) * sub + spw_cur;
        if (!(row_ok && x < xlim)) return;
        if (f32) {
            float* op = (float*)base + row + (size_t)x * cs + cout0;
            *(f32x4*)(op) = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            if (run1) {
                *(f32x4*)(op + 32) = (f32x4){v[8], v[9], v[10], v[11]};
                *(f32x4*)(op + 36) = (f32x4){v[12], v[13], v[14], v[15]};
            }
        } else {
            u32x4 lo, hi;
            pack_runs(v, lo, hi);
            uint16_t* op = (uint16_t*)base + row + (size_t)x * cs + cout0 + coff;
            *(u32x4*)(op) = lo;
            if (run1) *(u32x4*)(op + 32) = hi;
        }
    };
    // a.split_off (exact recogniser mode, fp16): v = hi + lo / 2048 with hi = fp16(v), lo = fp16((v - hi) * 2048): 22 significand bits in
    // two fp16 tensors, the low one scaled into fp16's normal range (its weight block carries the 2^-11, weights.cpp::upload_split_plan)
    auto split_hi_lo = [&](const float (&v)[16], float (&vh)[16], float (&vl)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            vh[i] = El<EL>::to_f32(El<EL>::from_f32(v[i]));
            vl[i] = (v[i] - vh[i]) * SPLIT_LO_SCALE;
        }
    };
    auto store_frag = [&](void* base, size_t row, int cs, int xb, int xlim, bool row_ok, const float (&v)[16], bool f32) {
        if (a.split_off) {
            float vh[16], vl[16];
            split_hi_lo(v, vh, vl);
            store_frag_at(base, row, cs, xb, xlim, row_ok, vh, false, 0);
            store_frag_at(base, row, cs, xb, xlim, row_ok, vl, false, a.split_off);
            return;
        }
        store_frag_at(base, row, cs, xb, xlim, row_ok, v, f32, 0);
    };
    if constexpr (ADDUP) {
        // 1x1 launch over the flattened N*up_H*up_W pixel axis whose epilogue adds up(z): F.interpolate(scale 2, bilinear,
        // align_corners=False) of the low-resolution tensor z = a.addup -- source coordinate (o + 0.5)/2 - 0.5 clamped at 0,
        // neighbour clamped at the far edge, four fp32 blend weights per pixel.  The 8 gather loads of fragment f+1 are issued
        // before fragment f is blended and stored (two register sets), otherwise every fragment would wait a full L2 round trip.
        // (launch_conv guarantees cout_store % 64 == 0 here: both runs of every lane exist.)
        struct Gather { u32x4 q[8]; f32x2_t w[4]; };
        const int hw = a.up_H * a.up_W, LH = a.up_H >> 1, LW = a.up_W >> 1;
        const float rhw = 1.f / (float)hw, rw = 1.f / (float)a.up_W;
        auto issue = [&](int f, Gather& G) {
            int p = ox0 + (wm * MF + f) * 16 + pl;
            p = p < a.OW ? p : a.OW - 1;                      // beyond the end: any valid address, the store is predicated
            int ni = (int)((float)p * rhw);
            int rem = p - ni * hw;
            if (rem < 0) { --ni; rem += hw; } else if (rem >= hw) { ++ni; rem -= hw; }
            int yy = (int)((float)rem * rw);
            int xx = rem - yy * a.up_W;
            if (xx < 0) { --yy; xx += a.up_W; } else if (xx >= a.up_W) { ++yy; xx -= a.up_W; }
            float sy = ((float)yy + 0.5f) * 0.5f - 0.5f, sx = ((float)xx + 0.5f) * 0.5f - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < LH - 1 ? 1 : 0), x1 = x0 + (x0 < LW - 1 ? 1 : 0);
            const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
            G.w[0] = (f32x2_t){hy * hx, hy * hx}; G.w[1] = (f32x2_t){hy * lx, hy * lx};
            G.w[2] = (f32x2_t){ly * hx, ly * hx}; G.w[3] = (f32x2_t){ly * lx, ly * lx};
            const uint16_t* zb = a.addup + (size_t)ni * LH * LW * a.up_cs + cout0;
            const uint16_t* z00 = zb + ((size_t)y0 * LW + x0) * a.up_cs;
            const uint16_t* z01 = zb + ((size_t)y0 * LW + x1) * a.up_cs;
            const uint16_t* z10 = zb + ((size_t)y1 * LW + x0) * a.up_cs;
            const uint16_t* z11 = zb + ((size_t)y1 * LW + x1) * a.up_cs;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                G.q[h * 4 + 0] = *(const u32x4*)(z00 + h * 32); G.q[h * 4 + 1] = *(const u32x4*)(z01 + h * 32);
                G.q[h * 4 + 2] = *(const u32x4*)(z10 + h * 32); G.q[h * 4 + 3] = *(const u32x4*)(z11 + h * 32);
            }
        };
        auto two = [](unsigned int q) { return El<EL>::unpack2(q); };
        Gather G[2];
        issue(0, G[0]);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            if (f + 1 < MF) issue(f + 1, G[(f + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);               // keep the next fragment's loads ahead of this fragment's math
            const Gather& C = G[f & 1];
            float v[16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x2_t t = two(C.q[h * 4][i]) * C.w[0];
                    t = __builtin_elementwise_fma(two(C.q[h * 4 + 1][i]), C.w[1], t);
                    t = __builtin_elementwise_fma(two(C.q[h * 4 + 2][i]), C.w[2], t);
                    t = __builtin_elementwise_fma(two(C.q[h * 4 + 3][i]), C.w[3], t);
                    const int k = h * 8 + i * 2;             // v index j*4+r == run h, position 2i / 2i+1
                    float x0 = acc[f][k >> 2][k & 3] + bs[k] + t[0], x1 = acc[f][(k + 1) >> 2][(k + 1) & 3] + bs[k + 1] + t[1];
                    if (a.relu_out) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
                    v[k] = x0; v[k + 1] = x1;
                }
            store_frag(a.out, 0, a.out_cs, ox0 + (wm * MF + f) * 16, a.OW, true, v, false);
        }
        return;
    }
    if (a.pool_mode == 0) {
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int F = wm * MF + f;
            const int fr = F / fpr, fc = F - fr * fpr;
            // sub > 1: dilated conv run as sub*sub plain convs on the phase sub-lattices, whose images are stacked along y
            // (a.stack rows + one zero row each): tile row -> phase (sph, spw) and row inside the phase image
            int oy = oy0 + fr;
            bool rowok = true;
            if (sub > 1) {
                const int q = oy / (a.stack + 1);
                const int ly = oy - q * (a.stack + 1);
                const int ph = q / sub;
                spw_cur = q - ph * sub;
                rowok = ly < a.stack && q < sub * sub;
                oy = ly * sub + ph;
            }
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = fmaf(acc[f][j][r], sc, bs[j * 4 + r]);
                    if (a.relu_out) x = fmaxf(x, 0.f);
                    v[j * 4 + r] = x;
                }
            store_frag(a.out, (size_t)(n * a.OH + oy) * a.OW * a.out_cs, a.out_cs, ox0 + fc * 16, a.OW, rowok && oy < a.OH, v, a.out_f32);
        }
        return;
    }
    // ---- fused max-pool (MaxPool2d(2,2) or MaxPool2d((2,1),(2,1))): the two rows of a pooling window are two fragments of
    // this wave (the launcher only picks tiles with MF % (2*fpr) == 0), the two columns are lanes l and l^1.  (sub == 1 here.)
    auto pooled = [&](auto fpr_c) {
        constexpr int FPR = decltype(fpr_c)::value;
        if constexpr (MF % (2 * FPR) == 0) {
            const int POH = a.OH >> 1, POW = a.pool_mode == 1 ? (a.OW >> 1) : a.OW;
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if ((f / FPR) & 1) continue;                 // odd tile rows are consumed by their even partner
                const int F = wm * MF + f;
                const int fr = F / FPR, fc = F - fr * FPR;
                const int oy = oy0 + fr, xb = ox0 + fc * 16, ox = xb + pl;
                float v0[16], v1[16], m[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x0 = fmaf(acc[f][j][r], sc, bs[j * 4 + r]), x1 = fmaf(acc[f + FPR][j][r], sc, bs[j * 4 + r]);
                        if (a.relu_out) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
                        v0[j * 4 + r] = x0;
                        v1[j * 4 + r] = x1;
                        float p = fmaxf(x0, x1);
                        if (a.pool_relu) p = fmaxf(p, 0.f);
                        m[j * 4 + r] = p;
                    }
                if (a.store_full) {
                    store_frag(a.out, (size_t)(n * a.OH + oy) * a.OW * a.out_cs, a.out_cs, xb, a.OW, oy < a.OH, v0, a.out_f32);
                    store_frag(a.out, (size_t)(n * a.OH + oy + 1) * a.OW * a.out_cs, a.out_cs, xb, a.OW, oy + 1 < a.OH, v1, a.out_f32);
                }
                const int py = oy >> 1;
                const size_t prow = (size_t)(n * POH + py) * POW * a.pool_cs;
                if (a.pool_mode == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)      // neighbour lane l^1 by DPP quad_perm [1,0,3,2]: one VALU op instead of a ds_bpermute
                        m[i] = fmaxf(m[i], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[i]), 0xB1, 0xF, 0xF, true)));
                    const int px = ox >> 1;
                    auto store_pooled = [&](const float (&mv)[16], int coff) {
                        u32x4 lo, hi;
                        pack_runs(mv, lo, hi);
                        uint16_t* op = (uint16_t*)a.pool_out + prow + (size_t)px * a.pool_cs + cout0 + coff;
                        if (fullw) {
                            // both lanes of a pair hold the pooled pixel: the even one stores run 0, the odd one run 1 -> whole lines
                            if (py < POH && px < POW) *(u32x4*)(op + (pl & 1) * 32) = (pl & 1) ? hi : lo;
                        } else if (!(pl & 1) && py < POH && px < POW) {
                            *(u32x4*)(op) = lo;
                            if (run1) *(u32x4*)(op + 32) = hi;
                        }
                    };
                    if (a.split_off) {
                        float mh[16], ml[16];
                        split_hi_lo(m, mh, ml);
                        store_pooled(mh, 0);
                        store_pooled(ml, a.split_off);
                    } else {
                        store_pooled(m, 0);
                    }
                } else {
                    store_frag(a.pool_out, prow, a.pool_cs, xb, POW, py < POH, m, false);
                }
            }
        }
    };
    if (fpr == 1) pooled(std::integral_constant<int, 1>{});
    else if (fpr == 2) pooled(std::integral_constant<int, 2>{});
    else pooled(std::integral_constant<int, 4>{});
}

// Epilogue of the fused conv1_1 + conv1_2 launch (16 x 16 tiles, 64 couts, bias + ReLU + MaxPool2d(2,2), only the pooled tensor is kept):
// the max over the 2x2 window is taken FIRST, on the raw accumulators -- x -> max(fma(x, sc, b), floor) is monotone for sc > 0, so the
// result is bit-identical to pooling the finished values (conv_epilogue) -- then ONE bias / ReLU / pack per pooled value instead of
// four, and each lane of a pixel pair finishes only the run it stores (even lane run 0, odd lane run 1: whole 128-byte lines).  The kernel
// carries none of the shared epilogue's variants: it was vector-issue bound with ~300 of its 1,170 VALU instructions per wave in here.
// (cout_base: first cout of the wave's 64 -- 0 in the fused conv1_2 launch, nt * BN + wn * 64 when the 128-cout trunk layers conv3_3 /
// conv4_3 take this path: 16 x 16 tiles, only the pooled tensor kept, see launch_dma's `lean` rule)
template <int EL, int MF>
__device__ __forceinline__ void conv_epilogue_pool2x2_lean(const ConvArgs& a, f32x4 (&acc)[MF][4], int n, int oy0, int ox0, int wm, int lane,
                                                           int cout_base = 0) {
    static_assert(MF % 2 == 0, "fragment f and f + 1 are the two rows of a pooling window");
    const int g = lane >> 4, pl = lane & 15, odd = pl & 1;
    const int cout0 = cout_base + g * 8;
    const float sc = a.acc_scale;
    const float floor_v = (a.relu_out || a.pool_relu) ? 0.f : -__builtin_inff();
    float bsel[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 b4 = *(const f32x4*)(a.bias + cout0 + odd * 32 + q * 4);
        bsel[q * 4 + 0] = b4[0]; bsel[q * 4 + 1] = b4[1]; bsel[q * 4 + 2] = b4[2]; bsel[q * 4 + 3] = b4[3];
    }
    const int POH = a.OH >> 1, POW = a.OW >> 1;
    const int px = (ox0 + pl) >> 1;
#pragma unroll
    for (int f = 0; f < MF; f += 2) {
        const int py = (oy0 + wm * MF + f) >> 1;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int i1 = 8 + i;
            float q0 = fmaxf(acc[f][i >> 2][i & 3], acc[f + 1][i >> 2][i & 3]);
            float q1 = fmaxf(acc[f][i1 >> 2][i1 & 3], acc[f + 1][i1 >> 2][i1 & 3]);
            q0 = fmaxf(q0, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q0), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true)));
            q1 = fmaxf(q1, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q1), 0xB1, 0xF, 0xF, true)));
            o[i] = fmaxf(fmaf(odd ? q1 : q0, sc, bsel[i]), floor_v);
        }
        const u32x4 pk = {El<EL>::pack2(o[0], o[1]), El<EL>::pack2(o[2], o[3]), El<EL>::pack2(o[4], o[5]), El<EL>::pack2(o[6], o[7])};
        uint16_t* op = (uint16_t*)a.pool_out + ((size_t)(n * POH + py) * POW + px) * a.pool_cs + cout0 + odd * 32;
        if (py < POH && px < POW) *(u32x4*)op = pk;
    }
}

// The plain trunk layer's epilogue -- bias (+ ReLU) -> 16-bit NHWC, all 64 couts of the wave stored -- and nothing else.  The shared
// epilogue above executes ~1,000 vector instructions per wave behind a k-loop whose co-resident workgroup keeps the SIMD's issue port half
// busy with MFMAs: 9-15 k cycles during which the workgroup's LDS and registers are held and its slot computes nothing (timing-only
// ablation, BBOCR_CONV_DBG=8: the pass is 20 % shorter without epilogues, conv2_1 33 %).  Here a value costs one add, half a convert and
// half an integer max: ReLU is applied to the PACKED pair (both element types are sign-magnitude, so max(int16, 0) == ReLU, and
// round-then-ReLU == ReLU-then-round bit for bit), bias is a plain add (acc_scale is 1 on this path), interior tiles skip the per-lane
// bounds tests.  WHOLE: regroup the two 32-byte runs of a pixel across lanes so that every store writes whole 128-byte lines (as above).
template <int EL, int MF, bool WHOLE>
__device__ __forceinline__ void conv_epilogue_plain_lean(const ConvArgs& a, f32x4 (&acc)[MF][4], int n, int nt, int oy0, int ox0, int wm, int wn,
                                                         int fpr, int lane, int BN) {
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    const int g = lane >> 4, pl = lane & 15;
    const int cout0 = nt * BN + wn * 64 + g * 8;
    float bs[16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 b4 = *(const f32x4*)(a.bias + cout0 + h * 32 + q * 4);
            bs[h * 8 + q * 4 + 0] = b4[0]; bs[h * 8 + q * 4 + 1] = b4[1]; bs[h * 8 + q * 4 + 2] = b4[2]; bs[h * 8 + q * 4 + 3] = b4[3];
        }
    const bool relu = a.relu_out != 0;                                               // wave-uniform
    const bool inside = oy0 + a.TH <= a.OH && ox0 + a.TW <= a.OW;                     // wave-uniform: a tile inside the image needs no per-lane tests
    const s16x2_t z2 = {0, 0};
    // element offset of this lane inside a fragment row: WHOLE -> pixel (pl & 7), run (pl >> 3); else pixel pl, run 0 (run 1 = + 32)
    const size_t lane_off = WHOLE ? (size_t)(pl & 7) * a.out_cs + cout0 + (pl >> 3) * 32 : (size_t)pl * a.out_cs + cout0;
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int F = wm * MF + f;
        const int fr = F / fpr, fc = F - fr * fpr;
        const int oy = oy0 + fr, xb = ox0 + fc * 16;
        unsigned int p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 2 * i;
            unsigned int q = El<EL>::pack2(acc[f][k >> 2][k & 3] + bs[k], acc[f][(k + 1) >> 2][(k + 1) & 3] + bs[k + 1]);
            if (relu) q = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, q), z2));
            p[i] = q;
        }
        uint16_t* op = (uint16_t*)a.out + ((size_t)(n * a.OH + oy) * a.OW + xb) * a.out_cs + lane_off;
        if constexpr (WHOLE) {
            u32x4 dA, dB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dA[i] = (unsigned)__builtin_amdgcn_update_dpp((int)p[i], (int)p[4 + i], 0x118 /*row_shr:8*/, 0xF, 0xC, false);
                dB[i] = (unsigned)__builtin_amdgcn_update_dpp((int)p[4 + i], (int)p[i], 0x108 /*row_shl:8*/, 0xF, 0x3, false);
            }
            if (inside) {
                *(u32x4*)op = dA;
                *(u32x4*)(op + (size_t)8 * a.out_cs) = dB;
            } else {
                const int xA = xb + (pl & 7);
                if (oy < a.OH && xA < a.OW) *(u32x4*)op = dA;
                if (oy < a.OH && xA + 8 < a.OW) *(u32x4*)(op + (size_t)8 * a.out_cs) = dB;
            }
        } else {
            const u32x4 lo = {p[0], p[1], p[2], p[3]}, hi = {p[4], p[5], p[6], p[7]};
            if (inside || (oy < a.OH && xb + pl < a.OW)) {
                *(u32x4*)op = lo;
                *(u32x4*)(op + 32) = hi;
            }
        }
    }
}

// Epilogue with a 1x1 conv 64 -> 64 behind the layer (a.post_w; 64-cout tiles whose wave holds all 64 channels of its pixels): the cout
// permutation makes a lane's finished values v[0..7] / v[8..15] exactly the B fragment (k = 8g + i) of channels 0-31 / 32-63 for the
// next MFMA, so z = W u costs two k-steps x four cout fragments per 16 pixels and u itself is never stored.  Same arithmetic as the
// separate 1x1 launch (u rounded to the element type, chunk 0 then chunk 1 accumulated in fp32): bit-identical z.
template <int EL, int MF>
__device__ __forceinline__ void conv_epilogue_post1x1(const ConvArgs& a, f32x4 (&acc)[MF][4], int n, int oy0, int ox0, int wm, int fpr, int lane) {
    const int g = lane >> 4, pl = lane & 15;
    const int cout0 = g * 8;
    const float sc = a.acc_scale;
    const float floor_v = a.relu_out ? 0.f : -__builtin_inff();
    float bs[16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 b4 = *(const f32x4*)(a.bias + cout0 + h * 32 + q * 4);
            bs[h * 8 + q * 4 + 0] = b4[0]; bs[h * 8 + q * 4 + 1] = b4[1]; bs[h * 8 + q * 4 + 2] = b4[2]; bs[h * 8 + q * 4 + 3] = b4[3];
        }
    typename El<EL>::v8 wf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[ks][j] = *(const typename El<EL>::v8*)(a.post_w + ((size_t)(ks * 4 + j) * 64 + lane) * 8);
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int F = wm * MF + f;
        const int fr = F / fpr, fc = F - fr * fpr;
        const int oy = oy0 + fr, xb = ox0 + fc * 16;
        u32x4 bk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = h * 8 + i * 2;
                const float x0 = fmaxf(fmaf(acc[f][k >> 2][k & 3], sc, bs[k]), floor_v), x1 = fmaxf(fmaf(acc[f][(k + 1) >> 2][(k + 1) & 3], sc, bs[k + 1]), floor_v);
                bk[h][i] = El<EL>::pack2(x0, x1);
            }
        f32x4 y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = El<EL>::mfma(wf[0][j], __builtin_bit_cast(typename El<EL>::v8, bk[0]), (f32x4){0.f, 0.f, 0.f, 0.f});
            y[j] = El<EL>::mfma(wf[1][j], __builtin_bit_cast(typename El<EL>::v8, bk[1]), y[j]);
        }
        // whole-line stores as in conv_epilogue (fullw): run 0 / run 1 of a pixel regrouped across lanes pl <-> pl ^ 8
        const u32x4 lo = {El<EL>::pack2(y[0][0], y[0][1]), El<EL>::pack2(y[0][2], y[0][3]), El<EL>::pack2(y[1][0], y[1][1]), El<EL>::pack2(y[1][2], y[1][3])};
        const u32x4 hi = {El<EL>::pack2(y[2][0], y[2][1]), El<EL>::pack2(y[2][2], y[2][3]), El<EL>::pack2(y[3][0], y[3][1]), El<EL>::pack2(y[3][2], y[3][3])};
        u32x4 dA, dB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dA[i] = (unsigned)__builtin_amdgcn_update_dpp((int)lo[i], (int)hi[i], 0x118 /*row_shr:8*/, 0xF, 0xC, false);
            dB[i] = (unsigned)__builtin_amdgcn_update_dpp((int)hi[i], (int)lo[i], 0x108 /*row_shl:8*/, 0xF, 0x3, false);
        }
        const int co = cout0 + (pl >> 3) * 32;
        const int xA = xb + (pl & 7), xB = xb + 8 + (pl & 7);
        uint16_t* op = (uint16_t*)a.out + (size_t)(n * a.OH + oy) * a.OW * a.out_cs + co;
        if (oy < a.OH && xA < a.OW) *(u32x4*)(op + (size_t)xA * a.out_cs) = dA;
        if (oy < a.OH && xB < a.OW) *(u32x4*)(op + (size_t)xB * a.out_cs) = dB;
    }
}

void pack_post1x1_weights(const float* w, uint16_t* out, int el) {
    size_t o = 0;
    for (int ks = 0; ks < 2; ++ks)
        for (int nf = 0; nf < 4; ++nf)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 15;
                const int cout = (nf >> 1) * 32 + (row >> 2) * 8 + (nf & 1) * 4 + (row & 3);       // the epilogue's run order (pack_conv_weights)
                for (int j = 0; j < 8; ++j) out[o++] = f32_to_el_host(el, w[(size_t)cout * 64 + ks * 32 + 8 * (l >> 4) + j]);
            }
}

// Generic register-staged variant (any kernel size / dilation / pooling): 256-thread workgroups built for TWO co-resident
// workgroups per CU (launch bound 2 waves/SIMD = 256 VGPRs) whose LDS-read and MFMA phases overlap each other.
template <int EL, int WM, int WN, int MF, int PITER>
__global__ void __launch_bounds__(WM * WN * 64, 2) conv_mfma_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN, NT = NW * 64, BN = WN * 64;
    constexpr int WBUF = BN * 64;          // bytes of one weight k-step slice (BN couts x 32 k x 2 B)
    constexpr int WPIECES = WBUF / 16;     // 16-B pieces per slice
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WRING = 2;               // weight k-step slices in flight
    unsigned char* const wbuf = smem;                  // [WRING][WBUF]
    unsigned char* const pbuf = smem + WRING * WBUF;   // [2][NP*64]
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
    // The patch of one chunk is staged in NPART parts so that at most PH x 4 VGPRs are in flight at once.
    constexpr int NPART = (PITER + 3) / 4, PH = PITER / NPART;   // PITER 4/8/16 -> 1/2/4 parts of 4 wave-instructions
    u32x4 pre[PH];
    // Loads only ISSUE here; masking of out-of-image slots and the ReLU-on-load are applied in store_part, right before the
    // ds_write, so nothing touches the loaded registers (and forces a vmcnt wait) while the loads are in flight.
    auto load_part = [&](int chunk, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        const int c = chunk * 32;
        const bool s0 = c < a.C0;
        const uint16_t* src = s0 ? a.in0 : a.in1;
        const int cs = s0 ? a.in0_cs : a.in1_cs;
        const int cb = (s0 ? c : c - a.C0) + l_kg * 8;
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int sp = src_pix[part * PH + i];
            u32x4 v = {0u, 0u, 0u, 0u};
            if (sp >= 0 && !CONV_DBG(a, 2)) v = *(const u32x4*)(src + (size_t)sp * cs + cb);   // dbg bit 2: timing-only ablation
            pre[i] = v;
        }
    };
    auto store_part = [&](int chunk, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        unsigned char* pb = pbuf + (chunk & 1) * patch_bytes;
        const bool relu = (chunk * 32 < a.C0) ? a.relu_in0 : a.relu_in1;
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int wi = wave + (part * PH + i) * NW;
            if (wi < n_wi) {
                u32x4 v = pre[i];
                if (relu) {
                    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    v = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
                }
                *(u32x4*)(pb + (size_t)(l_kg * a.NP + wi * 16 + l_pix) * 16) = v;
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    // staging schedule inside a chunk with >= 3 taps: part i of the NEXT chunk is loaded at tap i*S and stored at tap
    // i*S + S - 1, S = (ntaps-1)/NPART, so the last store lands at tap <= ntaps-2 (the pipelined schedule reads the next
    // chunk's first fragments during the last tap) and at most one part (PH x 4 VGPRs) is in flight.
    const int S = (a.ntaps - 1) / NPART;
    auto stage_patch_load = [&](int chunk, int tap) {
        if (tap == 0) load_part(chunk, P0{});
        if constexpr (NPART > 1) { if (tap == S) load_part(chunk, std::integral_constant<int, (NPART > 1 ? 1 : 0)>{}); }
        if constexpr (NPART > 2) { if (tap == 2 * S) load_part(chunk, std::integral_constant<int, (NPART > 2 ? 2 : 0)>{}); }
        if constexpr (NPART > 3) { if (tap == 3 * S) load_part(chunk, std::integral_constant<int, (NPART > 3 ? 3 : 0)>{}); }
    };
    auto stage_patch_store = [&](int chunk, int tap) {
        if (tap == S - 1) store_part(chunk, P0{});
        if constexpr (NPART > 1) { if (tap == 2 * S - 1) store_part(chunk, std::integral_constant<int, (NPART > 1 ? 1 : 0)>{}); }
        if constexpr (NPART > 2) { if (tap == 3 * S - 1) store_part(chunk, std::integral_constant<int, (NPART > 2 ? 2 : 0)>{}); }
        if constexpr (NPART > 3) { if (tap == 4 * S - 1) store_part(chunk, std::integral_constant<int, (NPART > 3 ? 3 : 0)>{}); }
    };
    auto stage_patch_now = [&](int chunk) {        // prologue: whole patch, synchronously
        load_part(chunk, P0{});
        store_part(chunk, P0{});
        if constexpr (NPART > 1) { load_part(chunk, std::integral_constant<int, (NPART > 1 ? 1 : 0)>{}); store_part(chunk, std::integral_constant<int, (NPART > 1 ? 1 : 0)>{}); }
        if constexpr (NPART > 2) { load_part(chunk, std::integral_constant<int, (NPART > 2 ? 2 : 0)>{}); store_part(chunk, std::integral_constant<int, (NPART > 2 ? 2 : 0)>{}); }
        if constexpr (NPART > 3) { load_part(chunk, std::integral_constant<int, (NPART > 3 ? 3 : 0)>{}); store_part(chunk, std::integral_constant<int, (NPART > 3 ? 3 : 0)>{}); }
    };
    // ---- weight slice: LDS-DMA, LDS image == global image (fragment order), lane-linear
    const unsigned char* wsrc = (const unsigned char*)a.wpk + (size_t)nt * nk * WBUF;
    auto issue_w = [&](int ks, int buf) {
        if (CONV_DBG(a, 1)) return;   // timing-only ablation (diagnostic build, BBOCR_CONV_DBG): no weight DMA, results are garbage
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

    {
        // 2-deep weight ring, fragments read at the top of every k-step
        issue_w(0, 0);
        stage_patch_now(0);
        __syncthreads();
        int ks = 0;
        for (int c = 0; c < a.nchunks; ++c) {
            const bool more = (c + 1 < a.nchunks);
            if (more && a.ntaps < 3) load_part(c + 1, P0{});
            const unsigned char* pbase = pbuf + (c & 1) * patch_bytes + lane_patch_off;
            int ky = 0, kx = 0;
            for (int tap = 0; tap < a.ntaps; ++tap, ++ks) {
                if (more && a.ntaps >= 3) stage_patch_load(c + 1, tap);
                if (ks + 1 < nk) issue_w(ks + 1, (ks + 1) & 1);
                const unsigned char* wb = wbuf + (ks & 1) * WBUF + lane_w_off;
                typename El<EL>::v8 af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j] = *(const typename El<EL>::v8*)(wb + j * 1024);
                const unsigned char* pb = pbase + ((ky * a.PW + kx) * a.dil) * 16;
                // all activation fragments are requested up front (MF ds_read_b128 in flight) so the LDS latency is paid
                // once per k-step; left to itself hipcc serialises read -> lgkmcnt(0) -> 4 MFMAs per fragment
                typename El<EL>::v8 bq[MF];
#pragma unroll
                for (int f = 0; f < MF; ++f) bq[f] = *(const typename El<EL>::v8*)(pb + frag_off[f]);
#pragma unroll
                for (int f = 0; f < MF; ++f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[f][j] = El<EL>::mfma(af[j], bq[f], acc[f][j]);
                }
                // pin the order: every fragment read first, then the MFMA stream behind counted lgkmcnt waits
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + MF, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * MF, 0);
                if (more) {
                    if (a.ntaps >= 3) stage_patch_store(c + 1, tap);
                    else if (tap == a.ntaps - 1) store_part(c + 1, P0{});
                }
                __syncthreads();
                if (++kx == a.KW) { kx = 0; ++ky; }
            }
        }
    }

    conv_epilogue<EL, MF>(a, acc, n, nt, oy0, ox0, wm, wn, fpr, lane, BN);
}

// ================================================================================================ 3x3, LDS-DMA staged
// Variant for 3x3 / pad 1 / dilation 1 layers (every big layer of CRAFT and the CRNN): BOTH operands reach LDS by LDS-DMA
// (global_load_lds_dwordx4), so the main loop holds no global->register loads at all, hipcc inserts no vmcnt waits of its
// own, and the schedule is fully static:
//   * activation patch: wave w copies channel group w of the NEXT chunk, one 64-pixel DMA per tap for the first NPB taps
//     (per-lane source address = the pixel's 16 bytes, or a zero page for padding); LDS image [group][NP = 64*NPB pixels];
//   * weights: slice ks+RING-1 is issued at k-step ks, first among the k-step's VMEM operations;
//   * every k-step ends with `s_waitcnt vmcnt(N)` + s_barrier where N (a compile-time constant per tap) is the number of
//     operations issued after slice ks+1, so DMAs ride across RING-2 barriers and nothing ever drains to vmcnt(0).
// vmcnt retires in order: a patch DMA issued at tap L has landed by the barrier of tap L+RING-1, hence the last one may be
// issued at tap 9-RING (NTAP-RING in general).  ReLU-on-load (BN-terminated VGG slices) is applied to the B fragments after the LDS read.
// FUSE1 (CRAFT conv1_2 only: Cin = 64, 16x16 tiles, BN = 64): the 64-channel input patch is not read from memory but PRODUCED
// in the prologue from the uint8 RGB page -- normalizeMeanVariance + conv1_1 (3x3, 3->64) + BN + ReLU, one 27-deep (padded
// to 32: a single k-step) MFMA product per 16 patch pixels -- and written straight into the two LDS patch buffers (both 32-channel chunks are
// resident from the start, the k-loop issues no patch DMA).  The 157 MB/page conv1_1 activation never exists in HBM.
// NF (cout fragments a wave multiplies, 4 or 2): layers with <= 32 real couts in a 64-cout tile (up4b, conv_cls.0/.2/.4) skip the two
// fragments that are pure padding (couts 32..63 of the tile, see the cout mapping of the epilogue) -- half the MFMAs, same results.
template <int EL, int WM, int WN, int MF, int NPB, int RING, int NPS = NPB * 64, bool FUSE1 = false, int NF = 4, int EPI = 0, int KS = 3>
__global__ void __launch_bounds__(WM * WN * 64, 2) conv3x3_dma_kernel(const ConvArgs a) {
    // KS = 3: 3x3 / padding 1 (nine taps per chunk); KS = 2: 2x2 / padding 0 (the CRNN's last conv: four taps per chunk)
    constexpr int NTAP = KS * KS, PAD = KS == 3 ? 1 : 0;
    static_assert((KS == 3 || KS == 2) && (KS == 3 || (!FUSE1 && EPI == 0)), "kernel size");
    constexpr int NW = WM * WN, NT = NW * 64, BN = WN * 64;
    static_assert(NW == 4, "one wave per 8-channel group of the patch");
    constexpr int WBUF = BN * 64, WPIECES = WBUF / 16, WPT = WPIECES / NT;
    static_assert(WPIECES % NT == 0 && (RING == 3 || RING == 4), "uniform DMA issue");
    // NPS: pixels per channel group in the LDS patch image; NPS < NPB*64 (exactly PH*PW) trims the patch to what the tile needs,
    // the last 64-pixel DMA block then runs with the lanes beyond NPS masked off
    constexpr int NP = NPS, PSLOTS = NTAP + 1 - RING;     // taps 0 .. NTAP-RING may issue patch DMAs
    constexpr int PPER = (NPB + PSLOTS - 1) / PSLOTS;     // patch DMA blocks per tap at most
    static_assert(NPS <= NPB * 64 && NPS > (NPB - 1) * 64 && NPS % 4 == 0, "patch size");
    static_assert(PSLOTS >= 1 && PPER * PSLOTS >= NPB, "patch does not fit the DMA schedule");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wbuf = smem;                    // [RING][WBUF]
    unsigned char* const pbuf = smem + RING * WBUF;      // [2][NP*64]
    constexpr int patch_bytes = NP * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int sub = a.sub;
    const int nk = a.nchunks * NTAP;
    // the fused conv1_2 kernel always runs 16 x 16 tiles (18-pixel patch rows): compile-time, so that every fragment read of the k-loop is
    // one base register + an immediate offset
    const int fpr = FUSE1 ? 1 : a.TW >> 4;
    const int PW = FUSE1 ? 18 : a.PW;

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    // dilation d (a.sub) with padding d == d*d independent plain 3x3 convs on the phase sub-lattices: image index n' enumerates
    // (n, phase_y, phase_x); tile coordinates are in sub-lattice units
    struct Geo {
        int n, nt, oy0, ox0, sph, spw;
        const unsigned char* wsrc;     // packed weight slices of this tile's cout tile
    };
    auto geom = [&](int id, Geo& g) {   // wave-uniform part
        g.nt = id % a.ntiles_n;
        id /= a.ntiles_n;
        const int tx = id % a.tiles_x;
        id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        int n = id / a.tiles_y;
        g.sph = 0; g.spw = 0;
        g.n = n;     // (sub > 1: all sub*sub phase images of page n are stacked along y inside the tile grid, see geom_pix)
        g.oy0 = ty * a.TH; g.ox0 = tx * a.TW;
        g.wsrc = (const unsigned char*)a.wpk + (size_t)g.nt * nk * WBUF;
    };
    // per-lane source pixel of each 64-pixel patch block (or -1: padding -> zero page)
    int spix[NPB];
    const int pw_magic = (65536 + a.PW - 1) / a.PW;     // pix / PW == (pix * magic) >> 16 exactly for pix < 448, PW <= 66
    const int stack_magic = ((1 << 20) + a.stack) / (a.stack + 1);   // vy / (stack+1) == (vy * magic) >> 20 (launch_conv checks the exactness bound)
    auto geom_pix = [&](const Geo& g) {
        const int iy0 = g.oy0 - PAD, ix0 = g.ox0 - PAD;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const int pix = pb * 64 + lane;
            const int py = (pix * pw_magic) >> 16, px = pix - py * a.PW;
            int ly = iy0 + py;
            const int lx = ix0 + px;                                // sub-lattice coordinates
            int sph = 0, spw = 0;
            bool rowok = ly >= 0;
            if (sub > 1) {
                // virtual row -> (phase q, row inside the phase image); row a.stack of every phase is the shared zero row
                const int vy = ly < 0 ? 0 : ly;
                const int q = (vy * stack_magic) >> 20;
                ly = vy - q * (a.stack + 1);
                sph = q / sub;
                spw = q - sph * sub;
                rowok = rowok && ly < a.stack && q < sub * sub;
            }
            const int iy = ly * sub + sph, ix = lx * sub + spw;
            spix[pb] = (py < a.PH && rowok && iy < a.H && lx >= 0 && ix < a.W) ? (g.n * a.H + iy) * a.W + ix : -1;
        }
    };
    auto stamp = [&](int i) {   // tile timeline: compiled in by -DBBOCR_DIAG only
#ifdef BBOCR_DIAG
        if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 4 + i] = __builtin_amdgcn_s_memtime();
#endif
    };
    stamp(0);
    Geo cur;
    geom(tile, cur);
    if constexpr (!FUSE1) geom_pix(cur);

    auto issue_p = [&](int sp, int chunk, int par, auto pb_c) {
        constexpr int pb = decltype(pb_c)::value;
        const int c = chunk * 32;
        const bool s0 = c < a.C0;
        const uint16_t* src = s0 ? a.in0 : a.in1;
        const int cs = s0 ? a.in0_cs : a.in1_cs;
        const int cb = (s0 ? c : c - a.C0) + wave * 8;
        const uint16_t* gp = sp >= 0 ? src + (size_t)sp * cs + cb : (const uint16_t*)a.zero;
        if ((pb + 1) * 64 <= NPS || pb * 64 + lane < NPS)     // (still one vmcnt event per wave: every wave has lanes below NPS)
            lds_dma16(gp, pbuf + par * patch_bytes + (wave * NP + pb * 64) * 16);
    };
    auto issue_w = [&](const unsigned char* slice, int slot) {
#pragma unroll
        for (int p0 = 0; p0 < WPIECES; p0 += NT) lds_dma16(slice + (size_t)(p0 + tid) * 16, wbuf + slot * WBUF + (p0 + wave * 64) * 16);
    };
    int frag_off[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int F = wm * MF + f;
        const int fr = F / fpr, fc = F - fr * fpr;
        frag_off[f] = (fr * PW + fc * 16) * 16;
    }
    const int lane_patch_off = ((lane >> 4) * NP + (lane & 15)) * 16;
    const int lane_w_off = wn * 4 * 1024 + lane * 16;

    // patch DMA schedule inside a chunk: PCNT(tap) blocks at tap (first taps take two while NPB > PSLOTS)
    struct Sched {
        static constexpr int pcnt(int tap) {       // the NPB blocks dealt over the first PSLOTS taps, the early taps take the larger shares
            if (FUSE1 || tap >= PSLOTS) return 0;
            const int base = NPB / PSLOTS, extra = NPB % PSLOTS;
            return base + (tap < extra ? 1 : 0);
        }
        static constexpr int pfirst(int tap) { int s = 0; for (int t = 0; t < tap; ++t) s += pcnt(t); return s; }
        static constexpr int wcnt(bool more, int tap) { return (more || tap < NTAP - (RING - 1)) ? WPT : 0; }
        // operations younger than slice ks+1 at the end of tap `tap` of a chunk of kind `more` (previous chunk: kind true)
        static constexpr int younger(bool more, int tap) {
            int nv = wcnt(more, tap) + (more ? pcnt(tap) : 0);
            for (int back = 1; back <= RING - 2; ++back) {
                const int t = tap - back;
                const bool m = t >= 0 ? more : true;
                const int tt = t >= 0 ? t : t + NTAP;
                const int pc = m ? pcnt(tt) : 0;
                nv += (back == RING - 2) ? pc : (wcnt(m, tt) + pc);   // the oldest k-step of the window: only what followed its slice
            }
            return nv;
        }
    };
    static_assert(FUSE1 || Sched::pfirst(NTAP) == NPB, "patch DMA schedule must cover the patch");

    // prologue: first RING-1 weight slices + the whole first patch of the first tile
#pragma unroll
    for (int i = 0; i < RING - 1; ++i)
        if (i < nk) issue_w(cur.wsrc + (size_t)i * WBUF, i);
    if constexpr (!FUSE1) {
        [&]<int... PB>(std::integer_sequence<int, PB...>) { (issue_p(spix[PB], 0, 0, std::integral_constant<int, PB>{}), ...); }(std::make_integer_sequence<int, NPB>{});
    } else {
        static_assert(!FUSE1 || (RING == 3 && BN == 64 && NPS == 324), "fused conv1_1 producer: 16x16 tiles, 3-deep ring of 4 KB slices");
        // (1) normalised RGB patch 20 x 20 (tile + 2-pixel halo) as 4 x bf16 per pixel into ring slot 2, which receives its first
        //     weight slice only at k-step 0.  Canvas semantics of detector_input: pixels of the H x W canvas beyond the page are
        //     raw zeros (normalised like any pixel), pixels beyond the canvas are conv1_1's zero padding.
        u32x2* const rgbp = (u32x2*)(wbuf + 2 * WBUF);
        const uint8_t* rgb = (const uint8_t*)a.in0;
        const float m0 = 0.485f * 255.0f, m1 = 0.456f * 255.0f, m2 = 0.406f * 255.0f;
        const float s0 = 0.229f * 255.0f, s1 = 0.224f * 255.0f, s2 = 0.225f * 255.0f;
        // Every global load of the prologue is ISSUED before the first one is consumed (the page bytes of both pixels a thread owns, the
        // conv1_1 fragments, the bias): one round trip instead of four dependent ones -- the per-tile timeline showed this prologue at
        // 13.7 k cycles against 9.9 k for the 18 k-steps it feeds, on a kernel whose workgroups are latency-bound.
        static_assert(NT == 256, "two patch pixels per thread");
        int pq[2];
        bool on_canvas[2];
        unsigned int cr[2], cg[2], cb[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int p = tid + it * NT;
            const int py = p / 20, px = p - py * 20;
            const int iy = cur.oy0 - 2 + py, ix = cur.ox0 - 2 + px;
            pq[it] = p;
            on_canvas[it] = p < 400 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const bool on_page = on_canvas[it] && iy < a.rgb_H && ix < a.rgb_W;
            // off the page: any valid address (pixel 0 of the image), the value is replaced by the canvas' raw zero below
            const uint8_t* q = rgb + (on_page ? ((size_t)(cur.n * a.rgb_H + iy) * a.rgb_W + ix) * 3 : (size_t)cur.n * a.rgb_H * a.rgb_W * 3);
            cr[it] = q[0]; cg[it] = q[1]; cb[it] = q[2];
            if (!on_page) { cr[it] = 0u; cg[it] = 0u; cb[it] = 0u; }
        }
        // conv1_1 weights as MFMA A fragments: one 32-deep k-step (layout: pack_conv1_1_weights_fused), couts in the run order of the
        // epilogue mapping, so that a lane's 16 outputs are exactly one 16-byte slot of each 32-channel chunk of the patch image
        typename El<EL>::v8 w1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w1[j] = *(const typename El<EL>::v8*)(a.c11_w + ((size_t)j * 64 + lane) * 8);
        const int g = lane >> 4, pl = lane & 15;
        f32x4 b1[4];                                    // conv1_1's bias in accumulator layout: the first MFMA of a fragment takes it as C
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) b1[h * 2 + q4] = *(const f32x4*)(a.c11_b + h * 32 + g * 8 + q4 * 4);
        __builtin_amdgcn_sched_barrier(0);              // loads above, their consumers below
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (pq[it] < 400) {
                u32x2 v = {0u, 0u};
                if (on_canvas[it]) {
                    v[0] = El<EL>::pack2(((float)cr[it] - m0) / s0, ((float)cg[it] - m1) / s1);
                    v[1] = El<EL>::pack2(((float)cb[it] - m2) / s2, 0.f);
                }
                rgbp[pq[it]] = v;
            }
        }
        __syncthreads();
        // (2) 21 fragments of 16 patch pixels (18 x 18 = 324 = 20*16 + 4), dealt round-robin to the four waves.  A fragment is a
        //     latency chain (LDS reads -> 4 MFMAs -> ReLU / pack -> LDS write, ~1,300 cycles on its own -- the tile timeline
        //     showed 7.8 k cycles for the 5-6 fragments of a wave); the fragments of a wave are independent, so they run as TWO
        //     interleaved batches of three: all B operands of a batch are read first, then its 12 MFMAs, then the three epilogues.
        auto frag_geo = [&](int fi, int& pp, int& py, int& px) {
            pp = fi * 16 + pl;
            py = pp / 18;
            px = pp - py * 18;                                   // patch pixel -> its 3x3 window starts at (py, px) of the RGB patch
        };
#pragma unroll
        for (int b3 = 0; b3 < 2; ++b3) {
            typename El<EL>::v8 bfr[3];
            int pps[3], pys[3], pxs[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int fi = wave + (b3 * 3 + q) * NW;         // fi >= 21: nothing is stored, the operands only need valid addresses
                frag_geo(fi < 21 ? fi : 20, pps[q], pys[q], pxs[q]);
                const int pc = pps[q] < 324 ? pys[q] * 20 + pxs[q] : 0;
                // K = 32 in one k-step: taps 2g, 2g+1 as [r g b X | r g b Y], the ninth tap's r / g / b in the pad slots X(g=0) / Y(g=0) / X(g=1)
                // (pack_conv1_1_weights_fused); a stored pixel is {r | g << 16, b | 0 << 16}
                const int t0 = 2 * g, t1 = t0 + 1;
                const u32x2 a0 = rgbp[pc + (t0 / 3) * 20 + (t0 % 3)];
                const u32x2 a1 = rgbp[pc + (t1 / 3) * 20 + (t1 % 3)];
                const u32x2 t8 = rgbp[pc + 2 * 20 + 2];
                const unsigned int x = g == 0 ? (t8[0] << 16) : (g == 1 ? (t8[1] << 16) : 0u);
                const unsigned int y = g == 0 ? (t8[0] & 0xffff0000u) : 0u;
                bfr[q] = __builtin_bit_cast(typename El<EL>::v8, (u32x4){a0[0], a0[1] | x, a1[0], a1[1] | y});
            }
            f32x4 d[3][4];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[q][j] = El<EL>::mfma(w1[j], bfr[q], b1[j]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int fi = wave + (b3 * 3 + q) * NW;
                const int pp = pps[q];
                const int iy = cur.oy0 - 1 + pys[q], ix = cur.ox0 - 1 + pxs[q];
                const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;     // else: conv1_2's own zero padding
                const float cap = inside ? __builtin_inff() : 0.f;                  // ReLU and the padding mask as one v_med3: med3(x, 0, cap)
                if (fi < 21 && pp < 324) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int k = h * 8 + i * 2;
                            const float x0 = __builtin_amdgcn_fmed3f(d[q][k >> 2][k & 3], 0.f, cap), x1 = __builtin_amdgcn_fmed3f(d[q][(k + 1) >> 2][(k + 1) & 3], 0.f, cap);
                            o[i] = El<EL>::pack2(x0, x1);
                        }
                        *(u32x4*)(pbuf + h * patch_bytes + (g * NP + pp) * 16) = o;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(1);

    int wslot = 0, par = 0;
    const unsigned char* wp = cur.wsrc + (size_t)(RING - 1) * WBUF;   // next weight slice to issue
    f32x4 acc[MF][4];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // One k-step.  MORE: a chunk follows (its patch is prefetched, weight slices keep streaming).
    // Fragment schedule (pinned with sched_barrier: left to itself hipcc keeps two B fragments live and puts a full `s_waitcnt lgkmcnt(0)`
    // LDS round trip in front of every group of 8 MFMAs -- four exposed round trips per k-step, ~500 of its ~1,250 cycles):
    //   [A(t) x NF][DMA issue] | MFMA group 0 | read B 4,5 | group 1 | read B 6,7 | group 2 | read B'0,1 | group 3 | read B'2,3 | barrier
    // group g = fragments 2g, 2g+1 x all couts; B' = the NEXT tap's first four fragments: the patch of a chunk is stable across its nine
    // taps, so they are read ahead of the barrier (only the weight slice needs it).  Every read has a group (8 MFMAs, 128 cycles) to land.
    // Tap 0 of a chunk reads its first four fragments itself (new patch buffer, ReLU pass).
    static_assert(MF % 2 == 0 && MF >= 4, "B fragments are scheduled in pairs, four ahead");
    typename El<EL>::v8 pre[4];
    auto step = [&](auto tap_c, auto more_c, int c) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr bool MORE = decltype(more_c)::value;
        constexpr int ky = tap / KS, kx = tap % KS;
        const unsigned char* wb = wbuf + wslot * WBUF + lane_w_off;
        const unsigned char* pb = pbuf + par * patch_bytes + lane_patch_off + (ky * PW + kx) * 16;
        typename El<EL>::v8 af[NF], bq[MF];
#pragma unroll
        for (int j = 0; j < NF; ++j) af[j] = *(const typename El<EL>::v8*)(wb + j * 1024);
        if constexpr (tap == 0) {
#pragma unroll
            for (int f = 0; f < 4; ++f) bq[f] = *(const typename El<EL>::v8*)(pb + frag_off[f]);
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f) bq[f] = pre[f];
        }
        // (dbg bits 16 / 32: timing-only ablations of a diagnostic build -- the k-loop without its weight / patch DMA, results are garbage)
        if constexpr (Sched::wcnt(MORE, tap) > 0) {
            int slot = wslot + RING - 1;
            if (slot >= RING) slot -= RING;
            if (!CONV_DBG(a, 16)) issue_w(wp, slot);
            wp += WBUF;
        }
        if constexpr (MORE && Sched::pcnt(tap) > 0) {
            constexpr int p0 = Sched::pfirst(tap);
            const int nc = c + 1;
            if (!CONV_DBG(a, 32))
                [&]<int... I>(std::integer_sequence<int, I...>) { (issue_p(spix[p0 + I], nc, par ^ 1, std::integral_constant<int, p0 + I>{}), ...); }(
                    std::make_integer_sequence<int, Sched::pcnt(tap)>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int kyn = (tap + 1) / KS, kxn = (tap + 1) % KS;
        const unsigned char* pbn = pbuf + par * patch_bytes + lane_patch_off + (kyn * PW + kxn) * 16;
#pragma unroll
        for (int g = 0; g < MF / 2; ++g) {
#pragma unroll
            for (int f = 2 * g; f < 2 * g + 2; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[f][j] = El<EL>::mfma(af[j], bq[f], acc[f][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 2 * g + 4; f < 2 * g + 6; ++f) {
                if (f < MF) bq[f] = *(const typename El<EL>::v8*)(pb + frag_off[f]);
                else if constexpr (tap < NTAP - 1) pre[f - MF] = *(const typename El<EL>::v8*)(pbn + frag_off[f - MF]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (no lgkmcnt here: this step's weight fragments were consumed by its MFMAs, the reads still in flight are the next tap's B
        // fragments from the patch buffer nobody writes during this chunk)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(Sched::younger(MORE, tap)) : "memory");
        wslot = wslot + 1 == RING ? 0 : wslot + 1;
    };
    // ReLU-on-load (the input is a BatchNorm-terminated VGG slice, conv3_3 / conv4_3): applied ONCE per chunk, in place in the LDS patch,
    // when the chunk begins (its DMA has landed by the previous k-step's barrier) -- 6 x (ds_read, 4 integer max, ds_write) per thread and
    // chunk.  Applied to the B fragments of every tap instead it was 32 extra VALU instructions per k-step on top of ~50, in a loop whose
    // SIMD issue port is ~90 % busy: conv3_3 / conv4_3 ran 9 % below their un-ReLU'd twins (PMC instruction counts).
    auto relu_patch = [&](int c) {
        if (!((c * 32 < a.C0) ? a.relu_in0 : a.relu_in1)) return;        // wave-uniform
        const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        s16x8* pp = (s16x8*)(pbuf + par * patch_bytes);
#pragma unroll
        for (int i = 0; i < (NP * 4 + NT - 1) / NT; ++i) {
            const int k = tid + i * NT;
            if ((i + 1) * NT <= NP * 4 || k < NP * 4) pp[k] = __builtin_elementwise_max(pp[k], z);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto chunk = [&](auto more_c, int c) {
        relu_patch(c);
        [&]<int... T>(std::integer_sequence<int, T...>) { (step(std::integral_constant<int, T>{}, more_c, c), ...); }(std::make_integer_sequence<int, NTAP>{});
        par ^= 1;
    };
    for (int c = 0; c + 1 < a.nchunks; ++c) chunk(std::true_type{}, c);
    chunk(std::false_type{}, a.nchunks - 1);
    stamp(2);
#ifdef BBOCR_DIAG
    if (CONV_DBG(a, 8)) {   // timing-only ablation, no epilogue
        if (acc[0][0][0] == 123.456f) *(float*)a.out = acc[MF - 1][3][3];
        return;
    }
#endif
    if constexpr (FUSE1) conv_epilogue_pool2x2_lean<EL, MF>(a, acc, cur.n, cur.oy0, cur.ox0, wm, lane);       // (launch_dma checks its preconditions)
    else if constexpr (EPI == 1) conv_epilogue_post1x1<EL, MF>(a, acc, cur.n, cur.oy0, cur.ox0, wm, fpr, lane);                  // (launch_dma checks its preconditions)
    else if constexpr (KS == 3 && NF == 4) {
        // a.lean (set by launch_dma_one when the layer is a plain one, wave-uniform): 1 / 2 = bias (+ ReLU) -> 16-bit stores, whole lines / half lines;
        // 3 = the pooled-only epilogue of the fused conv1_2 launch on this layer's couts; 0 = the shared epilogue with all its variants
        if (a.lean == 1) conv_epilogue_plain_lean<EL, MF, true>(a, acc, cur.n, cur.nt, cur.oy0, cur.ox0, wm, wn, fpr, lane, BN);
        else if (a.lean == 2) conv_epilogue_plain_lean<EL, MF, false>(a, acc, cur.n, cur.nt, cur.oy0, cur.ox0, wm, wn, fpr, lane, BN);
        else if (a.lean == 3) conv_epilogue_pool2x2_lean<EL, MF>(a, acc, cur.n, cur.oy0, cur.ox0, wm, lane, cur.nt * BN + wn * 64);
        else conv_epilogue<EL, MF>(a, acc, cur.n, cur.nt, cur.oy0, cur.ox0, wm, wn, fpr, lane, BN, sub, cur.sph, cur.spw);
    } else conv_epilogue<EL, MF>(a, acc, cur.n, cur.nt, cur.oy0, cur.ox0, wm, wn, fpr, lane, BN, sub, cur.sph, cur.spw);
#ifdef BBOCR_DIAG
    if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3); }
#endif
}

// ================================================================================================ 3x3, resident weights
// The half-resolution tail of CRAFT (upconv4.3x3 64->32, conv_cls.0/.2 32->32, conv_cls.4 32->16 + classifier tail) has <= 32 real
// couts and 1-2 input chunks: its tiles are 9-18 short k-steps, and in conv3x3_dma_kernel every 16x16 tile re-fetches ALL of the
// layer's weights through LDS-DMA (36-72 KB of weight slices against a 20-41 KB activation patch) behind one barrier per k-step --
// those launches ran at the LDS-DMA fill rate, not at the HBM rate (3.5 TB/s of algorithmic bytes).  Here a PERSISTENT workgroup
// loads the NF = 2 real cout fragments of every (chunk, tap) slice ONCE (2 KB each, 18-36 KB in all) and then walks tiles
// tile0 + i * gridDim.x: per tile only the 18x18 activation patch is DMA'd (the next tile's while this one is multiplied when DB),
// the k-loop reads weights and patch from LDS without a single barrier, and the shared epilogue stores.  Two barriers per tile.
template <int EL, int NCH, bool DB, int NF = 2, bool LEAN = false>
__global__ void __launch_bounds__(256, 2) conv3x3_resw_kernel(const ConvArgs a) {
    // NF = 2: the <= 32-cout layers (fragments 0 and 1 of every slice are resident); NF = 4, LEAN: a 64-cout layer with one input chunk whose
    // 2x2 max-pool is fused and only the pooled tensor kept (the CRNN's 32 -> 64 layer: nine 4 KB slices resident, conv_epilogue_pool2x2_lean)
    static_assert(NF == 2 || (NF == 4 && NCH == 1), "resident weights");
    constexpr int MF = 4, NP = 324, NPB = 6, NK = NCH * 9;
    constexpr int WSL = NF * 1024;                       // resident bytes per (chunk, tap) slice
    constexpr int PCH = NP * 64;                         // patch bytes per chunk
    constexpr int NBUF = DB ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wres = smem;                    // [NK][WSL]
    unsigned char* const pbuf = smem + NK * WSL;         // [NBUF][NCH][PCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    // resident weights: NK slices x 2 KB, 4 KB per workgroup-wide DMA instruction (two slices)
    {
        const unsigned char* wsrc = (const unsigned char*)a.wpk;
        if constexpr (NF == 4) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)              // a whole 4 KB slice per workgroup-wide DMA instruction
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)ks * 4096 + (size_t)tid * 16),
                                                 (__attribute__((address_space(3))) void*)(wres + ks * WSL + wave * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < (NK + 1) / 2; ++k2) {
                const int ks = k2 * 2 + (tid >> 7);          // threads 0..127 -> slice 2*k2, 128..255 -> slice 2*k2 + 1
                if (ks < NK)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)ks * 4096 + (size_t)(tid & 127) * 16),
                                                     (__attribute__((address_space(3))) void*)(wres + (k2 * 2 + (wave >> 1)) * WSL + (wave & 1) * 1024), 16, 0, 0);
            }
        }
    }
    const int pw_magic = (65536 + 18 - 1) / 18;
    auto issue_patch = [&](int tile, int buf) {
        int id = tile;
        const int tx = id % a.tiles_x;
        id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int n = id / a.tiles_y;
        const int iy0 = ty * 16 - 1, ix0 = tx * 16 - 1;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const int pix = pb * 64 + lane;
            const int py = (pix * pw_magic) >> 16, px = pix - py * 18;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = py < 18 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const size_t sp = ((size_t)(n * a.H + iy) * a.W + ix) * a.in0_cs + wave * 8;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint16_t* gp = ok ? a.in0 + sp + c * 32 : (const uint16_t*)a.zero;
                if ((pb + 1) * 64 <= NP || pix < NP)
                    // (the builtin, not lds_dma16: with counted lgkmcnt waits hipcc's schedule of this unrolled tile body measured 10 % slower)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                     (__attribute__((address_space(3))) void*)(pbuf + (buf * NCH + c) * PCH + (wave * NP + pb * 64) * 16), 16, 0, 0);
            }
        }
    };
    int frag_off[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) frag_off[f] = ((wm * MF + f) * 18) * 16;      // fragment F = tile row F (fpr = 1)
    const int lane_patch_off = ((lane >> 4) * NP + (lane & 15)) * 16;
    const int lane_w_off = lane * 16;

    int tile = xcd_remap(blockIdx.x, gridDim.x);         // the tiles in flight on one XCD are neighbours: their halos meet in one L2
    if (tile < ntiles) issue_patch(tile, 0);
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if constexpr (DB) {
            // every wave has finished reading buffer buf^1 (tile - gridDim.x): the next tile's patch may land there
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (next < ntiles) {
                issue_patch(next, buf ^ 1);
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPB * NCH) : "memory");     // vmcnt retires in order: this tile's patch (and the weights) have landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        f32x4 acc[MF][4];
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const unsigned char* pc = pbuf + (buf * NCH + c) * PCH + lane_patch_off;
            const bool relu = a.relu_in0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const unsigned char* wb = wres + (c * 9 + tap) * WSL + lane_w_off;
                const unsigned char* pb = pc + ((tap / 3) * 18 + (tap % 3)) * 16;
                typename El<EL>::v8 af[NF], bq[MF];
#pragma unroll
                for (int j = 0; j < NF; ++j) af[j] = *(const typename El<EL>::v8*)(wb + j * 1024);
#pragma unroll
                for (int f = 0; f < MF; ++f) bq[f] = *(const typename El<EL>::v8*)(pb + frag_off[f]);
                if (relu) {
                    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int f = 0; f < MF; ++f) bq[f] = __builtin_bit_cast(typename El<EL>::v8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, bq[f]), z));
                }
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[f][j] = El<EL>::mfma(af[j], bq[f], acc[f][j]);
            }
        }
        int id = tile;
        const int tx = id % a.tiles_x;
        id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int n = id / a.tiles_y;
        if constexpr (!DB) {
            // single patch buffer: everybody is done reading it, the next tile's patch travels while this tile's epilogue stores
            // (lgkmcnt(0): the compiler may sink the last MFMAs and their LDS wait below a bare barrier -- see conv3x3_up4_kernel)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (next < ntiles) issue_patch(next, 0);
        }
        if constexpr (LEAN) conv_epilogue_pool2x2_lean<EL, MF>(a, acc, n, ty * 16, tx * 16, wm, lane);
        else conv_epilogue<EL, MF>(a, acc, n, 0, ty * 16, tx * 16, wm, 0, 1, lane, 64);
        if constexpr (DB) buf ^= 1;
    }
}

template <int EL>
static hipError_t launch_resw(const ConvArgs& a, hipStream_t s) {
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    static const int db_knob = diag_knob("BBOCR_RESW_DB", 1);
    const int ncu = device_cus();
    auto go = [&](auto kern, size_t smem, int per_cu, LdsOptIn& attr) -> hipError_t {
        if (hipError_t e = lds_opt_in(attr, (const void*)kern, smem); e != hipSuccess) return e;
        const int grid = ntiles < ncu * per_cu ? ntiles : ncu * per_cu;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, a);
        return hipGetLastError();
    };
    constexpr size_t P = 324 * 64, W1 = 9 * 2048, W2 = 18 * 2048;
    static LdsOptIn at[5];
    if (a.cout_store == 64) return go(conv3x3_resw_kernel<EL, 1, true, 4, true>, 9 * 4096 + 2 * P, 2, at[4]);       // 78.3 KB: two workgroups per CU
    if (a.nchunks == 1) {
        if (db_knob) return go(conv3x3_resw_kernel<EL, 1, true>, W1 + 2 * P, 2, at[0]);          // 59.9 KB: two workgroups per CU
        return go(conv3x3_resw_kernel<EL, 1, false>, W1 + P, 4, at[1]);                           // 39.2 KB: four
    }
    if (db_knob >= 2) return go(conv3x3_resw_kernel<EL, 2, true>, W2 + 4 * P, 1, at[2]);           // 119.8 KB: one
    return go(conv3x3_resw_kernel<EL, 2, false>, W2 + 2 * P, 2, at[3]);                            // 78.3 KB: two
}

// ================================================================================================ CRAFT upconv4 in one launch
// upconv4 = [1x1 (cat[up(y), s1]) 192 -> 64 + BN + ReLU] -> [3x3 64 -> 32 + BN + ReLU] at the half-resolution grid (480 x 640 for a 1280x960
// page).  As two launches (conv1x1_dma<ADDUP> + the resident 3x3) the 64-channel intermediate u4a makes a round trip through HBM
// (39 MB written + 50 MB read per page) and the 1x1 moves 128 MB per page for 2.5 GFLOP.  Here the resident-weights 3x3 PRODUCES its own
// 18x18x64 input patch in LDS, the way conv1_2 produces conv1_1: per 16 patch pixels a wave loads the s1 pixels straight into MFMA B
// fragments (4 x 16 B per lane: 128 channels), multiplies them with the 1x1 weights held in REGISTERS (16 fragments = 64 VGPRs, loaded
// once per persistent workgroup), adds bias and the 2x bilinear up-sampling of z (8 gathered 16-byte loads), applies ReLU, rounds to the
// element type and writes the two 32-channel chunk slots of the pixel -- bit for bit what the 1x1 launch stored.  The loads of fragment
// i + 1 are in flight while fragment i is multiplied and blended.  u4a never exists in HBM.
template <int EL>
__global__ void __launch_bounds__(256, 2) conv3x3_up4_kernel(const ConvArgs a) {
    constexpr int MF = 4, NF = 2, NP = 324, NK = 18, WSL = NF * 1024, PCH = NP * 64, NFRAG = 21;
    typedef typename El<EL>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wres = smem;                    // [18][2 KB]   upconv4.conv.3, fragments 0 and 1 of every (chunk, tap) slice
    unsigned char* const pbuf = smem + NK * WSL;         // [2 chunks][4 groups][324 px] x 16 B   u4a patch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pl = lane & 15;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    {
        const unsigned char* wsrc = (const unsigned char*)a.wpk;
#pragma unroll
        for (int k2 = 0; k2 < NK / 2; ++k2) {
            const int ks = k2 * 2 + (wave >> 1);
            lds_dma16(wsrc + (size_t)ks * 4096 + (size_t)(tid & 127) * 16, wres + ks * WSL + (wave & 1) * 1024);
        }
    }
    // 1x1 weights: packed slice c (32 input channels) x fragment j (16 couts in the epilogue's run order) -> registers for the whole launch
    v8 w1[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) w1[c][j] = *(const v8*)((const unsigned char*)a.aux_w + (size_t)c * 4096 + j * 1024 + lane * 16);
    // bias of the 1x1 in the accumulator order of a lane (run h, position i): 16 floats per lane group, kept in LDS (registers are what
    // this kernel is short of: 64 hold the 1x1 weights) and added after the MFMAs in the two-launch path's order ((acc + bias) + up(z))
    float* const b1s = (float*)(pbuf + 2 * PCH);         // [4 groups][16]
    if (tid < 64) b1s[tid] = a.aux_b[((tid & 15) >> 3) * 32 + (tid >> 4) * 8 + (tid & 7)];
    __syncthreads();
    const int LH = a.up_H >> 1, LW = a.up_W >> 1;

    struct Geo { int pp, cy, cx; bool inside; };
    auto geo = [&](int oy0, int ox0, int fi) {
        Geo G;
        G.pp = fi * 16 + pl;
        const int ppc = G.pp < NP ? G.pp : NP - 1;
        const int py = ppc / 18, px = ppc - py * 18;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        G.inside = G.pp < NP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        G.cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);       // any valid address: masked at the store
        G.cx = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
        return G;
    };
    // s1 pixel -> the four B fragments of the 1x1 (128 channels), straight from memory
    auto issue_s = [&](int n, const Geo& G, v8 (&sv)[4]) {
        const uint16_t* sp = a.in0 + ((size_t)(n * a.H + G.cy) * a.W + G.cx) * a.in0_cs + g * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) sv[c] = *(const v8*)(sp + c * 32);
    };
    // one fragment: gather z (issued first: it has the MFMA stream to arrive), then the NEXT fragment's s1 loads, then 16 MFMAs, blend, store
    auto produce = [&](int n, const Geo& G, const v8 (&sv)[4], bool more, const Geo& Gn, v8 (&sn)[4]) {
        // F.interpolate(scale_factor 2, bilinear, align_corners=False) of z at (cy, cx): the arithmetic of conv_epilogue<ADDUP>
        float sy = ((float)G.cy + 0.5f) * 0.5f - 0.5f, sx = ((float)G.cx + 0.5f) * 0.5f - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < LH - 1 ? 1 : 0), x1 = x0 + (x0 < LW - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        // The four blend weights are FINISHED here, before any load of this fragment is issued, and pinned in registers (conv_epilogue<ADDUP>
        // computes its weights in issue() as well).  Left to itself hipcc sinks their computation into the `pp < NP` block below, i.e.
        // directly behind the 16-MFMA chain, whose intermediate accumulators it renames (dst != srcC) and whose temporaries it then reuses
        // for the weights: packed-fp32 VALU writes to registers that MFMAs still in flight read as srcC.  The hazard recogniser's wait
        // states did not cover it (two waves share the matrix pipe, a dependent 8-pass MFMA starts late): thousands of u4b values per launch
        // differed from the two-launch path, differently on every run.  Round 4, tools/micro/up4_check.hip: pinned 0 of 2.4 M values differ
        // in 16 runs; unpinned 7,000-40,000 per run (also with nops or a full vmcnt wait around the chain).
        float a00 = hy * hx, a01 = hy * lx, a10 = ly * hx, a11 = ly * lx;
        asm volatile("" : "+v"(a00), "+v"(a01), "+v"(a10), "+v"(a11));
        const f32x2_t w00 = {a00, a00}, w01 = {a01, a01}, w10 = {a10, a10}, w11 = {a11, a11};
        const uint16_t* zb = a.addup + (size_t)n * LH * LW * a.up_cs + g * 8;
        const uint16_t* z00 = zb + ((size_t)y0 * LW + x0) * a.up_cs;
        const uint16_t* z01 = zb + ((size_t)y0 * LW + x1) * a.up_cs;
        const uint16_t* z10 = zb + ((size_t)y1 * LW + x0) * a.up_cs;
        const uint16_t* z11 = zb + ((size_t)y1 * LW + x1) * a.up_cs;
        u32x4 q[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            q[h * 4 + 0] = *(const u32x4*)(z00 + h * 32); q[h * 4 + 1] = *(const u32x4*)(z01 + h * 32);
            q[h * 4 + 2] = *(const u32x4*)(z10 + h * 32); q[h * 4 + 3] = *(const u32x4*)(z11 + h * 32);
        }
        if (more) issue_s(n, Gn, sn);
        __builtin_amdgcn_sched_barrier(0);                   // loads first (vmcnt retires in order: z of this fragment before s1 of the next)
        f32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = El<EL>::mfma(w1[c][j], sv[c], d[j]);

        if (G.pp < NP) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 o;
                const f32x4 bA = *(const f32x4*)(b1s + g * 16 + h * 8), bB = *(const f32x4*)(b1s + g * 16 + h * 8 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float bk0 = i < 2 ? bA[2 * i] : bB[2 * i - 4], bk1 = i < 2 ? bA[2 * i + 1] : bB[2 * i - 3];
                    f32x2_t t = El<EL>::unpack2(q[h * 4][i]) * w00;
                    t = __builtin_elementwise_fma(El<EL>::unpack2(q[h * 4 + 1][i]), w01, t);
                    t = __builtin_elementwise_fma(El<EL>::unpack2(q[h * 4 + 2][i]), w10, t);
                    t = __builtin_elementwise_fma(El<EL>::unpack2(q[h * 4 + 3][i]), w11, t);
                    const int k = h * 8 + i * 2;             // run h, positions 2i / 2i + 1 (conv_epilogue's v index)
                    const float x0v = fmaxf(d[k >> 2][k & 3] + bk0 + t[0], 0.f), x1v = fmaxf(d[(k + 1) >> 2][(k + 1) & 3] + bk1 + t[1], 0.f);
                    o[i] = G.inside ? El<EL>::pack2(x0v, x1v) : 0u;      // outside the image: the 3x3's zero padding
                }
                *(u32x4*)(pbuf + h * PCH + (g * NP + G.pp) * 16) = o;
            }
        }
    };
    int frag_off[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) frag_off[f] = ((wave * MF + f) * 18) * 16;
    const int lane_patch_off = (g * NP + pl) * 16;
    const int lane_w_off = lane * 16;

    for (int tile = xcd_remap(blockIdx.x, gridDim.x); tile < ntiles; tile += gridDim.x) {
        int id = tile;
        const int tx = id % a.tiles_x;
        id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int n = id / a.tiles_y;
        const int oy0 = ty * 16, ox0 = tx * 16;
        // ---- phase A: the patch.  Fragments wave, wave + 4, ... (21 in all); s1 loads run one fragment ahead
        v8 sA[4], sB[4];
        Geo G0 = geo(oy0, ox0, wave);
        issue_s(n, G0, sA);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int fi = wave + 4 * i;
            if (fi < NFRAG) {
                const bool more = fi + 4 < NFRAG;
                const Geo G1 = geo(oy0, ox0, more ? fi + 4 : fi);
                if (i & 1) produce(n, G0, sB, more, G1, sA);
                else produce(n, G0, sA, more, G1, sB);
                G0 = G1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // patch complete (and, first tile, the resident weights)
        // ---- phase B: 3x3 from LDS, no barrier inside
        f32x4 acc[MF][4];
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned char* pc = pbuf + c * PCH + lane_patch_off;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const unsigned char* wb = wres + (c * 9 + tap) * WSL + lane_w_off;
                const unsigned char* pb = pc + ((tap / 3) * 18 + (tap % 3)) * 16;
                v8 af[NF], bq[MF];
#pragma unroll
                for (int j = 0; j < NF; ++j) af[j] = *(const v8*)(wb + j * 1024);
#pragma unroll
                for (int f = 0; f < MF; ++f) bq[f] = *(const v8*)(pb + frag_off[f]);
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[f][j] = El<EL>::mfma(af[j], bq[f], acc[f][j]);
            }
        }
        // Everybody is done reading the patch: the next tile's phase A may overwrite it.  lgkmcnt(0) is PART of this barrier: the "memory"
        // clobber keeps the ds_reads above it, but hipcc sinks the last MFMAs -- and with them the s_waitcnt of the last fragment read -- below
        // the asm, so a wave could pass a bare s_barrier with a patch read still queued, a faster wave's phase-A ds_write of the NEXT tile could
        // be served first, and one tap of one fragment row was then multiplied with the next tile's pixels: heat-maps that differed from run
        // to run in ~1 % of the values on tiles after a workgroup's first (round 4: tools/determinism_probe.py, tools/micro/up4_check.hip).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        conv_epilogue<EL, MF>(a, acc, n, 0, oy0, ox0, wave, 0, 1, lane, 64);
    }
}

template <int EL>
static hipError_t launch_up4_el(const ConvArgs& a, hipStream_t s) {
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int ncu = device_cus();
    const size_t smem = 18 * 2048 + 2 * 324 * 64 + 256;  // 78.6 KB: two workgroups per CU
    static LdsOptIn attr;
    if (hipError_t e = lds_opt_in(attr, (const void*)conv3x3_up4_kernel<EL>, smem); e != hipSuccess) return e;
    const int grid = ntiles < 2 * ncu ? ntiles : 2 * ncu;
    hipLaunchKernelGGL(conv3x3_up4_kernel<EL>, dim3(grid), dim3(256), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_up4_fused(const ConvPlan& p1, const ConvPlan& p3, ConvArgs a, hipStream_t s) {
    static const bool on = (diag_knob("BBOCR_UP4_FUSED", 1) != 0);      // A/B knob (diagnostic builds)
    if (!on || p1.KH != 1 || p1.Cin != 128 || p1.Cout != 64 || p1.BN != 64 || p3.KH != 3 || p3.Cin != 64 || p3.Cout_pad != 64 || p3.BN != 64 ||
        p1.el != p3.el || p1.split || p3.split || a.in0_cs != 128 || a.up_cs != 64 || !a.addup || !a.zero || a.cout_store > 32 || (a.up_H & 1) || (a.up_W & 1) ||
        a.up_H != a.H || a.up_W != a.W)
        return hipErrorNotSupported;
    a.KH = a.KW = 3; a.pad_h = a.pad_w = 1; a.dil = 1; a.sub = 1;
    a.OH = a.H; a.OW = a.W; a.TH = a.TW = 16;
    a.tiles_x = (a.OW + 15) / 16; a.tiles_y = (a.OH + 15) / 16; a.ntiles_n = 1; a.nchunks = 2; a.ntaps = 9;
    a.aux_w = p1.d_w; a.aux_b = p1.d_b;
    a.wpk = p3.d_w; a.bias = p3.d_b;
    a.acc_scale = 1.f; a.dbg = 0;
    if ((long long)a.N * a.tiles_x * a.tiles_y > 0x7fffffffLL) return hipErrorNotSupported;
    return p3.el ? launch_up4_el<1>(a, s) : launch_up4_el<0>(a, s);
}

// ================================================================================================ 1x1, LDS-DMA staged
// 1x1 convolutions (fc7, the U-net "concat + 1x1" layers, the LSTM input projections, the linear layers, the class
// projection) are plain GEMMs over the flattened pixel axis: [pixels, Cin] x [Cin, Cout].  One k-step per 32-channel chunk;
// BOTH operand tiles of k-step ks+RING-1 are issued by LDS-DMA at k-step ks (weights: lane-linear slice; activations: wave w
// gathers channel group w of 256 consecutive pixels, 16 B per lane), and the barrier waits with the constant counted
// vmcnt((RING-2) * DMAs-per-k-step).  No halo, so the tile is simply 256 consecutive pixels of N*H*W.
template <int EL, int WM, int WN, int MF, int RING, bool ADDUP>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN == 8 ? 1 : 2)) conv1x1_dma_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN, NT = NW * 64, BN = WN * 64;
    static_assert(NW == 4 || NW == 8, "wave w gathers channel group w & 3 of the activation tile (8 waves: half of the pixel blocks each)");
    constexpr int WBUF = BN * 64, WPIECES = WBUF / 16, WPT = WPIECES / NT;
    constexpr int NPX = WM * MF * 16, NPB = NPX / 64, PBUF = NPX * 64, SLOT = WBUF + PBUF;
    constexpr int PBW = NPB / (NW / 4);            // activation blocks (64 pixels x 8 channels) a wave gathers per k-step
    static_assert(WPIECES % NT == 0 && NPX % 64 == 0 && NPB % (NW / 4) == 0 && RING >= 3, "uniform DMA issue");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [RING][weights WBUF | activations PBUF]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int kg = wave & 3, pb0 = (wave >> 2) * PBW;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % a.ntiles_n;
    const int tile = bid / a.ntiles_n;
    const int px0 = tile * NPX;                // first pixel of the tile in the flattened N*H*W axis (a.OW = total pixels)
    const int nk = a.nchunks;
    int pix[PBW];
#pragma unroll
    for (int i = 0; i < PBW; ++i) {
        const int p = px0 + (pb0 + i) * 64 + lane;
        pix[i] = p < a.OW ? p : -1;
    }
    const unsigned char* wsrc = (const unsigned char*)a.wpk + (size_t)nt * nk * WBUF;
    auto issue = [&](int ks, int slot) {
        unsigned char* sb = smem + slot * SLOT;
#pragma unroll
        for (int p0 = 0; p0 < WPIECES; p0 += NT) lds_dma16(wsrc + (size_t)ks * WBUF + (size_t)(p0 + tid) * 16, sb + (p0 + wave * 64) * 16);
        const int c = ks * 32;
        const bool s0 = c < a.C0;
        const uint16_t* src = s0 ? a.in0 : a.in1;
        const int cs = s0 ? a.in0_cs : a.in1_cs;
        const int cb = (s0 ? c : c - a.C0) + kg * 8;
#pragma unroll
        for (int i = 0; i < PBW; ++i) {
            const uint16_t* g = pix[i] >= 0 ? src + (size_t)pix[i] * cs + cb : (const uint16_t*)a.zero;
            lds_dma16(g, sb + WBUF + (kg * NPX + (pb0 + i) * 64) * 16);
        }
    };
    const int lane_p_off = WBUF + ((lane >> 4) * NPX + (lane & 15)) * 16 + wm * MF * 256;
    const int lane_w_off = wn * 4 * 1024 + lane * 16;
    f32x4 acc[MF][4];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RING - 1; ++i)
        if (i < nk) issue(i, i);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int slot = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const bool ahead = ks + RING - 1 < nk;
        if (ahead) {
            int sl = slot + RING - 1;
            if (sl >= RING) sl -= RING;
            issue(ks + RING - 1, sl);
        }
        const unsigned char* sb = smem + slot * SLOT;
        // (the pair-wise read-ahead schedule of conv3x3_dma_kernel measured SLOWER here -- fc7 393 -> 446 us: every k-step brings new A and
        // new B fragments, nothing can be read across the barrier, and the reads it spreads out delay the next DMA issue)
        typename El<EL>::v8 af[4], bq[MF];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *(const typename El<EL>::v8*)(sb + lane_w_off + j * 1024);
#pragma unroll
        for (int f = 0; f < MF; ++f) bq[f] = *(const typename El<EL>::v8*)(sb + lane_p_off + f * 256);
        if ((ks * 32 < a.C0) ? a.relu_in0 : a.relu_in1) {
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int f = 0; f < MF; ++f) bq[f] = __builtin_bit_cast(typename El<EL>::v8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, bq[f]), z));
        }
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[f][j] = El<EL>::mfma(af[j], bq[f], acc[f][j]);
        // k-step ks+1 was issued RING-2 k-steps ago; while the pipeline is full exactly RING-2 later issues follow it
        if (ahead) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((RING - 2) * (WPT + PBW)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        slot = slot + 1 == RING ? 0 : slot + 1;
    }
    conv_epilogue<EL, MF, ADDUP>(a, acc, 0, nt, 0, px0, wm, wn, 16, lane, BN);
}

template <int EL, int WM, int WN, int MF, bool ADDUP>
static hipError_t launch_dma1x1_k(const ConvArgs& a, long long grid, hipStream_t s) {
    constexpr int RING = 3, NPX = WM * MF * 16;
    auto k = conv1x1_dma_kernel<EL, WM, WN, MF, RING, ADDUP>;
    const size_t smem = (size_t)RING * ((size_t)WN * 64 * 64 + (size_t)NPX * 64);
    static LdsOptIn attr;
    if (hipError_t e = lds_opt_in(attr, (const void*)k, smem); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((int)grid), dim3(WM * WN * 64), smem, s, a);
    return hipGetLastError();
}

template <int EL, int WM, int WN, int MF>
static hipError_t launch_dma1x1(ConvArgs a, hipStream_t s) {
    constexpr int NPX = WM * MF * 16;
    const long long total = (long long)a.N * a.H * a.W;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    a.N = 1; a.OH = 1; a.OW = (int)total; a.TH = 1; a.TW = NPX; a.tiles_y = 1;
    a.tiles_x = (int)((total + NPX - 1) / NPX);
    const long long grid = (long long)a.tiles_x * a.ntiles_n;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    // the epilogue that adds an up-sampled tensor is its own instantiation: its register needs must not leak into the others
    // (8-wave tiles, one workgroup per CU: a ring of 4 measured no faster than 3 -- 389 vs 383 us on fc7)
    return a.addup ? launch_dma1x1_k<EL, WM, WN, MF, true>(a, grid, s) : launch_dma1x1_k<EL, WM, WN, MF, false>(a, grid, s);
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
                        const int cout = nt * BN + wn * 64 + (nf >> 1) * 32 + (row >> 2) * 8 + (nf & 1) * 4 + (row & 3);   // see conv_epilogue
                        for (int j = 0; j < 8; ++j) {
                            const int cin = c * 32 + 8 * (l >> 4) + j;
                            float v = 0.f;
                            if (cout < p.Cout && cin < p.Cin) v = w[((size_t)cout * p.Cin + cin) * ntaps + tap];
                            out[o++] = f32_to_el_host(p.el, v);
                        }
                    }
                }
}

template <int EL, int WM, int WN, int MF, int PITER>
static hipError_t launch_one(const ConvArgs& a, size_t smem, int grid, hipStream_t s) {
    auto k = conv_mfma_kernel<EL, WM, WN, MF, PITER>;
    static LdsOptIn attr;    // per instantiation and device: high-water mark of the opt-in dynamic LDS size
    if (hipError_t e = lds_opt_in(attr, (const void*)k, smem); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), smem, s, a);
    return hipGetLastError();
}

template <int EL, int WM, int WN, int MF, int PITER>
static hipError_t launch_cfg(const ConvArgs& a, int grid, hipStream_t s) {
    const size_t smem = (size_t)2 * WN * 64 * 64 + (size_t)2 * a.NP * 64;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    return launch_one<EL, WM, WN, MF, PITER>(a, smem, grid, s);
}

static bool conv_dma() {   // BBOCR_CONV_DMA=0 disables the LDS-DMA staged 3x3 variant (A/B runs)
    static const bool v = (diag_knob("BBOCR_CONV_DMA", 1) != 0);
    return v;
}

// BBOCR_CONV_STAMPS=<dir>: diagnostic tile timeline — every 3x3 DMA launch is followed by a sync and dumps its per-workgroup
// s_memtime stamps to <dir>/stamps_<seq>_<Cin>x<Cout>_<H>x<W>_g<grid>.bin (tools/tile_timeline.py reads them)
static const char* conv_stamps_dir() {
#ifdef BBOCR_DIAG
    static const char* d = getenv("BBOCR_CONV_STAMPS");
    return d;
#else
    return nullptr;
#endif
}

template <int EL, int WM, int WN, int MF, int NPB, int RING, int NPS = NPB * 64, bool FUSE1 = false, int NF = 4, int EPI = 0, int KS = 3>
static hipError_t launch_dma_one(ConvArgs a, int grid, hipStream_t s) {
    auto k = conv3x3_dma_kernel<EL, WM, WN, MF, NPB, RING, NPS, FUSE1, NF, EPI, KS>;
    const size_t smem_max = (size_t)RING * WN * 64 * 64 + (size_t)2 * NPS * 64;
    // a single 32-channel chunk (conv_cls: Cin = 32) never touches the second patch buffer: without it a workgroup needs 33 KB and
    // FOUR share a CU -- these launches are bound by per-tile latency, not by MFMA or HBM
    static const bool one_buf = (diag_knob("BBOCR_CONV_1BUF", 1) != 0);   // A/B knob
    const size_t smem = (one_buf && a.nchunks == 1 && !FUSE1) ? smem_max - (size_t)NPS * 64 : smem_max;
    static LdsOptIn attr;
    if (hipError_t e = lds_opt_in(attr, (const void*)k, smem_max); e != hipSuccess) return e;
    {   // which epilogue (conv_epilogue_plain_lean / conv_epilogue_pool2x2_lean / the shared one): plain layers whose waves store all their 64 couts
        static const int lean_knob = diag_knob("BBOCR_CONV_LEAN", 1);      // A/B knob: 0 off, 1 whole-line stores, 2 half-line stores (no DPP regroup)
        a.lean = 0;
        const bool plain = lean_knob && !FUSE1 && EPI == 0 && KS == 3 && NF == 4 && a.sub == 1 && !a.out_f32 && !a.tail && !a.split_off && !a.addup && !a.post_w &&
                           a.acc_scale == 1.f && a.cout_store == a.ntiles_n * WN * 64 && a.out_cs % 8 == 0;
        if (plain && a.pool_mode == 0 && a.out) a.lean = lean_knob == 2 ? 2 : 1;
        else if (plain && a.pool_mode == 1 && !a.store_full && a.TH == 16 && a.TW == 16 && a.pool_out) a.lean = 3;
    }
    if (const char* dir = conv_stamps_dir()) {
        static int seq = 0;
        unsigned long long* dev = nullptr;
        const size_t bytes = (size_t)grid * 4 * sizeof(unsigned long long);
        if (hipMalloc((void**)&dev, bytes) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(dev, 0, bytes, s);
        a.stamps = dev;
        hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), smem, s, a);
        hipError_t e = hipStreamSynchronize(s);
        if (e == hipSuccess) {
            unsigned long long* h = (unsigned long long*)malloc(bytes);
            e = hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost);
            char path[512];
            snprintf(path, sizeof(path), "%s/stamps_%03d_%dx%d_%dx%d_g%d_bn%d_r%d.bin", dir, seq++, a.C0 + a.C1, a.ntiles_n * WN * 64, a.H, a.W, grid,
                     WN * 64, RING);
            if (FILE* f = fopen(path, "wb")) { fwrite(h, 1, bytes, f); fclose(f); }
            free(h);
        }
        (void)hipFree(dev);
        return e;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), smem, s, a);
    return hipGetLastError();
}

template <int EL, int WM, int WN, int MF>
static hipError_t launch_dma(const ConvArgs& a, int npb, int grid, hipStream_t s) {
    const size_t wb = (size_t)WN * 64 * 64, pb = (size_t)2 * npb * 64 * 64;
    static const int ring_cap = diag_knob("BBOCR_DMA_RING", 4);
    const bool r4 = ring_cap >= 4 && 4 * wb + pb <= 80 * 1024;      // deepest ring that still lets two workgroups share a CU
    if constexpr (WN == 1) {
        // BN = 64: k-steps are short (16 MFMAs per wave), what pays is a THIRD co-resident workgroup: 16x16 tiles with the patch
        // trimmed to its 18 x 18 = 324 pixels and a 3-deep weight ring are 53,760 B of LDS (3 x 53,760 <= 160 KB)
        static const bool three = (diag_knob("BBOCR_CONV_3WG", 1) != 0);
        if (a.c11_w) {   // conv1_2 with the conv1_1 producer fused in
            if (!(npb == 6 && a.PH == 18 && a.PW == 18 && a.TH == 16 && a.TW == 16 && a.nchunks == 2 && a.sub == 1)) return hipErrorInvalidValue;
            // the kernel's only epilogue: 64 stored couts, MaxPool2d(2,2) fused, nothing but the pooled 16-bit tensor kept
            if (!(a.pool_mode == 1 && !a.store_full && !a.split_off && !a.out_f32 && !a.tail && a.cout_store == 64 && a.ntiles_n == 1 && a.acc_scale > 0.f))
                return hipErrorInvalidValue;
            return launch_dma_one<EL, WM, WN, MF, 6, 3, 324, true>(a, grid, s);
        }
        if (a.post_w) {   // 1x1 64 -> 64 behind the layer: 16 x 16 tiles of a plain 64-cout layer only, anything else is declined
            if (!(WN == 1 && three && npb == 6 && a.PH == 18 && a.PW == 18 && a.cout_store == 64 && a.ntiles_n == 1 && a.sub == 1 && !a.pool_mode &&
                  !a.split_off && !a.out_f32 && !a.tail && !a.addup))
                return hipErrorNotSupported;
            if constexpr (WN == 1) return launch_dma_one<EL, WM, WN, MF, 6, 3, 324, false, 4, 1>(a, grid, s);
        }
        if (three && npb == 6 && a.PH * a.PW == 324) {
            static const bool half = (diag_knob("BBOCR_CONV_NF2", 1) != 0);   // A/B knob
            static const bool resw = (diag_knob("BBOCR_CONV_RESW", 1) != 0);    // A/B knob
            if (resw && half && a.cout_store <= 32 && a.nchunks <= 2 && a.sub == 1 && !a.C1 && !a.pool_mode && a.TH == 16 && a.TW == 16 && a.ntiles_n == 1)
                return launch_resw<EL>(a, s);
            // one input chunk, 64 couts, MaxPool2d(2,2) fused and only the pooled tensor kept (the CRNN's 32 -> 64 layer): nine 4 KB slices resident
            static const bool resw64 = (diag_knob("BBOCR_CONV_RESW64", 1) != 0);    // A/B knob
            if (resw && resw64 && a.cout_store == 64 && a.nchunks == 1 && a.sub == 1 && !a.C1 && a.pool_mode == 1 && !a.store_full && !a.split_off && !a.out_f32 &&
                !a.tail && !a.relu_in0 && a.TH == 16 && a.TW == 16 && a.ntiles_n == 1 && a.acc_scale > 0.f)
                return launch_resw<EL>(a, s);
            if (half && a.cout_store <= 32) return launch_dma_one<EL, WM, WN, MF, 6, 3, 324, false, 2>(a, grid, s);
            return launch_dma_one<EL, WM, WN, MF, 6, 3, 324>(a, grid, s);
        }
        if (a.post_w) return hipErrorNotSupported;
        if (npb == 6) return launch_dma_one<EL, WM, WN, MF, 6, 3>(a, grid, s);      // measured faster than the 4-deep ring at this tile
    }
    if (a.post_w) return hipErrorNotSupported;
    if (npb == 6) return r4 ? launch_dma_one<EL, WM, WN, MF, 6, 4>(a, grid, s) : launch_dma_one<EL, WM, WN, MF, 6, 3>(a, grid, s);
    if (npb == 7) return launch_dma_one<EL, WM, WN, MF, 7, 3>(a, grid, s);
    return hipErrorInvalidValue;
}

int conv_plan_bn(int Cout) {   // couts per workgroup tile (BBOCR_BN64_UPTO: A/B knob, measured no gain for the 128-cout layers)
    static const int bn64_upto = diag_knob("BBOCR_BN64_UPTO", 64);
    return Cout > bn64_upto ? 128 : 64;
}

template <int EL>
static hipError_t launch_conv_el(const ConvPlan& p, ConvArgs a, hipStream_t s) {
    const int BN = p.BN;
    constexpr int NWV = 4;       // waves per workgroup
    constexpr int BM = 256;      // output pixels per workgroup tile
    a.KH = p.KH; a.KW = p.KW; a.pad_h = p.pad_h; a.pad_w = p.pad_w; a.dil = p.dil;
    a.OH = a.H + 2 * p.pad_h - (p.KH - 1) * p.dil;
    a.OW = a.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    if (a.OH <= 0 || a.OW <= 0) return hipErrorInvalidValue;
    a.ntaps = p.KH * p.KW;
    a.sub = 1;
    // a 1x1 behind the layer (a.post_w) exists for plain 3x3 layers with one 64-cout tile on the LDS-DMA kernel; the caller falls back to two launches
    if (a.post_w && !(conv_dma() && a.zero && p.KH == 3 && p.KW == 3 && p.dil == 1 && p.pad_h == 1 && p.pad_w == 1 && BN == 64 && p.Cout_pad == 64))
        return hipErrorNotSupported;
    if (conv_dma() && a.zero && !a.pool_mode && p.KH == 3 && p.KW == 3 && p.dil > 1 && p.pad_h == p.dil && p.pad_w == p.dil) {
        const int d = p.dil, LH = cdiv(a.OH, d), LW = cdiv(a.OW, d);   // phase sub-lattice size
        long long best = -1;
        for (int th = 16; th >= 4; th >>= 1) {
            const int tw = 256 / th, ph = th + 2, pw = tw + 2, npb = cdiv(ph * pw, 64);
            if (npb != 6 && npb != 7) continue;
            // the d*d phase images of a page are stacked along y with one zero row between them (it is the padding row of both
            // neighbours), so that tiles do not end at every 10-row phase image: fc6 fills 80 % of its MFMA rows instead of 55 %
            const long long cost = (long long)cdiv(d * d * (LH + 1), th) * cdiv(LW, tw);
            if (best < 0 || cost < best) { best = cost; a.TH = th; a.TW = tw; a.PH = ph; a.PW = pw; a.NP = npb * 64; }
        }
        if (best > 0 && (a.C0 + a.C1 == p.Cin_pad) && !(a.C0 & 31) && !(a.C1 & 31) && [&] {
                // the kernel divides a virtual row by LH+1 as (vy * ceil(2^20 / (LH+1))) >> 20: exact while vy * (magic * (LH+1) - 2^20) < 2^20
                const long long vmax = (long long)d * d * (LH + 1) + 18, magic = ((1LL << 20) + LH) / (LH + 1);
                return vmax * (magic * (LH + 1) - (1LL << 20)) < (1LL << 20) && vmax * magic < (1LL << 31);
            }()) {
            a.sub = d;
            a.stack = LH;
            a.tiles_x = cdiv(LW, a.TW);
            a.tiles_y = cdiv(d * d * (LH + 1), a.TH);
            a.ntiles_n = p.Cout_pad / BN;
            a.nchunks = p.Cin_pad / 32;
            a.wpk = p.d_w;
            a.bias = p.d_b;
            a.dbg = 0;
            const long long g = (long long)a.N * a.tiles_x * a.tiles_y * a.ntiles_n;
            if (g > 0 && g <= 0x7fffffffLL)
                return BN == 128 ? launch_dma<EL, 2, 2, 8>(a, a.NP / 64, (int)g, s) : launch_dma<EL, 4, 1, 4>(a, a.NP / 64, (int)g, s);
            a.sub = 1;
            a.stack = 0;
        }
    }
    const int ring = 2;
    const int max_piter = 16;
    // 2x2 / padding 0 (the CRNN's last conv, 4 input rows -> 3) on the LDS-DMA kernel: 4 x 64 tiles, whose 5 x 65 patch is six 64-pixel blocks
    const bool dma2x2 = conv_dma() && a.zero && p.KH == 2 && p.KW == 2 && p.dil == 1 && p.pad_h == 0 && p.pad_w == 0 && !a.pool_mode && !a.tail &&
                        !a.addup && !a.post_w && (BN == 128 || BN == 64);
    // tile shape: the TH x (BM/TH) rectangle with the least (MFMA work on partial tiles + patch staging) per layer
    if (dma2x2) {
        a.TH = 4; a.TW = 64; a.PH = 5; a.PW = 65; a.NP = 384;
    } else {
        long long best = -1;
        for (int th = 16; th >= 4; th >>= 1) {
            const int tw = BM / th;
            const int ph = th + (p.KH - 1) * p.dil, pw = tw + (p.KW - 1) * p.dil;
            const int np = cdiv(ph * pw, 16) * 16;
            const int pit = cdiv(np / 16, NWV), pit_r = pit <= 4 ? 4 : (pit <= 8 ? 8 : 16), nparts = pit_r / 4;
            if (pit > max_piter || (size_t)ring * BN * 64 + (size_t)2 * np * 64 > 160 * 1024) continue;
            if (a.ntaps >= 3 ? (a.ntaps - 1) / nparts < 1 : nparts > 1) continue;
            if (a.pool_mode && (BN == 128 ? 8 : 4) % (2 * (tw / 16)) != 0) continue;
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
#ifdef BBOCR_DIAG
    static const int dbg = diag_knob("BBOCR_CONV_DBG", 0);
    a.dbg = dbg;
#else
    a.dbg = 0;
#endif
    const long long grid_ll = (long long)a.N * a.tiles_x * a.tiles_y * a.ntiles_n;
    if (grid_ll <= 0 || grid_ll > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)grid_ll;
    if (a.addup && !(conv_dma() && p.KH == 1 && p.KW == 1 && a.zero && !a.pool_mode && !a.out_f32 && !a.tail && a.cout_store % 64 == 0 &&
                     a.cout_store == p.Cout_pad))
        return hipErrorInvalidValue;
    if (conv_dma() && p.KH == 1 && p.KW == 1 && p.pad_h == 0 && p.pad_w == 0 && a.zero && !a.pool_mode)
        return BN == 256 ? launch_dma1x1<EL, 2, 4, 8>(a, s) : (BN == 128 ? launch_dma1x1<EL, 2, 2, 8>(a, s) : launch_dma1x1<EL, 4, 1, 4>(a, s));
    if (dma2x2)
        return BN == 128 ? launch_dma_one<EL, 2, 2, 8, 6, 4, 384, false, 4, 0, 2>(a, grid, s) : launch_dma_one<EL, 4, 1, 4, 6, 3, 384, false, 4, 0, 2>(a, grid, s);
    if (conv_dma() && p.KH == 3 && p.KW == 3 && p.dil == 1 && p.pad_h == 1 && p.pad_w == 1 && a.zero) {
        const int npb = cdiv(a.PH * a.PW, 64);
        if (npb == 6 || npb == 7) {
            a.NP = npb * 64;
            return BN == 128 ? launch_dma<EL, 2, 2, 8>(a, npb, grid, s) : launch_dma<EL, 4, 1, 4>(a, npb, grid, s);
        }
    }
    if (a.post_w) return hipErrorNotSupported;
    // the generic kernel below knows neither the fused conv1_1 producer (in0 would be read as a 64-channel tensor: an out-of-bounds access on
    // the uint8 page -- what BBOCR_CONV_DMA=0 did in a diagnostic build in round 4) nor the up-sampling epilogue: refuse, never run garbage
    if (a.c11_w || a.addup) return hipErrorNotSupported;
    const int piter = cdiv(a.NP / 16, NWV);
    if (BN == 128) return piter <= 4 ? launch_cfg<EL, 2, 2, 8, 4>(a, grid, s) : (piter <= 8 ? launch_cfg<EL, 2, 2, 8, 8>(a, grid, s) : launch_cfg<EL, 2, 2, 8, 16>(a, grid, s));
    if (BN == 64) return piter <= 4 ? launch_cfg<EL, 4, 1, 4, 4>(a, grid, s) : (piter <= 8 ? launch_cfg<EL, 4, 1, 4, 8>(a, grid, s) : launch_cfg<EL, 4, 1, 4, 16>(a, grid, s));
    return hipErrorInvalidValue;
}

hipError_t launch_conv(const ConvPlan& p, ConvArgs a, hipStream_t s) {
    a.acc_scale = p.acc_scale;
    if (a.split_off && (a.out_f32 || a.tail || a.addup || p.el != 1)) return hipErrorInvalidValue;   // hi|lo outputs: fp16 stores only
    return p.el ? launch_conv_el<1>(p, a, s) : launch_conv_el<0>(p, a, s);
}
