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
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// 512-thread configs run one workgroup per CU (2 waves/SIMD inside it); 256-thread configs are built for TWO co-resident
// workgroups per CU (launch bound 2 waves/SIMD = 256 VGPRs) whose LDS-read and MFMA phases overlap each other.
template <int WM, int WN, int MF, int PITER, bool PIPE>
__global__ void __launch_bounds__(WM * WN * 64, 2) conv_mfma_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN, NT = NW * 64, BN = WN * 64;
    constexpr int WBUF = BN * 64;          // bytes of one weight k-step slice (BN couts x 32 k x 2 B)
    constexpr int WPIECES = WBUF / 16;     // 16-B pieces per slice
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WRING = PIPE ? 3 : 2;
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
    auto load_part = [&](int chunk, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        const int c = chunk * 32;
        const bool s0 = c < a.C0;
        const uint16_t* src = s0 ? a.in0 : a.in1;
        const int cs = s0 ? a.in0_cs : a.in1_cs;
        const int cb = (s0 ? c : c - a.C0) + l_kg * 8;
        const bool relu = s0 ? a.relu_in0 : a.relu_in1;
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (src_pix[part * PH + i] >= 0) {
                v = *(const u32x4*)(src + (size_t)src_pix[part * PH + i] * cs + cb);
                if (relu) {
                    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    v = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
                }
            }
            pre[i] = v;
        }
    };
    auto store_part = [&](int buf, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        unsigned char* pb = pbuf + buf * patch_bytes;
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int wi = wave + (part * PH + i) * NW;
            if (wi < n_wi) *(u32x4*)(pb + (size_t)(l_kg * a.NP + wi * 16 + l_pix) * 16) = pre[i];
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
        if (tap == S - 1) store_part(chunk & 1, P0{});
        if constexpr (NPART > 1) { if (tap == 2 * S - 1) store_part(chunk & 1, std::integral_constant<int, (NPART > 1 ? 1 : 0)>{}); }
        if constexpr (NPART > 2) { if (tap == 3 * S - 1) store_part(chunk & 1, std::integral_constant<int, (NPART > 2 ? 2 : 0)>{}); }
        if constexpr (NPART > 3) { if (tap == 4 * S - 1) store_part(chunk & 1, std::integral_constant<int, (NPART > 3 ? 3 : 0)>{}); }
    };
    auto stage_patch_now = [&](int chunk) {        // prologue: whole patch, synchronously
        load_part(chunk, P0{});
        store_part(chunk & 1, P0{});
        if constexpr (NPART > 1) { load_part(chunk, std::integral_constant<int, (NPART > 1 ? 1 : 0)>{}); store_part(chunk & 1, std::integral_constant<int, (NPART > 1 ? 1 : 0)>{}); }
        if constexpr (NPART > 2) { load_part(chunk, std::integral_constant<int, (NPART > 2 ? 2 : 0)>{}); store_part(chunk & 1, std::integral_constant<int, (NPART > 2 ? 2 : 0)>{}); }
        if constexpr (NPART > 3) { load_part(chunk, std::integral_constant<int, (NPART > 3 ? 3 : 0)>{}); store_part(chunk & 1, std::integral_constant<int, (NPART > 3 ? 3 : 0)>{}); }
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

    if constexpr (!PIPE) {
        // reference schedule: 2-deep weight ring, fragments read at the top of every k-step
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
                bf16x8 af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j] = *(const bf16x8*)(wb + j * 1024);
                const unsigned char* pb = pbase + ((ky * a.PW + kx) * a.dil) * 16;
                // all activation fragments are requested up front (MF ds_read_b128 in flight) so the LDS latency is paid
                // once per k-step; left to itself hipcc serialises read -> lgkmcnt(0) -> 4 MFMAs per fragment
                bf16x8 bq[MF];
#pragma unroll
                for (int f = 0; f < MF; ++f) bq[f] = *(const bf16x8*)(pb + frag_off[f]);
#pragma unroll
                for (int f = 0; f < MF; ++f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bq[f], acc[f][j], 0, 0, 0);
                }
                // pin the order: every fragment read first, then the MFMA stream behind counted lgkmcnt waits
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + MF, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * MF, 0);
                if (more) {
                    if (a.ntaps >= 3) stage_patch_store(c + 1, tap);
                    else if (tap == a.ntaps - 1) store_part((c + 1) & 1, P0{});
                }
                __syncthreads();
                if (++kx == a.KW) { kx = 0; ++ky; }
            }
        }
    } else {
        // software-pipelined schedule (taps >= 3): 3-deep weight ring; each k-step is two halves of 2*MF MFMAs.  While
        // the first half issues, the second half's activation fragments are in flight; while the second half issues, the
        // NEXT k-step's weight fragments and first-half activation fragments are in flight.  A wave therefore leaves the
        // barrier with operands in registers and never waits on LDS latency in front of an MFMA group.  The next chunk's
        // patch must be visible one k-step earlier than in the reference schedule (stored by tap ntaps-2).
        constexpr int HF = MF / 2;
        issue_w(0, 0);
        if (nk > 1) issue_w(1, 1);
        stage_patch_now(0);
        __syncthreads();
        bf16x8 af[4], an[4], bq[MF];
        {
            const unsigned char* wb = wbuf + lane_w_off;
            const unsigned char* pb = pbuf + lane_patch_off;
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *(const bf16x8*)(wb + j * 1024);
#pragma unroll
            for (int f = 0; f < HF; ++f) bq[f] = *(const bf16x8*)(pb + frag_off[f]);
        }
        int c = 0, tap = 0, ky = 0, kx = 0, wslot = 0;
        for (int ks = 0; ks < nk; ++ks) {
            const bool more = (c + 1 < a.nchunks);
            if (more) stage_patch_load(c + 1, tap);
            if (ks + 2 < nk) issue_w(ks + 2, wslot == 0 ? 2 : wslot - 1);   // slot (ks+2)%3 == (ks-1)%3
            int c1 = c, tap1 = tap + 1, ky1 = ky, kx1 = kx + 1;
            if (kx1 == a.KW) { kx1 = 0; ++ky1; }
            if (tap1 == a.ntaps) { tap1 = 0; ky1 = 0; kx1 = 0; ++c1; }
            const bool has_next = (ks + 1 < nk);
            const int wslot1 = wslot == 2 ? 0 : wslot + 1;
            const unsigned char* pb0 = pbuf + (c & 1) * patch_bytes + lane_patch_off + ((ky * a.PW + kx) * a.dil) * 16;
            // the last k-step re-reads its own operands instead of branching around the prefetch
            const unsigned char* wb1 = wbuf + (has_next ? wslot1 : wslot) * WBUF + lane_w_off;
            const unsigned char* pb1 = has_next ? pbuf + (c1 & 1) * patch_bytes + lane_patch_off + ((ky1 * a.PW + kx1) * a.dil) * 16 : pb0;
#pragma unroll
            for (int f = HF; f < MF; ++f) bq[f] = *(const bf16x8*)(pb0 + frag_off[f]);
#pragma unroll
            for (int f = 0; f < HF; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bq[f], acc[f][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) an[j] = *(const bf16x8*)(wb1 + j * 1024);
#pragma unroll
            for (int f = 0; f < HF; ++f) bq[f] = *(const bf16x8*)(pb1 + frag_off[f]);
#pragma unroll
            for (int f = HF; f < MF; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bq[f], acc[f][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, HF, 0);          // second-half activation reads
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * HF, 0);      // first-half MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4 + HF, 0);      // next k-step: weights + first-half activations
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * HF, 0);      // second-half MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = an[j];
            if (more) stage_patch_store(c + 1, tap);
            __syncthreads();
            c = c1; tap = tap1; ky = ky1; kx = kx1; wslot = wslot1;
        }
    }

    // ---- epilogue: bias (+ReLU), each lane owns 16 contiguous couts of its pixel per fragment
    const int g = lane >> 4, pl = lane & 15;
    const int cout0 = nt * BN + wn * 64 + g * 16;
    if (cout0 >= a.cout_store) return;
    float bs[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 b4 = *(const f32x4*)(a.bias + cout0 + j * 4);
        bs[j * 4 + 0] = b4[0]; bs[j * 4 + 1] = b4[1]; bs[j * 4 + 2] = b4[2]; bs[j * 4 + 3] = b4[3];
    }
    auto store16 = [&](void* base, size_t o, const float (&v)[16], bool f32) {
        if (f32) {
            float* op = (float*)base + o;
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(op + j * 4) = (f32x4){v[j * 4], v[j * 4 + 1], v[j * 4 + 2], v[j * 4 + 3]};
        } else {
            uint16_t* op = (uint16_t*)base + o;
            const u32x4 lo = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            const u32x4 hi = {pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15])};
            *(u32x4*)(op) = lo;
            *(u32x4*)(op + 8) = hi;
        }
    };
    if (a.pool_mode == 0) {
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
                store16(a.out, ((size_t)(n * a.OH + oy) * a.OW + ox) * a.out_cs + cout0, v, a.out_f32);
            }
        }
        return;
    }
    // ---- fused max-pool (MaxPool2d(2,2) or MaxPool2d((2,1),(2,1))): the two rows of a pooling window are two fragments of
    // this wave (the launcher only picks tiles with MF % (2*fpr) == 0), the two columns are lanes l and l^1.
    auto pooled = [&](auto fpr_c) {
        constexpr int FPR = decltype(fpr_c)::value;
        if constexpr (MF % (2 * FPR) == 0) {
            const int POH = a.OH >> 1, POW = a.pool_mode == 1 ? (a.OW >> 1) : a.OW;
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if ((f / FPR) & 1) continue;                 // odd tile rows are consumed by their even partner
                const int F = wm * MF + f;
                const int fr = F / FPR, fc = F - fr * FPR;
                const int oy = oy0 + fr, ox = ox0 + fc * 16 + pl;
                float v0[16], v1[16], m[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x0 = acc[f][j][r] + bs[j * 4 + r], x1 = acc[f + FPR][j][r] + bs[j * 4 + r];
                        if (a.relu_out) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
                        v0[j * 4 + r] = x0;
                        v1[j * 4 + r] = x1;
                        float p = fmaxf(x0, x1);
                        if (a.pool_relu) p = fmaxf(p, 0.f);
                        m[j * 4 + r] = p;
                    }
                if (a.store_full && ox < a.OW) {
                    if (oy < a.OH) store16(a.out, ((size_t)(n * a.OH + oy) * a.OW + ox) * a.out_cs + cout0, v0, a.out_f32);
                    if (oy + 1 < a.OH) store16(a.out, ((size_t)(n * a.OH + oy + 1) * a.OW + ox) * a.out_cs + cout0, v1, a.out_f32);
                }
                if (a.pool_mode == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) m[i] = fmaxf(m[i], __shfl_xor(m[i], 1));
                    const int py = oy >> 1, px = ox >> 1;
                    if (!(pl & 1) && py < POH && px < POW)
                        store16(a.pool_out, ((size_t)(n * POH + py) * POW + px) * a.pool_cs + cout0, m, false);
                } else {
                    const int py = oy >> 1;
                    if (py < POH && ox < POW) store16(a.pool_out, ((size_t)(n * POH + py) * POW + ox) * a.pool_cs + cout0, m, false);
                }
            }
        }
    };
    if (fpr == 1) pooled(std::integral_constant<int, 1>{});
    else if (fpr == 2) pooled(std::integral_constant<int, 2>{});
    else pooled(std::integral_constant<int, 4>{});
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

template <int WM, int WN, int MF, int PITER, bool PIPE>
static hipError_t launch_one(const ConvArgs& a, size_t smem, int grid, hipStream_t s) {
    auto k = conv_mfma_kernel<WM, WN, MF, PITER, PIPE>;
    static size_t cur = 0;   // per-instantiation high-water mark of the opt-in dynamic LDS size
    if (smem > cur) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        cur = smem;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), smem, s, a);
    return hipGetLastError();
}

static bool conv_pipelined() {   // BBOCR_CONV_PIPE=1 selects the software-pipelined schedule (A/B runs; measured 6 % slower)
    static const bool v = [] { const char* e = getenv("BBOCR_CONV_PIPE"); return e && e[0] == '1'; }();
    return v;
}

#define NP_TWO_PART(P) ((P) > 4)
template <int WM, int WN, int MF, int PITER>
static hipError_t launch_cfg(const ConvArgs& a, int grid, hipStream_t s) {
    const bool pipe = conv_pipelined() && a.ntaps >= 3 && PITER <= 8;
    const size_t smem = (size_t)(pipe ? 3 : 2) * WN * 64 * 64 + (size_t)2 * a.NP * 64;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    return pipe ? launch_one<WM, WN, MF, PITER, true>(a, smem, grid, s) : launch_one<WM, WN, MF, PITER, false>(a, smem, grid, s);
}

static bool conv_small_wg() {   // BBOCR_CONV_WG=512 selects the original one-workgroup-per-CU configurations (A/B runs)
    static const bool v = [] { const char* e = getenv("BBOCR_CONV_WG"); return !(e && atoi(e) == 512); }();
    return v;
}

int conv_plan_bn(int Cout) {
    if (conv_small_wg()) return Cout > 64 ? 128 : 64;
    return Cout >= 256 ? 256 : (Cout > 64 ? 128 : 64);
}

hipError_t launch_conv(const ConvPlan& p, ConvArgs a, hipStream_t s) {
    const int BN = p.BN;
    const bool small = conv_small_wg();
    const int NWV = small ? 4 : 8;                                  // waves per workgroup
    const int BM = small ? 256 : ((BN == 256) ? 256 : 512);
    a.KH = p.KH; a.KW = p.KW; a.pad_h = p.pad_h; a.pad_w = p.pad_w; a.dil = p.dil;
    a.OH = a.H + 2 * p.pad_h - (p.KH - 1) * p.dil;
    a.OW = a.W + 2 * p.pad_w - (p.KW - 1) * p.dil;
    if (a.OH <= 0 || a.OW <= 0) return hipErrorInvalidValue;
    a.ntaps = p.KH * p.KW;
    const int ring = conv_pipelined() ? 3 : 2;
    const int max_piter = small ? 16 : 8;
    // tile shape: the TH x (BM/TH) rectangle with the least (MFMA work on partial tiles + patch staging) per layer
    {
        long long best = -1;
        for (int th = 16; th >= 4; th >>= 1) {
            const int tw = BM / th;
            const int ph = th + (p.KH - 1) * p.dil, pw = tw + (p.KW - 1) * p.dil;
            const int np = cdiv(ph * pw, 16) * 16;
            const int pit = cdiv(np / 16, NWV), pit_r = pit <= 4 ? 4 : (pit <= 8 ? 8 : 16), nparts = pit_r / 4;
            if (pit > max_piter || (size_t)ring * BN * 64 + (size_t)2 * np * 64 > 160 * 1024) continue;
            if (a.ntaps >= 3 ? (a.ntaps - 1) / nparts < 1 : nparts > 1) continue;
            if (a.pool_mode && (small ? (BN == 128 ? 8 : 4) : (BN == 64 ? 4 : 8)) % (2 * (tw / 16)) != 0) continue;
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
    const long long grid_ll = (long long)a.N * a.tiles_x * a.tiles_y * a.ntiles_n;
    if (grid_ll <= 0 || grid_ll > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)grid_ll;
    const int piter = cdiv(a.NP / 16, NWV);
    if (small) {
        if (BN == 128) return piter <= 4 ? launch_cfg<2, 2, 8, 4>(a, grid, s) : (piter <= 8 ? launch_cfg<2, 2, 8, 8>(a, grid, s) : launch_cfg<2, 2, 8, 16>(a, grid, s));
        if (BN == 64) return piter <= 4 ? launch_cfg<4, 1, 4, 4>(a, grid, s) : (piter <= 8 ? launch_cfg<4, 1, 4, 8>(a, grid, s) : launch_cfg<4, 1, 4, 16>(a, grid, s));
        return hipErrorInvalidValue;
    }
    if (BN == 256) return piter <= 4 ? launch_cfg<2, 4, 8, 4>(a, grid, s) : launch_cfg<2, 4, 8, 8>(a, grid, s);
    if (BN == 128) return piter <= 4 ? launch_cfg<4, 2, 8, 4>(a, grid, s) : launch_cfg<4, 2, 8, 8>(a, grid, s);
    if (BN == 64) return piter <= 4 ? launch_cfg<8, 1, 4, 4>(a, grid, s) : launch_cfg<8, 1, 4, 8>(a, grid, s);
    return hipErrorInvalidValue;
}
