// Bidirectional LSTM recurrence (hidden 256) as a persistent per-sequence-tile kernel on MFMA.
//
// Restates torch.nn.LSTM(256, 256, bidirectional=True, batch_first=True) as used by
// easyocr/model/modules.py::BidirectionalLSTM (recogniser SequenceModeling.{0,1}.rnn), reached from
// reader.readtext (pipeline_demo/extractor/enhanced_extractor.py:520).  The input projection x W_ih^T + b_ih + b_hh
// is ONE batched GEMM done by conv_mfma (1x1 conv, permuted output channels); this kernel does the T sequential
// steps.  One workgroup = 16 sequences x one direction, 8 waves (two per SIMD); wave w owns hidden units [32w, 32w+32) and
// all four of their gates, so the gate non-linearity is lane-local:
//   D[seq][gate col] = h_{t-1}[seq][k] * W_hh^T[k][gate col]   (A = h from LDS, B = W_hh in fragment order)
//   fragment (a, gate): lane l holds unit 32w+16*a+(l&15) for sequences 4(l>>4)+r.
// c stays in fp32 registers for all T steps; h crosses LDS as bf16 (it is the next step's A operand).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

size_t lstm_whh_packed_elems() { return (size_t)2 * 1024 * 256; }
__device__ __forceinline__ constexpr size_t lstm_whh_packed_elems_dev() { return (size_t)2 * 1024 * 256; }

// v_exp_f32 + v_rcp_f32 (1 ulp each): 2 transcendental issues per non-linearity instead of an IEEE division sequence
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.442695041f)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.885390082f)); }

// tiles[b] = {first row of the tile's first sequence in the pooled [rows, C] tensors, sequences in the tile (<= 16), T, -};
// sequence s of a tile owns rows [row0 + s*T, row0 + (s+1)*T).  Tiles of different buckets (different T) share one launch.
//
// Wave w owns hidden units [32w, 32w+32) = two 16-unit groups
// x four gates = 64 weight fragments (1 KiB each), one MFMA per fragment and step.  A step is [64 MFMAs][gate math][barrier];
// what bounds it is latency, so every operand of the MFMA stream is in a register BEFORE its MFMA is due:
//   visit order v = kk*8 + group*4 + gate (kk = 32-wide slice of h), fragment v lives
//     v in [10,38)            28 fragments  in registers for all T steps,
//     v in [46,64)            18 fragments  in LDS (read LSTM8_LD MFMAs ahead into a small register ring),
//     v in [0,10) u [38,46)   18 fragments  re-streamed from L2 every step through a 10-fragment register buffer: the first
//                                           ten are loaded during the PREVIOUS step's gate math (the buffer is idle then), the
//                                           other eight as the first ten are consumed -- 28 register MFMAs later they are due.
//   h fragments are read one kk ahead (2 x 4 registers instead of all eight), x_t is loaded at the top of its own step.
// Packed layout [dir][wave 8][v 64][lane 64][8]; input-projection channels permuted so that a lane's 8 gate
// pre-activations of one sequence are 16 contiguous bytes (lstm8_xproj_channel).
// LSTM8_ABL: timing-only ablations of lstm8_kernel (garbage results; variant builds for tools/lstm_abl.sh only): 1 no transcendentals in the gate
// math, 2 no L2-streamed weight fragments, 4 no LDS weight reads, 8 no x loads, 16 no output store.  Measured (2,048 sequences, T = 255, 887 us):
// -5 / -11 / -3 / -15 / -11 %, all five together -46 %: no single term bounds a step, the memory side as a whole is a third of it.
#ifndef LSTM8_ABL
#define LSTM8_ABL 0
#endif
#ifndef LSTM8_S0
#define LSTM8_S0 10     // streamed fragments held at the top of a step
#endif
#ifndef LSTM8_NR
#define LSTM8_NR 28
#endif
#ifndef LSTM8_S1
#define LSTM8_S1 8      // streamed fragments fetched while the first ones are consumed
#endif
#ifndef LSTM8_NL
#define LSTM8_NL 18
#endif
#ifndef LSTM8_LD
#define LSTM8_LD 8      // LDS read-ahead (fragments); the ring reuses the registers of the (by then consumed) streamed buffer
#endif
void pack_lstm_whh8(const float* whh_fwd, const float* whh_bwd, uint16_t* out, int el) {
    size_t o = 0;
    for (int d = 0; d < 2; ++d) {
        const float* W = d ? whh_bwd : whh_fwd;
        for (int w = 0; w < 8; ++w)
            for (int kk = 0; kk < 8; ++kk)
                for (int a = 0; a < 2; ++a)
                    for (int gate = 0; gate < 4; ++gate)
                        for (int l = 0; l < 64; ++l) {
                            const int unit = w * 32 + a * 16 + (l & 15);
                            const int row = gate * 256 + unit;
                            for (int j = 0; j < 8; ++j) out[o++] = f32_to_el_host(el, W[(size_t)row * 256 + kk * 32 + 8 * (l >> 4) + j]);
                        }
    }
}
int lstm8_xproj_channel(int dir, int gate, int unit) {
    return dir * 1024 + (((unit >> 5) * 16 + (unit & 15)) * 8) + ((unit >> 4) & 1) * 4 + gate;
}

template <int EL>
__global__ void __launch_bounds__(512, 2) lstm8_kernel(const uint16_t* __restrict__ xproj, const uint16_t* __restrict__ whh,
                                                       uint16_t* __restrict__ out, const int4* __restrict__ tiles, int pf_dist) {
    static_assert(LSTM8_S0 + LSTM8_NR + LSTM8_S1 + LSTM8_NL == 64 && LSTM8_S1 <= LSTM8_S0, "fragment classes");
    constexpr int R0 = LSTM8_S0, S1 = LSTM8_S0 + LSTM8_NR, L0 = S1 + LSTM8_S1;   // first v of the register / late-streamed / LDS class
    extern __shared__ __attribute__((aligned(16))) unsigned char lstm_smem[];
    unsigned char (*hbuf)[32 * 16 * 16] = (unsigned char (*)[32 * 16 * 16])lstm_smem;   // [2][kgroup 32][seq 16] x 16 B
    unsigned char* const wlds = lstm_smem + 2 * 32 * 16 * 16;                             // [wave 8][LSTM8_NL][lane 64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const int4 tile = tiles[blockIdx.x];
    const int row0 = tile.x, n = tile.y, T = tile.z;
    const int g = lane >> 4, u = lane & 15;
    for (int i = tid; i < 2 * 32 * 16 * 16 / 16; i += 512) ((u32x4*)lstm_smem)[i] = (u32x4){0u, 0u, 0u, 0u};
    float c[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[a][r] = 0.f;
    typedef const __attribute__((address_space(1))) typename El<EL>::v8* gfrag_ptr;     // global_load (vmcnt only), never flat_load
    const gfrag_ptr wv0 = (gfrag_ptr)((const typename El<EL>::v8*)whh + ((size_t)(dir * 8 + wave) * 64) * 64 + lane);
    typename El<EL>::v8 wreg[LSTM8_NR];
#pragma unroll
    for (int i = 0; i < LSTM8_NR; ++i) wreg[i] = wv0[(size_t)(R0 + i) * 64];
    typename El<EL>::v8* const wl = (typename El<EL>::v8*)(wlds + (size_t)wave * LSTM8_NL * 1024) + lane;
#pragma unroll 2
    for (int i = 0; i < LSTM8_NL; ++i) wl[(size_t)i * 64] = wv0[(size_t)(L0 + i) * 64];
    const int xch = dir * 1024 + (wave * 16 + u) * 8;
    const int wb_seq = tid >> 5, wb_kg = tid & 31;   // h write-back: 16 B per thread
    // rows beyond n only feed their own (never stored) outputs: clamp their address instead of branching around the load
    size_t xrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xrow[r] = ((size_t)row0 + (size_t)(g * 4 + r < n ? g * 4 + r : n - 1) * T) * 2048 + xch;
    // x_t comes from HBM (the projection of ALL time steps was written long before), ~2 us away, and the gate math needs it ~1 us after
    // the top of the step: threads 0..255 touch the 256 cache lines of x_{t+2} (16 sequences x 2 KB) one dword each, two steps ahead,
    // so that the real load finds them in L2.  The value is only kept alive until the end of the step (so that hipcc accounts for it).
    const bool pf_on = tid < 256 && pf_dist > 0;
    const size_t pf_row = ((size_t)row0 + (size_t)((tid >> 4) < n ? (tid >> 4) : n - 1) * T) * 2048 + dir * 1024 + (tid & 15) * 64;
    typename El<EL>::v8 sb[LSTM8_S0];
    auto stream_head = [&]() {        // fragments v = 0 .. S0-1 of the NEXT step
        gfrag_ptr wv = wv0;
        asm volatile("" : "+v"(wv));  // keep these loads inside the time loop (they would otherwise be hoisted and spilled)
#pragma unroll
        for (int i = 0; i < LSTM8_S0; ++i) sb[i] = (LSTM8_ABL & 2) ? wreg[i] : wv[(size_t)i * 64];
    };
    stream_head();
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        u32x4 xq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xq[r] = (LSTM8_ABL & 8) ? (u32x4){0u, 0u, 0u, 0u} : *(const u32x4*)(xproj + xrow[r] + (size_t)t * 2048);
        const unsigned char* hb = hbuf[cur] + (g * 16 + u) * 16;
        unsigned char* hn = hbuf[cur ^ 1];
        gfrag_ptr wv = wv0;
        asm volatile("" : "+v"(wv));
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[a][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        typename El<EL>::v8 af[2], lb[LSTM8_LD];
        af[0] = *(const typename El<EL>::v8*)(hb);
        auto visit = [&](auto v_c) {
            constexpr int v = decltype(v_c)::value;
            constexpr int kk = v >> 3, a = (v >> 2) & 1, q = v & 3;
            if constexpr ((v & 7) == 0 && kk + 1 < 8) af[(kk + 1) & 1] = *(const typename El<EL>::v8*)(hb + (kk + 1) * 4 * 256);   // h slice kk+1, one slice ahead
            typename El<EL>::v8 w;
            if constexpr (v < R0) w = sb[v];
            else if constexpr (v < S1) w = wreg[v - R0];
            else if constexpr (v < L0) w = sb[v - S1];
            else w = lb[(v - L0) % LSTM8_LD];
            acc[a][q] = El<EL>::mfma(af[kk & 1], w, acc[a][q]);
            if constexpr (v < LSTM8_S1) sb[v] = (LSTM8_ABL & 2) ? wreg[v] : wv[(size_t)(S1 + v) * 64];      // the buffer slot is free again: late fragment v
            // LDS read-ahead: fragment v+LD goes into the ring slot fragment v just left (the first LD reads fill the empty ring)
            if constexpr (v + LSTM8_LD >= L0 && v + LSTM8_LD < 64) lb[(v + LSTM8_LD - L0) % LSTM8_LD] = (LSTM8_ABL & 4) ? wreg[(v + LSTM8_LD - L0) % LSTM8_NR] : wl[(size_t)(v + LSTM8_LD - L0) * 64];
            // hipcc sinks loads towards their use to save registers, which would put the L2 latency back in front of the MFMA:
            // nothing may be scheduled across the end of the refill block
            if constexpr (v == LSTM8_S1 - 1 || v >= L0 - LSTM8_LD - 1) __builtin_amdgcn_sched_barrier(0);   // (and the LDS read-ahead distance)
        };
        [&]<int... V>(std::integer_sequence<int, V...>) { (visit(std::integral_constant<int, V>{}), ...); }(std::make_integer_sequence<int, 64>{});
        stream_head();               // next step's first fragments travel while the gate math runs
        unsigned int pf = 0;         // issued AFTER the step's last weight loads: vmcnt retires in order, nothing in the MFMA phase may queue behind it
        if (pf_on && step + pf_dist < T) pf = *(const unsigned int*)(xproj + pf_row + (size_t)(dir ? t - pf_dist : t + pf_dist) * 2048);
        __builtin_amdgcn_sched_barrier(0);
        // gate math on PAIRS of sequences (v_pk_mul/add/fma_f32): the step is VALU-bound, and only the exp2/rcp stay scalar
        auto gate_group = [&](auto a_c) {
            constexpr int a = decltype(a_c)::value;
            const int unit = wave * 32 + a * 16 + u;
            auto ex2 = [](f32x2_t x) { return (f32x2_t){__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])}; };
            auto rcp2 = [](f32x2_t x) { return (f32x2_t){__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1])}; };
            const f32x2_t one = {1.f, 1.f}, nl = {-1.442695041f, -1.442695041f}, l2 = {2.885390082f, 2.885390082f}, m2 = {-2.f, -2.f};
            auto sig2 = [&](f32x2_t x) { if constexpr ((LSTM8_ABL & 1) != 0) return x * nl; else return rcp2(one + ex2(x * nl)); };
            auto tanh2 = [&](f32x2_t x) { if constexpr ((LSTM8_ABL & 1) != 0) return x * l2; else return __builtin_elementwise_fma(rcp2(one + ex2(x * l2)), m2, one); };   // 1 - 2/(1+e^{2x}), exact fma
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
                const unsigned int a0 = xq[rp][a * 2], a1 = xq[rp][a * 2 + 1], b0 = xq[rp + 1][a * 2], b1 = xq[rp + 1][a * 2 + 1];
                const f32x2_t xa0 = El<EL>::unpack2(a0), xa1 = El<EL>::unpack2(a1), xb0 = El<EL>::unpack2(b0), xb1 = El<EL>::unpack2(b1);   // (i, f), (g, o)
                const f32x2_t gi = (f32x2_t){acc[a][0][rp], acc[a][0][rp + 1]} + (f32x2_t){xa0[0], xb0[0]};
                const f32x2_t gf = (f32x2_t){acc[a][1][rp], acc[a][1][rp + 1]} + (f32x2_t){xa0[1], xb0[1]};
                const f32x2_t gg = (f32x2_t){acc[a][2][rp], acc[a][2][rp + 1]} + (f32x2_t){xa1[0], xb1[0]};
                const f32x2_t go = (f32x2_t){acc[a][3][rp], acc[a][3][rp + 1]} + (f32x2_t){xa1[1], xb1[1]};
                const f32x2_t cp = {c[a][rp], c[a][rp + 1]};
                const f32x2_t cn = sig2(gf) * cp + sig2(gi) * tanh2(gg);
                c[a][rp] = cn[0];
                c[a][rp + 1] = cn[1];
                const f32x2_t hv = sig2(go) * tanh2(cn);
                unsigned char* hp = hn + ((unit >> 3) * 16 + g * 4 + rp) * 16 + (unit & 7) * 2;
                *(unsigned short*)(hp) = El<EL>::from_f32(hv[0]);
                *(unsigned short*)(hp + 16) = El<EL>::from_f32(hv[1]);
            }
        };
        gate_group(std::integral_constant<int, 0>{});
        gate_group(std::integral_constant<int, 1>{});
        asm volatile("" ::"v"(pf));
        __syncthreads();
        if (wb_seq < n && !(LSTM8_ABL & 16)) {
            const u32x4 h0 = *(const u32x4*)(hn + (wb_kg * 16 + wb_seq) * 16);
            *(u32x4*)(out + ((size_t)row0 + (size_t)wb_seq * T + t) * 512 + dir * 256 + wb_kg * 8) = h0;
        }
        cur ^= 1;
    }
}

// ================================================================================================ exact mode (REC_SPLIT)
// The same recurrence with every operand at fp32-grade precision: h (kept in fp32 registers) crosses LDS as the fp16 pair
// hi = fp16(h), lo = fp16(h - hi); W_hh is the pair (w_hi, w_lo) of fp16 fragments of w * 2^s; per fragment the three MFMAs
// h_hi w_hi + h_hi w_lo + h_lo w_hi give  pre = acc * 2^-s + x  with x the FP32 input projection; gates with expf / tanhf and IEEE
// division, c and h in fp32; h leaves as the [hi | lo] pair the next split-fp16 GEMM reads.  Fragment layout of lstm8_kernel (8 waves,
// wave w owns units [32w, 32w + 32) x 4 gates, one direction per workgroup); all 128 weight fragments of a wave are re-streamed from L2
// every step through a short register ring.
size_t lstm_whh_split_packed_elems() { return 2 * lstm_whh_packed_elems(); }     // [hi image | lo image], each in pack_lstm_whh8's layout
float pack_lstm_whh_split(const float* whh_fwd, const float* whh_bwd, uint16_t* out) {
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)1024 * 256; ++i) { mx = fmaxf(mx, fabsf(whh_fwd[i])); mx = fmaxf(mx, fabsf(whh_bwd[i])); }
    int s = 0;
    if (mx > 0.f) s = 9 - (int)floor(log2((double)mx));          // 2^9 <= mx 2^s < 2^10
    s = s < -14 ? -14 : (s > 24 ? 24 : s);
    const float sc = ldexpf(1.f, s);
    const size_t half = lstm_whh_packed_elems();
    size_t o = 0;
    for (int d = 0; d < 2; ++d) {
        const float* W = d ? whh_bwd : whh_fwd;
        for (int w = 0; w < 8; ++w)
            for (int kk = 0; kk < 8; ++kk)
                for (int a = 0; a < 2; ++a)
                    for (int gate = 0; gate < 4; ++gate)
                        for (int l = 0; l < 64; ++l) {
                            const int unit = w * 32 + a * 16 + (l & 15);
                            const int row = gate * 256 + unit;
                            for (int j = 0; j < 8; ++j, ++o) {
                                const float v = W[(size_t)row * 256 + kk * 32 + 8 * (l >> 4) + j] * sc;
                                const uint16_t hi = f32_to_f16_host(v);
                                out[o] = hi;
                                out[half + o] = f32_to_f16_host(v - f16_to_f32_host(hi));
                            }
                        }
    }
    return ldexpf(1.f, -s);
}

// NG 16-sequence groups per workgroup share every streamed weight fragment pair; RING pairs (hi + lo) are in flight ahead of the MFMA
// stream (sched_barrier keeps hipcc from sinking the loads next to their use).  Measured on the 2,200 sequences of a 64-page step
// (T ~ 240, `tools/sb.sh x --precision exact`): NG = 1 / RING = 6 with the loads left to the compiler 6.7 ms per layer; NG = 1 with
// RING = 8 / 10 / 12 / 14 pinned in flight 5.7 / 5.7 / 5.8 / 5.6 (spills) ms -- the ring depth is not what bounds a step (~24 us: 1 MB
// of fragments per workgroup from L2, 192 MFMAs and ~5 us of expf / tanhf gate math per wave, one barrier); NG = 2 / RING = 5 (half
// the streamed bytes per sequence, a single round of 198 workgroups) 8.8 ms: the doubled MFMA + gate work per workgroup costs more
// than the halved stream saves.
#ifndef LSTMX_NG
#define LSTMX_NG 1
#endif
#ifndef LSTMX_RING
#define LSTMX_RING (LSTMX_NG == 1 ? 8 : 5)
#endif
#define LSTMX_SEQS (16 * LSTMX_NG)
// What bounds this kernel is the L2 -> CU stream of the weight fragments (1 MB per workgroup and time step), so one workgroup carries
// TWO 16-sequence groups through every fragment pair it loads (6 MFMAs per pair): half the streamed bytes per sequence, and the 2,200
// sequences of a 64-page step fit the 256 CUs in one round.  The lo half of h is kept UNSCALED here (fp16 keeps subnormals in
// v_mfma_f32_16x16x32_f16 -- tools/micro/mfma_f16_denorm.hip -- and |h| <= 1 bounds its absolute error by 3e-8), so the h_lo term
// shares the accumulator of the other two; the linear layer that reads the output pair is packed with lo scale 1 accordingly.
__global__ void __launch_bounds__(512, 1) lstm_exact_kernel(const float* __restrict__ xproj, const uint16_t* __restrict__ whh, uint16_t* __restrict__ out,
                                                            const int4* __restrict__ tiles, float acc_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lstmx_smem[];
    constexpr int NG = LSTMX_NG;
    // [parity 2][hi / lo][group NG][kgroup 32][seq 16] x 16 B = NG x 32 KB
    auto hb = [&](int par, int kind, int grp) { return lstmx_smem + (size_t)(((par * 2 + kind) * NG + grp) * 32 * 16 * 16); };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const int4 tile = tiles[blockIdx.x];
    const int row0 = tile.x, n = tile.y, T = tile.z;
    const int g = lane >> 4, u = lane & 15;
    for (int i = tid; i < 2 * 2 * NG * 32 * 16; i += 512) ((u32x4*)lstmx_smem)[i] = (u32x4){0u, 0u, 0u, 0u};
    float c[NG][2][4];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[grp][a][r] = 0.f;
    typedef const __attribute__((address_space(1))) f16x8* gfrag_ptr;
    const size_t half_frags = lstm_whh_packed_elems_dev() / 8;
    const gfrag_ptr whi = (gfrag_ptr)((const f16x8*)whh + ((size_t)(dir * 8 + wave) * 64) * 64 + lane);
    const gfrag_ptr wlo = whi + half_frags;
    const int xch = dir * 1024 + (wave * 16 + u) * 8;
    size_t xrow[NG][4];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sq = grp * 16 + g * 4 + r;
            xrow[grp][r] = ((size_t)row0 + (size_t)(sq < n ? sq : n - 1) * T) * 2048 + xch;
        }
    const int wb_seq = tid >> 5, wb_kg = tid & 31;   // h write-back: per group 16 B of hi and 16 B of lo per thread
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        f32x4 acc[NG][2][4];
#pragma unroll
        for (int grp = 0; grp < NG; ++grp)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[grp][a][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 x0[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* xp = xproj + xrow[0][r] + (size_t)t * 2048;
            x0[r][0] = *(const f32x4*)(xp);
            x0[r][1] = *(const f32x4*)(xp + 4);
        }
        gfrag_ptr ph = whi, pl = wlo;
        asm volatile("" : "+v"(ph), "+v"(pl));          // keep the weight loads inside the time loop
        f16x8 rh[LSTMX_RING], rl[LSTMX_RING];
#pragma unroll
        for (int i = 0; i < LSTMX_RING; ++i) { rh[i] = ph[(size_t)i * 64]; rl[i] = pl[(size_t)i * 64]; }
        __builtin_amdgcn_sched_barrier(0);              // hipcc otherwise sinks every load next to its use: the ring must really be in flight
        const int loff = (g * 16 + u) * 16;
        f16x8 ah[NG], al[NG];
#pragma unroll
        for (int v = 0; v < 64; ++v) {
            const int kk = v >> 3, a = (v >> 2) & 1, q = v & 3;
            if ((v & 7) == 0) {
#pragma unroll
                for (int grp = 0; grp < NG; ++grp) {
                    ah[grp] = *(const f16x8*)(hb(cur, 0, grp) + loff + kk * 4 * 256);
                    al[grp] = *(const f16x8*)(hb(cur, 1, grp) + loff + kk * 4 * 256);
                }
            }
            const f16x8 wh = rh[v % LSTMX_RING], wl = rl[v % LSTMX_RING];
            if (v + LSTMX_RING < 64) { rh[v % LSTMX_RING] = ph[(size_t)(v + LSTMX_RING) * 64]; rl[v % LSTMX_RING] = pl[(size_t)(v + LSTMX_RING) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                acc[grp][a][q] = El<1>::mfma(ah[grp], wh, acc[grp][a][q]);
                acc[grp][a][q] = El<1>::mfma(ah[grp], wl, acc[grp][a][q]);
                acc[grp][a][q] = El<1>::mfma(al[grp], wh, acc[grp][a][q]);
            }
        }
        // gate math, group by group: group 0's input projection was requested before the MFMA stream, group 1's travels while group 0's
        // transcendentals run (64 more live registers would not fit next to the accumulators of both groups)
        auto load_x = [&](int grp, f32x4 (&xv)[4][2]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* xp = xproj + xrow[grp][r] + (size_t)t * 2048;     // FP32 (i, f, g, o) of the lane's two units for this (sequence, step)
                xv[r][0] = *(const f32x4*)(xp);
                xv[r][1] = *(const f32x4*)(xp + 4);
            }
        };
        auto gates = [&](int grp, const f32x4 (&xv)[4][2]) {
            unsigned char* hnh = hb(cur ^ 1, 0, grp);
            unsigned char* hnl = hb(cur ^ 1, 1, grp);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int unit = wave * 32 + a * 16 + u;
                    auto pre = [&](int q) { return fmaf(acc[grp][a][q][r], acc_scale, xv[r][a][q]); };
                    const float gi = 1.0f / (1.0f + expf(-pre(0)));
                    const float gf = 1.0f / (1.0f + expf(-pre(1)));
                    const float gg = tanhf(pre(2));
                    const float go = 1.0f / (1.0f + expf(-pre(3)));
                    const float cn = gf * c[grp][a][r] + gi * gg;
                    c[grp][a][r] = cn;
                    const float hv = go * tanhf(cn);
                    const unsigned short hi = El<1>::from_f32(hv);
                    const unsigned short lo = El<1>::from_f32(hv - El<1>::to_f32(hi));
                    const int off = ((unit >> 3) * 16 + g * 4 + r) * 16 + (unit & 7) * 2;
                    *(unsigned short*)(hnh + off) = hi;
                    *(unsigned short*)(hnl + off) = lo;
                }
        };
        if constexpr (NG == 2) {
            f32x4 x1[4][2];
            load_x(NG - 1, x1);
            __builtin_amdgcn_sched_barrier(0);
            gates(0, x0);
            gates(NG - 1, x1);
        } else {
            gates(0, x0);
        }
        __syncthreads();
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int sq = grp * 16 + wb_seq;
            if (sq < n) {
                const size_t orow = ((size_t)row0 + (size_t)sq * T + t) * 1024 + dir * 256 + wb_kg * 8;
                *(u32x4*)(out + orow) = *(const u32x4*)(hb(cur ^ 1, 0, grp) + (wb_kg * 16 + wb_seq) * 16);
                *(u32x4*)(out + orow + 512) = *(const u32x4*)(hb(cur ^ 1, 1, grp) + (wb_kg * 16 + wb_seq) * 16);
            }
        }
        cur ^= 1;
    }
}
int lstm_tile_seqs(int mode) { return mode == REC_SPLIT ? LSTMX_SEQS : 16; }

hipError_t launch_lstm(const void* xproj, const uint16_t* whh_pk, uint16_t* out, const int* tiles_dev, int ntiles, int mode, float acc_scale,
                       hipStream_t s) {
    if (ntiles <= 0) return hipSuccess;
    if (mode == REC_SPLIT) {
        const size_t smemx = (size_t)2 * 2 * LSTMX_NG * 32 * 16 * 16;
        static LdsOptIn attrx;
        if (hipError_t e = lds_opt_in(attrx, (const void*)lstm_exact_kernel, smemx); e != hipSuccess) return e;
        hipLaunchKernelGGL(lstm_exact_kernel, dim3(ntiles, 2), dim3(512), smemx, s, (const float*)xproj, whh_pk, out, (const int4*)tiles_dev, acc_scale);
        return hipGetLastError();
    }
    const size_t smem8 = 2 * 32 * 16 * 16 + (size_t)8 * LSTM8_NL * 1024;
    static const int pf_dist = diag_knob("BBOCR_LSTM_PF", 2);   // x prefetch distance in steps (0 = off)
    auto go = [&](auto kern, LdsOptIn& attr) -> hipError_t {
        if (hipError_t e = lds_opt_in(attr, (const void*)kern, smem8); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(ntiles, 2), dim3(512), smem8, s, (const uint16_t*)xproj, whh_pk, out, (const int4*)tiles_dev, pf_dist);
        return hipGetLastError();
    };
    static LdsOptIn attr_bf, attr_f16;
    return mode == REC_F16 ? go(lstm8_kernel<1>, attr_f16) : go(lstm8_kernel<0>, attr_bf);
}
