// Greedy CTC decode on wavefront primitives: row softmax + arg-max, then run-length collapse + confidence product.
//
// Restates easyocr/recognition.py::recognizer_predict (F.softmax, preds_prob.max / argmax, values[indices != 0]) and
// easyocr/utils.py::CTCLabelConverter.decode_greedy (drop repeats, drop blank=0); custom_mean's final
// prod ** (2/sqrt(n)) is finished on the host in double.  Reference call site
// pipeline_demo/extractor/enhanced_extractor.py:520 (decoder defaults to 'greedy').
#include "common.h"
#include "kernels.h"

// one wave per logits row (C <= 128): lanes hold classes lane and lane+64
// ignore: recognizer_predict's ignore_idx (allowlist / blocklist) as a 128-bit class mask: `preds_prob[:, :, ignore_idx] = 0`, then
// the division by the remaining sum that upstream always performs
__global__ void __launch_bounds__(256) ctc_rows_kernel(const float* __restrict__ logits, size_t rows, int C, int cs, int* __restrict__ idx,
                                                       float* __restrict__ pmax, uint4 ignore, float* __restrict__ probs) {
    const int lane = threadIdx.x & 63;
    const size_t wave0 = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    for (size_t row = wave0; row < rows; row += nwaves) {
        const float* p = logits + row * cs;
        const float v0 = lane < C ? p[lane] : -INFINITY;
        const float v1 = lane + 64 < C ? p[lane + 64] : -INFINITY;
        float m = fmaxf(v0, v1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e0 = lane < C ? expf(v0 - m) : 0.f;
        const float e1 = lane + 64 < C ? expf(v1 - m) : 0.f;
        float sum = e0 + e1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        float q0 = e0 / sum, q1 = e1 / sum;
        const unsigned int w0 = lane < 32 ? ignore.x : ignore.y, w1 = lane < 32 ? ignore.z : ignore.w;
        if ((w0 >> (lane & 31)) & 1u) q0 = 0.f;
        if ((w1 >> (lane & 31)) & 1u) q1 = 0.f;
        float norm = q0 + q1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) norm += __shfl_xor(norm, o);
        q0 = q0 / norm;
        q1 = q1 / norm;
        if (probs) {   // decoder='beamsearch': the host search reads the whole distribution (row stride cs)
            if (lane < C) probs[row * cs + lane] = q0;
            if (lane + 64 < C) probs[row * cs + lane + 64] = q1;
        }
        // arg-max over probabilities, first index wins ties (numpy argmax)
        float bp = q0;
        int bi = lane;
        if (q1 > bp) { bp = q1; bi = lane + 64; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float op = __shfl_xor(bp, o);
            const int oi = __shfl_xor(bi, o);
            if (op > bp || (op == bp && oi < bi)) { bp = op; bi = oi; }
        }
        if (lane == 0) { idx[row] = bi; pmax[row] = bp; }
    }
}

// one wave per sequence: collapse repeats / blanks with ballot + prefix popcount; confidence product kept strictly
// sequential in t (float32), as numpy's multiply.reduce does.
// seqs[i] = {first row, T}: sequences of different lengths (buckets) share one launch
__global__ void __launch_bounds__(64) ctc_collapse_kernel(const int* __restrict__ idx, const float* __restrict__ pmax,
                                                          const int2* __restrict__ seqs, int* __restrict__ out_idx,
                                                          CtcOut* __restrict__ out) {
    const int seq = blockIdx.x;
    const int lane = threadIdx.x;
    const int2 sd = seqs[seq];
    const int T = sd.y;
    const int* ip = idx + (size_t)sd.x;
    const float* pp = pmax + (size_t)sd.x;
    int* op = out_idx + (size_t)sd.x;
    int len = 0, cnt = 0;
    float prod = 1.f;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int cur = t < T ? ip[t] : 0;
        const int prev = (t > 0 && t < T) ? ip[t - 1] : -1;
        const float pv = t < T ? pp[t] : 1.f;
        const bool nb = (t < T) && cur != 0;
        const bool keep = nb && (t == 0 || cur != prev);
        const unsigned long long km = __ballot(keep);
        if (keep) op[len + __popcll(km & ((1ULL << lane) - 1ULL))] = cur;
        len += __popcll(km);
        unsigned long long nm = __ballot(nb);
        cnt += __popcll(nm);
        while (nm) {   // wave-uniform loop over the non-blank steps in time order
            const int j = __ffsll((long long)nm) - 1;
            nm &= nm - 1;
            prod = prod * __shfl(pv, j);
        }
    }
    if (lane == 0) {
        CtcOut o;
        o.len = len;
        o.cnt = cnt;
        o.prod = prod;
        o.pad = 0;
        out[seq] = o;
    }
}

hipError_t launch_ctc(const float* logits, size_t rows, int C, int cs, const int* seqs_dev, int nseq, int* idx_tmp, float* pmax_tmp,
                      int* out_idx, CtcOut* out, hipStream_t s, const unsigned int* ignore, float* probs_out) {
    if (nseq <= 0 || rows == 0) return hipSuccess;
    if (C > 128) return hipErrorInvalidValue;
    const size_t blocks = (rows + 3) / 4;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(ctc_rows_kernel, dim3(grid), dim3(256), 0, s, logits, rows, C, cs, idx_tmp, pmax_tmp,
                       make_uint4(ignore ? ignore[0] : 0u, ignore ? ignore[1] : 0u, ignore ? ignore[2] : 0u, ignore ? ignore[3] : 0u), probs_out);
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(nseq), dim3(64), 0, s, idx_tmp, pmax_tmp, (const int2*)seqs_dev, out_idx, out);
    return hipGetLastError();
}
