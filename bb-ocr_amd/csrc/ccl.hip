// Region/affinity heat-map -> connected components (4-connectivity) with the statistics getDetBoxes_core needs.
//
// Restates the pixel-level part of easyocr/craft_utils.py::getDetBoxes_core (cv2.threshold, np.clip(text+link),
// cv2.connectedComponentsWithStats(connectivity=4), per-label area / bbox / max(textmap), link-only pixel removal) as
// label-equivalence union-find on the GPU.  The per-component geometry that follows (rect dilation, minAreaRect) only
// depends on each component's per-row x-extremes of TEXT pixels, which is what this file emits; boxpost.cpp finishes
// on the host.  A component's id is its smallest pixel index == OpenCV's raster-order label order.
#include "common.h"
#include "kernels.h"

__device__ __forceinline__ int ccl_find(const int* __restrict__ label, int x) {
    int p = __hip_atomic_load(label + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(label + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

__device__ __forceinline__ void ccl_union(int* label, int a, int b) {
    for (;;) {
        a = ccl_find(label, a);
        b = ccl_find(label, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(label + b, a);   // b was a root iff old == b
        if (old == b) return;
        b = old;
    }
}

// Initial labels: every foreground pixel starts as the FIRST pixel of its horizontal run inside its 64-pixel wave segment (ballot +
// count-leading-zeros, no memory traffic), not as itself: a run is then already one set, and the merge kernel only has to unite runs
// where they first touch -- an order of magnitude fewer union-find walks than two per foreground pixel.  (i & 63) == lane because the
// grid stride is a multiple of 256.
__global__ void __launch_bounds__(256) ccl_init_kernel(const float* __restrict__ heat, size_t total, int w, float low_text, float link_thr,
                                                       int* __restrict__ label, int* __restrict__ stat, int* __restrict__ slot) {
    const int lane = threadIdx.x & 63;
    const size_t nround = (total + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);
    for (size_t it = 0; it < nround; ++it) {
        const size_t i = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        bool fg = false;
        int x = 0;
        if (i < total) {
            const float2 v = *(const float2*)(heat + i * 2);
            fg = (v.x > low_text) || (v.y > link_thr);
            x = (int)(i % w);
        }
        const unsigned long long fgm = __ballot(fg);
        // a run starts at lane 0, at a row start, or after a background lane
        const bool start = fg && (lane == 0 || x == 0 || !((fgm >> (lane - 1)) & 1ULL));
        const unsigned long long sm = __ballot(start);
        if (i < total) {
            int lab = -1;
            if (fg) {
                const unsigned long long below = sm & ((lane == 63) ? ~0ULL : ((1ULL << (lane + 1)) - 1ULL));
                lab = (int)i - (lane - (63 - __clzll((long long)below)));
            }
            label[i] = lab;
            slot[i] = -1;
            int* st = stat + i * 6;
            st[0] = 0x7fffffff; st[1] = -1; st[2] = 0x7fffffff; st[3] = -1; st[4] = 0; st[5] = 0;
        }
    }
}

__global__ void __launch_bounds__(256) ccl_merge_kernel(size_t total, int h, int w, int* label) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        if (label[i] < 0) continue;
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const bool left = x > 0 && label[i - 1] >= 0;
        // horizontal: only where a run continues across a wave-segment boundary (inside a segment the run is one set already)
        if (left && (i & 63) == 0) ccl_union(label, (int)i, (int)i - 1);
        // vertical: only at the first contact of the two runs -- if my left neighbour and ITS upper neighbour are both foreground, that
        // pixel (or one further left) has united the same two runs
        if (y > 0 && label[i - w] >= 0 && !(left && label[i - w - 1] >= 0)) ccl_union(label, (int)i, (int)i - w);
    }
}

// Flatten + per-component statistics.  A wave covers 64 consecutive pixels of the flattened image; lanes that continue the
// previous lane's component in the same row form a run, and only the run's first lane touches the component's counters
// (area = run length, x extremes = run ends, max text score by a segmented shuffle reduction): ~10x fewer atomics than
// one set per foreground pixel.
__global__ void __launch_bounds__(256) ccl_stats_kernel(const float* __restrict__ heat, size_t total, int h, int w, int* label,
                                                        int* __restrict__ stat) {
    const int lane = threadIdx.x & 63;
    const size_t nround = (total + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);
    for (size_t it = 0; it < nround; ++it) {
        const size_t i = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        int r = -1, x = 0, y = 0;
        float t = 0.f;
        if (i < total && label[i] >= 0) {
            r = ccl_find(label, (int)i);
            label[i] = r;
            x = (int)(i % w);
            y = (int)((i / w) % h);
            t = fmaxf(heat[i * 2], 0.f);
        }
        const int rp = __shfl_up(r, 1);
        const bool head = r >= 0 && (lane == 0 || rp != r || x == 0);      // a run never crosses a row (or image) boundary
        const unsigned long long hm = __ballot(head || r < 0);              // run breaks: heads and background lanes
        if (r >= 0) {
            // run = [first break at or below lane, next break above lane)
            const unsigned long long below = hm & ((lane == 63) ? ~0ULL : ((1ULL << (lane + 1)) - 1ULL));
            const int first = 63 - __clzll((long long)below);
            const unsigned long long above = (lane == 63) ? 0ULL : (hm >> (lane + 1));
            const int len_after = above ? (__ffsll((long long)above) - 1) : (63 - lane);   // lanes of this run after me
            // segmented max of the text score over the run (suffix max towards the head)
            float m = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float mo = __shfl_down(m, o);
                if (o <= len_after) m = fmaxf(m, mo);
            }
            if (head) {
                int* st = stat + (size_t)r * 6;
                atomicMin(st + 0, x);
                atomicMax(st + 1, x + len_after);
                atomicMin(st + 2, y);
                atomicMax(st + 3, y);
                atomicAdd(st + 4, len_after + 1);
                if (m > 0.f) atomicMax(st + 5, __float_as_int(m));
            }
            (void)first;
        }
    }
}

__global__ void __launch_bounds__(256) ccl_accept_kernel(size_t total, int hw, double text_thr, const int* __restrict__ label,
                                                         const int* __restrict__ stat, int* __restrict__ slot, CclOut* comps,
                                                         int* rowext, int* counters, int cap_comps, int cap_rows) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        if (label[i] != (int)i) continue;
        const int* st = stat + i * 6;
        const int area = st[4];
        if (area < 10) continue;
        // upstream compares the float32 maximum with the Python float 0.7 in double precision
        if ((double)__int_as_float(st[5]) < text_thr) continue;
        const int img = (int)(i / hw);
        const int hh = st[3] - st[2] + 1;
        const int idx = atomicAdd(counters + 0, 1);          // one compact list for the whole batch (single D2H copy)
        const int off = atomicAdd(counters + 1, hh);
        if (idx >= cap_comps || off + hh > cap_rows) { atomicOr(counters + 2, 1); continue; }
        CclOut c;
        c.root = (int)(i - (size_t)img * hw);
        c.left = st[0]; c.top = st[2]; c.right = st[1]; c.bottom = st[3]; c.area = area; c.row_off = off; c.img = img;
        comps[idx] = c;
        slot[i] = idx;
        int* re = rowext + (size_t)off * 2;
        for (int k = 0; k < hh; ++k) { re[2 * k] = 0x7fffffff; re[2 * k + 1] = -1; }
    }
}

// per-row x-extremes of the TEXT pixels of every accepted component.  Like ccl_stats_kernel, lanes that continue the previous lane's
// component in the same row form a run and only the run's first lane touches the row's two counters (min = run start, max = run end).
__global__ void __launch_bounds__(256) ccl_rowext_kernel(const float* __restrict__ heat, size_t total, int h, int w, float low_text,
                                                         const int* __restrict__ label, const int* __restrict__ slot,
                                                         const CclOut* __restrict__ comps, int* rowext, int cap_comps, int cap_rows) {
    const int lane = threadIdx.x & 63;
    const size_t nround = (total + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);
    for (size_t it = 0; it < nround; ++it) {
        const size_t i = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        int r = -1, x = 0;
        if (i < total) {
            r = label[i];
            if (r >= 0 && !(heat[i * 2] > low_text)) r = -1;   // link-only pixels are removed from the segmentation map
            x = (int)(i % w);
        }
        const int rp = __shfl_up(r, 1);
        const bool head = r >= 0 && (lane == 0 || rp != r || x == 0);
        const unsigned long long hm = __ballot(head || r < 0);
        if (head) {
            const int s = slot[r];
            if (s >= 0) {
                const unsigned long long above = (lane == 63) ? 0ULL : (hm >> (lane + 1));
                const int len_after = above ? (__ffsll((long long)above) - 1) : (63 - lane);   // lanes of this run after me
                const CclOut c = comps[s];
                const int y = (int)((i / w) % h);
                int* re = rowext + ((size_t)c.row_off + (y - c.top)) * 2;
                atomicMin(re, x);
                atomicMax(re + 1, x + len_after);
            }
        }
    }
}

hipError_t launch_ccl(const float* heat, int N, int h, int w, float low_text, float link_thr, double text_thr, int* label, int* stat,
                      int* slot, CclOut* comps, int* rowext, int* counters, int cap_comps, int cap_rows, hipStream_t s) {
    const size_t total = (size_t)N * h * w;
    if (total == 0) return hipSuccess;
    if (total > 0x7fffffffULL) return hipErrorInvalidValue;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipError_t e = hipMemsetAsync(counters, 0, sizeof(int) * 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ccl_init_kernel, dim3(grid), dim3(256), 0, s, heat, total, w, low_text, link_thr, label, stat, slot);
    hipLaunchKernelGGL(ccl_merge_kernel, dim3(grid), dim3(256), 0, s, total, h, w, label);
    hipLaunchKernelGGL(ccl_stats_kernel, dim3(grid), dim3(256), 0, s, heat, total, h, w, label, stat);
    hipLaunchKernelGGL(ccl_accept_kernel, dim3(grid), dim3(256), 0, s, total, h * w, text_thr, label, stat, slot, comps, rowext, counters,
                       cap_comps, cap_rows);
    hipLaunchKernelGGL(ccl_rowext_kernel, dim3(grid), dim3(256), 0, s, heat, total, h, w, low_text, label, slot, comps, rowext, cap_comps,
                       cap_rows);
    return hipGetLastError();
}
