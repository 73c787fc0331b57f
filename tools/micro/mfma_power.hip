// Sustained dense bf16 MFMA rate under the board's power limit: v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16, operands in
// registers only (no LDS / memory in the loop), 128 accumulator registers per wave as in conv3x3_dma_kernel, on random / half-zero /
// all-zero data.  Decides whether re-tiling the detector trunk for 32x32 MFMAs could buy clock at the power limit.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/mfma_power tools/micro/mfma_power.hip && tools/micro/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// per iteration: two operand sets alternate (so that operand buses toggle like a k-loop reading new fragments every step)
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k16(const bf16x8* __restrict__ src, float* out, int iters) {
    bf16x8 a[2][4], b[2][8];
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < 4; ++i) a[s][i] = src[((s * 12 + i) * 64 + lane)];
        for (int i = 0; i < 8; ++i) b[s][i] = src[((s * 12 + 4 + i) * 64 + lane)];
    }
    f32x4 acc[8][4];
    for (int f = 0; f < 8; ++f)
        for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int f = 0; f < 8; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][j], b[s][f], acc[f][j], 0, 0, 0);
    }
    float t = 0.f;
    for (int f = 0; f < 8; ++f)
        for (int j = 0; j < 4; ++j) t += acc[f][j][0] + acc[f][j][3];
    if (t == 123.456f) out[0] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // core clocks (s_memtime) against the constant 100 MHz counter (s_memrealtime)
        ((unsigned long long*)out)[1] = __builtin_amdgcn_s_memtime() - t0;
        ((unsigned long long*)out)[2] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// same FLOPs per iteration: 2 sets x (2 A fragments x 4 B fragments) x 32x32x16, two k-halves
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k32(const bf16x8* __restrict__ src, float* out, int iters) {
    bf16x8 a[2][2][2], b[2][2][4];
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < 2; ++s)
        for (int h = 0; h < 2; ++h) {
            for (int i = 0; i < 2; ++i) a[s][h][i] = src[((s * 12 + h * 6 + i) * 64 + lane)];
            for (int i = 0; i < 4; ++i) b[s][h][i] = src[((s * 12 + h * 6 + 2 + i) * 64 + lane)];
        }
    f32x16 acc[4][2];
    for (int f = 0; f < 4; ++f)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[f][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][h][j], b[s][h][f], acc[f][j], 0, 0, 0);
    }
    float t = 0.f;
    for (int f = 0; f < 4; ++f)
        for (int j = 0; j < 2; ++j) t += acc[f][j][0] + acc[f][j][15];
    if (t == 123.456f) out[0] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // core clocks (s_memtime) against the constant 100 MHz counter (s_memrealtime)
        ((unsigned long long*)out)[1] = __builtin_amdgcn_s_memtime() - t0;
        ((unsigned long long*)out)[2] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static uint16_t bf16_of(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int main() {
    const int n = 24 * 64 * 8;
    std::vector<uint16_t> h(n);
    bf16x8* d;
    float* out;
    hipMalloc(&d, n * 2);
    hipMalloc(&out, 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[3] = {"random N(0,1)", "half zeros (post-ReLU like)", "all zeros"};
    for (int wps = 1; wps <= 2; ++wps)
        for (int data = 0; data < 3; ++data) {
            srand(1);
            for (int i = 0; i < n; ++i) {
                float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
                float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
                if (data == 1 && g < 0.f) g = 0.f;
                if (data == 2) g = 0.f;
                h[i] = bf16_of(g);
            }
            hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
            for (int variant = 0; variant < 2; ++variant) {
                const int iters = 4000, grid = 256 * wps * 8;          // 8 rounds of workgroups per launch
                const double flop_per_launch = (double)grid * 4 /*waves*/ * iters * 64 /*MFMA 16x16x32 equivalents*/ * 16384.0;
                auto launch = [&]() {
                    if (variant == 0) { if (wps == 1) hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(256), 0, 0, d, out, iters); else hipLaunchKernelGGL(k16<2>, dim3(grid), dim3(256), 0, 0, d, out, iters); }
                    else { if (wps == 1) hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, d, out, iters); else hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, d, out, iters); }
                };
                launch();
                hipDeviceSynchronize();
                // ~0.5 s of back-to-back launches so that the power controller settles; report the second half
                float ms = 0.f;
                int reps = 0;
                double best_window = 0.0;
                for (int w = 0; w < 4; ++w) {
                    hipEventRecord(e0, 0);
                    int r = 0;
                    for (; r < 6; ++r) launch();
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                    best_window = flop_per_launch * r / (ms * 1e-3) / 1e12;
                    reps += r;
                    unsigned long long hc[4];
                    hipMemcpy(hc, out, 32, hipMemcpyDeviceToHost);
                    const double mhz = (double)hc[1] / ((double)hc[2] / 100.0);
                    printf("  waves/SIMD %d  %-28s %s  window %d: %7.1f TFLOP/s (%.1f ms)  s_memtime/s_memrealtime -> %.0f MHz, %.1f memtime ticks per MFMA-16x16x32-equivalent of a wave\n",
                           wps, names[data], variant ? "32x32x16" : "16x16x32", w, best_window, ms, mhz, (double)hc[1] / (iters * 64.0));
                }
            }
        }
    return 0;
}
