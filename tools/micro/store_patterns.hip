// Microbenchmark (diagnostic, not part of the library): cost of an epilogue-like burst of global_store_dwordx4 by address
// pattern.  4 waves per workgroup, 16 stores per lane, 2 workgroups per CU resident, many bursts per workgroup.
//   P0: 16 pixels x 64 B per instruction (pixel stride 256 B)        -- conv epilogue today (C = 128, wave owns 64 couts)
//   P1:  8 pixels x 128 B per instruction (full lines)
//   P2: 1 KB contiguous per instruction
//   P3: 16 pixels x 4 x 16 B with 16-B gaps                          -- conv epilogue before the cout re-mapping
//   P4: 64 pixels x 16 B (row-per-lane, stride 256 B)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int P>
__global__ void __launch_bounds__(256, 2) burst(unsigned char* out, size_t tile_bytes, int tiles_per_wg, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, g = lane >> 4;
    unsigned long long t_acc = 0;
    u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    for (int t = 0; t < tiles_per_wg; ++t) {
        // tile: 256 pixels x 128 couts bf16 = 64 KB; image row of 640 pixels x 256 B; tile 16 x 16 pixels
        const size_t tile = (size_t)blockIdx.x * tiles_per_wg + t;
        unsigned char* base = out + (tile % 4096) * tile_bytes;
        const int wm = wave >> 1, wn = wave & 1;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int row = wm * 8 + f;          // tile row (16 pixels of 256 B each, rows 640*256 B apart in a real image)
            unsigned char* rp = base + (size_t)row * 4096 + wn * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                size_t off;
                if (P == 0) off = (size_t)pl * 256 + s * 64 + g * 16;
                else if (P == 1) off = (size_t)(s * 8 + (lane >> 3)) * 256 + (lane & 7) * 16;
                else if (P == 2) off = (size_t)(f * 2 + s) * 1024 + lane * 16 - (size_t)row * 4096 + (size_t)wm * 16384 - wn * 128 + wn * 32768;
                else if (P == 3) off = (size_t)pl * 256 + g * 32 + s * 16;
                else off = (size_t)lane * 256 + s * 16 + (size_t)f * 32 - (size_t)row * 4096 - wn * 128 + (size_t)wave * 16384;
                v[1] += f;
                *(u32x4*)(rp + off) = v;
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        t_acc += t1 - t0;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t_acc / tiles_per_wg;
}

template <int P>
static void run(unsigned char* out, unsigned long long* cyc, int grid, int tiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    burst<P><<<grid, 256>>>(out, 65536, tiles, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    burst<P><<<grid, 256>>>(out, 65536, tiles, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : h) s += (double)x;
    printf("P%d: %.1f us for %d bursts/WG -> %.2f us per burst; issue %0.f cycles per burst (64 store instrs per WG); %.1f GB/s\n", P, ms * 1e3, tiles,
           ms * 1e3 / tiles, s / grid, (double)grid * tiles * 65536 / (ms * 1e-3) / 1e9);
}

int main() {
    unsigned char* out; unsigned long long* cyc;
    hipMalloc((void**)&out, (size_t)4096 * 65536 + (1 << 20));
    hipMalloc((void**)&cyc, 4096 * 8);
    const int grid = 512, tiles = 64;
    run<0>(out, cyc, grid, tiles);
    run<1>(out, cyc, grid, tiles);
    run<2>(out, cyc, grid, tiles);
    run<3>(out, cyc, grid, tiles);
    run<4>(out, cyc, grid, tiles);
    return 0;
}
