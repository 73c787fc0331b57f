// Diagnostic harness: conv3x3_up4_kernel (launch_up4_fused) against the two-launch path (conv1x1<ADDUP> + 3x3) on random fp16 data, and against
// ITSELF run to run.  Build on a GPU box:  hipcc -O2 -std=c++20 --offload-arch=gfx950 -Ibb-ocr_amd/csrc -Iinclude tools/micro/up4_check.hip -Lbb-ocr_amd -lbbocr -o /tmp/up4_check
#include "kernels.h"
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

static ConvPlan plan(int Cin, int Cout, int K, int pad, const std::vector<float>& w, const std::vector<float>& b) {
    ConvPlan p;
    p.el = 1; p.Cin = Cin; p.Cout = Cout; p.KH = p.KW = K; p.pad_h = p.pad_w = pad; p.dil = 1;
    p.Cin_pad = (Cin + 31) / 32 * 32; p.BN = 64; p.Cout_pad = 64;
    std::vector<uint16_t> pk(conv_packed_elems(p));
    pack_conv_weights(p, w.data(), pk.data());
    std::vector<float> bp(64, 0.f);
    memcpy(bp.data(), b.data(), b.size() * 4);
    CK(hipMalloc((void**)&p.d_w, pk.size() * 2)); CK(hipMemcpy(p.d_w, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&p.d_b, 64 * 4)); CK(hipMemcpy(p.d_b, bp.data(), 64 * 4, hipMemcpyHostToDevice));
    return p;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4, H = argc > 2 ? atoi(argv[2]) : 240, W = argc > 3 ? atoi(argv[3]) : 320, reps = 8;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto rnd16 = [&](size_t n, float sc) { std::vector<uint16_t> v(n); for (auto& x : v) x = f32_to_f16_host(nd(rng) * sc); return v; };
    std::vector<float> w1(64 * 128), b1(64), w3(32 * 64 * 9), b3(32);
    for (auto& x : w1) x = nd(rng) / 11.3f;
    for (auto& x : b1) x = nd(rng) * 0.1f;
    for (auto& x : w3) x = nd(rng) / 24.f;
    for (auto& x : b3) x = nd(rng) * 0.1f;
    ConvPlan p1 = plan(128, 64, 1, 0, w1, b1), p3 = plan(64, 32, 3, 1, w3, b3);
    const size_t ns1 = (size_t)N * H * W * 128, nz = (size_t)N * (H / 2) * (W / 2) * 64, nu = (size_t)N * H * W * 32, nu4a = (size_t)N * H * W * 64;
    std::vector<uint16_t> hs1 = rnd16(ns1, 1.f), hz = rnd16(nz, 1.f);
    uint16_t *s1, *z, *out, *u4a, *ref;
    void* zero;
    // ONE allocation in arena order (... z | out ...), so that an out-of-range read of z lands in `out` exactly as in the library
    char* arena;
    CK(hipMalloc((void**)&arena, (ns1 + nz + nu + nu4a + nu) * 2 + 4096));
    s1 = (uint16_t*)arena; z = s1 + ns1; out = z + nz; u4a = out + nu; ref = u4a + nu4a;
    CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
    CK(hipMemcpy(s1, hs1.data(), ns1 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(z, hz.data(), nz * 2, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto args = [&]() { ConvArgs a{}; a.in0 = s1; a.C0 = 128; a.in0_cs = 128; a.N = N; a.H = H; a.W = W; a.addup = z; a.up_H = H; a.up_W = W; a.up_cs = 64; a.relu_out = 1; a.zero = zero; return a; };
    // reference: two launches
    { ConvArgs a = args(); a.out = u4a; a.out_cs = 64; a.cout_store = 64; CK(launch_conv(p1, a, st)); }
    { ConvArgs a{}; a.in0 = u4a; a.C0 = 64; a.in0_cs = 64; a.N = N; a.H = H; a.W = W; a.relu_out = 1; a.out = ref; a.out_cs = 32; a.cout_store = 32; a.zero = zero; CK(launch_conv(p3, a, st)); }
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> href(nu), h0(nu), h(nu);
    CK(hipMemcpy(href.data(), ref, nu * 2, hipMemcpyDeviceToHost));
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(out, 0xff, nu * 2, st));
        ConvArgs a = args(); a.out = out; a.out_cs = 32; a.cout_store = 32;
        hipError_t e = launch_up4_fused(p1, p3, a, st);
        if (e != hipSuccess) { printf("launch_up4_fused: %s\n", hipGetErrorString(e)); return 2; }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), out, nu * 2, hipMemcpyDeviceToHost));
        if (r == 0) h0 = h;
        size_t dref = 0, drun = 0; long first = -1;
        for (size_t i = 0; i < nu; ++i) { dref += h[i] != href[i]; if (h[i] != h0[i]) { if (first < 0) first = (long)i; ++drun; } }
        printf("run %d: %zu values differ from the two-launch path, %zu from run 0", r, dref, drun);
        if (first >= 0) { size_t px = first / 32; printf("  (first: n=%zu y=%zu x=%zu c=%zu)", px / ((size_t)H * W), (px / W) % H, px % W, (size_t)first % 32); }
        printf("\n");
        if (r == reps - 1 || drun) {      // where do fused and reference differ?
            size_t shown = 0;
            for (size_t i = 0; i < nu && shown < 12; ++i) if (h[i] != href[i]) { size_t px = i / 32; printf("   vs ref: n=%zu y=%zu x=%zu c=%zu  fused %g  ref %g\n", px / ((size_t)H * W), (px / W) % H, px % W, i % 32, f16_to_f32_host(h[i]), f16_to_f32_host(href[i])); ++shown; }
        }
    }
    return 0;
}
