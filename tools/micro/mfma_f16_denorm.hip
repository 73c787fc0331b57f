// Does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs?  (decides whether the lo halves of split-fp16 operands need the x2048 scale)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float tests[][2] = {{1.0f, 1.0f}, {3e-6f, 1024.f}, {3e-6f, 3e-6f}, {6e-8f, 1024.f}, {5e-5f, 1.0f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const double ea = (double)(float)(_Float16)t[0], eb = (double)(float)(_Float16)t[1];
        printf("a=%g (fp16 %g) b=%g: mfma sum of 32 products = %g, expected %g\n", t[0], ea, t[1], h, 32.0 * ea * eb);
    }
    return 0;
}
