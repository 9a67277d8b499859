// Does v_mfma_f32_32x32x16_f16 on gfx950 keep fp16 subnormal INPUTS (or flush them to zero)?  Decides whether the low
// halves of split-half operands may be stored at their true scale (single accumulator) -- see csrc/sh.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float* out) {
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float as[] = {1.0f, 0x1p-14f, 0x1p-15f, 0x1p-20f, 0x1p-24f, 3 * 0x1p-24f};
    const float bs[] = {1.0f, 1024.0f, 0x1p-15f};
    for (float a : as) for (float b : bs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  mfma=%g  exact=%g  %s\n", a, b, h, 16.0 * (double)(float)(_Float16)a * (double)(float)(_Float16)b,
               h == (float)(16.0 * (double)(float)(_Float16)a * (double)(float)(_Float16)b) ? "KEPT" : "FLUSHED/DIFF");
    }
    return 0;
}
