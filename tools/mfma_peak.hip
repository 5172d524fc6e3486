// Practical fp32-MFMA peak: waves issuing nothing but v_mfma_f32_32x32x2_f32 on register operands.
// usage: mfma_peak [waves_per_simd] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    f32x16 s = c0 + c1 + c2 + c3;
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    if (r == 123.456f) out[0] = r;
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 20000;
    float* out; hipMalloc(&out, 4);
    int blocks = 256 * wps;          // 256 CUs x (wps workgroups of 4 waves -> wps waves per SIMD)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<<<blocks, 256>>>(out, 1000, 1.0f, 0.0f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<<<blocks, 256>>>(out, iters, 1.0f, 0.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
        printf("waves/SIMD %d  iters %d  %.3f ms  %.1f TFLOP/s (fp32 32x32x2)\n", wps, iters, ms, flops / ms / 1e9);
    }
    return 0;
}
