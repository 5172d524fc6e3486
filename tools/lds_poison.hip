// Fills every CU's LDS with a NaN pattern (0x7FFF7FFF: NaN in f16, bf16 and as float32): after it, a kernel that reads LDS
// it has not written -- or reads a DMA-staged tile before it landed -- produces NaNs instead of plausible stale values.
// tools only: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_poison.hip -o tools/bin/liblds_poison.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void lds_poison_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds[];
    const int words = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (sink && lds[(threadIdx.x * 977u) % words] == 1u) sink[0] = 1u;   // keep the stores alive
}
extern "C" int lds_poison(unsigned pattern, int blocks, void* sink, void* stream) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        once = true;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(blocks), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, (unsigned*)sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
