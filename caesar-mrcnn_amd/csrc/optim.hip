// Optimiser step of MaskRCNN.compile (mrcnn/model.py:2255-2291) on flat fp32 buffers:
//   * L2 regulariser keras.regularizers.l2(WEIGHT_DECAY)(w) / size(w) for every trainable weight whose
//     name lacks gamma/beta: gradient 2*WEIGHT_DECAY*w/size(w), folded in as a per-segment coefficient;
//   * keras.optimizers.SGD(lr, momentum, clipnorm): [3P Keras 2.2.4] global-norm clipping
//     g *= clipnorm / max(norm, clipnorm), then v = momentum*v - lr*g, w += v.
// HBM-bound: 3 reads + 2 writes per parameter; 16-byte accesses, grid-stride.
#include "common.h"

// Global squared norm in two fixed-order stages (per-workgroup partials, then one workgroup sums them in
// index order): every data-parallel rank must derive bit-identical clip factors from identical gradients,
// so no float atomics here.
#define SUMSQ_BLOCKS 1024
__device__ float g_sumsq_partials[SUMSQ_BLOCKS];

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n) {
    __shared__ float sbuf[4];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n; i += stride) { float v = g[i]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) g_sumsq_partials[blockIdx.x] = (sbuf[0] + sbuf[1]) + (sbuf[2] + sbuf[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(float* out, int nblocks) {
    __shared__ float sbuf[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += g_sumsq_partials[i];
    sbuf[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sbuf[threadIdx.x] += sbuf[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sbuf[0];
}

extern "C" int mrcnn_sumsq(const float* g, int64_t n, float* out_scalar, void* stream) {
    if (!g || !out_scalar || n <= 0) return MRCNN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int64_t blocks = cdiv64(n, 256 * 8);
    if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, n);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, out_scalar, (int)blocks);
    return mrcnn_launch_status();
}

// gran_coef[i] describes the 64-float granule i of the flat buffers (every tensor starts on a granule):
//   >= 0 : trainable; the value is the L2 gradient coefficient 2*WEIGHT_DECAY/numel (0 for gamma/beta)
//   <  0 : frozen tensor or alignment padding -> gradient forced to 0, parameter and momentum untouched
__global__ __launch_bounds__(256) void grad_prepare_kernel(f32x4* grads, const f32x4* __restrict__ params,
                                                           float grad_scale, const float* __restrict__ gran_coef,
                                                           int64_t n4) {
    __shared__ float sbuf[4];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n4; i += stride) {
        const float c = gran_coef[i >> 4];
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (c >= 0.f) {
            g = grads[i];
            const f32x4 w = params[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[k] = g[k] * grad_scale;
                if (c != 0.f) g[k] += c * w[k];
                s += g[k] * g[k];
            }
        }
        grads[i] = g;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) g_sumsq_partials[blockIdx.x] = (sbuf[0] + sbuf[1]) + (sbuf[2] + sbuf[3]);
}

__global__ __launch_bounds__(256) void sgd_kernel(f32x4* params, f32x4* mom, const f32x4* __restrict__ grads,
                                                  const float* __restrict__ sumsq, float clipnorm, float lr,
                                                  float momentum, const float* __restrict__ gran_coef, int64_t n4,
                                                  unsigned* skipped) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (skipped && !isfinite(sumsq[0])) {                  // guarded form: an overflowed step leaves weights and momentum alone
        if (i == 0) atomicAdd(skipped, 1u);
        return;
    }
    float clip = 1.f;
    if (clipnorm > 0.f) {
        const float norm = sqrtf(sumsq[0]);
        if (norm >= clipnorm) clip = clipnorm / norm;      // K.clip_norm
    }
    for (; i < n4; i += stride) {
        if (gran_coef[i >> 4] < 0.f) continue;
        const f32x4 g = grads[i];
        f32x4 v = mom[i], w = params[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = momentum * v[k] - lr * (g[k] * clip);
            w[k] += v[k];
        }
        mom[i] = v;
        params[i] = w;
    }
}

extern "C" int mrcnn_sgd_momentum(float* params, float* momentum_buf, const float* grads, const float* sumsq,
                                  float clipnorm, float lr, float momentum, const float* gran_coef, int64_t n,
                                  void* stream) {
    if (!params || !momentum_buf || !grads || !sumsq || !gran_coef || n <= 0 || (n & 63)) return MRCNN_ERR_ARG;
    int64_t blocks = cdiv64(n / 4, 256 * 2);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (f32x4*)params,
                       (f32x4*)momentum_buf, (const f32x4*)grads, sumsq, clipnorm, lr, momentum, gran_coef, n / 4,
                       (unsigned*)nullptr);
    return mrcnn_launch_status();
}

// Mixed-precision form (not in the reference, whose graph is float32 throughout): float16 gradients can overflow under the
// static loss scale; when the global squared norm is not finite the step is skipped on the device -- no host
// synchronisation -- and *skipped_steps counts it, so that the host can lower the loss scale when it next looks.
extern "C" int mrcnn_sgd_momentum_guarded(float* params, float* momentum_buf, const float* grads, const float* sumsq,
                                          float clipnorm, float lr, float momentum, const float* gran_coef, int64_t n,
                                          unsigned* skipped_steps, void* stream) {
    if (!params || !momentum_buf || !grads || !sumsq || !gran_coef || !skipped_steps || n <= 0 || (n & 63)) return MRCNN_ERR_ARG;
    int64_t blocks = cdiv64(n / 4, 256 * 2);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (f32x4*)params,
                       (f32x4*)momentum_buf, (const f32x4*)grads, sumsq, clipnorm, lr, momentum, gran_coef, n / 4, skipped_steps);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_grad_prepare(float* grads, const float* params, float grad_scale, const float* gran_coef,
                                  int64_t n, float* sumsq_out, void* stream) {
    if (!grads || !params || !gran_coef || !sumsq_out || n <= 0 || (n & 63)) return MRCNN_ERR_ARG;
    int64_t blocks = cdiv64(n / 4, 256 * 2);
    if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (f32x4*)grads, (const f32x4*)params,
                       grad_scale, gran_coef, n / 4);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, sumsq_out, (int)blocks);
    return mrcnn_launch_status();
}
