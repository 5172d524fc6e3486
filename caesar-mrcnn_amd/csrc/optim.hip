// Optimiser step of MaskRCNN.compile (mrcnn/model.py:2255-2291) on flat fp32 buffers:
//   * L2 regulariser keras.regularizers.l2(WEIGHT_DECAY)(w) / size(w) for every trainable weight whose
//     name lacks gamma/beta: gradient 2*WEIGHT_DECAY*w/size(w), folded in as a per-segment coefficient;
//   * keras.optimizers.SGD(lr, momentum, clipnorm): [3P Keras 2.2.4] global-norm clipping
//     g *= clipnorm / max(norm, clipnorm), then v = momentum*v - lr*g, w += v.
// HBM-bound: 3 reads + 2 writes per parameter; 16-byte accesses, grid-stride.
#include "common.h"

__device__ __forceinline__ int find_seg(const int64_t* __restrict__ seg_offset, int num_seg, int64_t i) {
    int lo = 0, hi = num_seg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (seg_offset[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

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

__global__ void sgd_kernel(float* params, float* mom, float* grads,
                           const float* __restrict__ sumsq, float clipnorm, float lr, float momentum,
                           float grad_scale, const uint8_t* __restrict__ trainable,
                           const int64_t* __restrict__ seg_offset, const int64_t* __restrict__ seg_numel,
                           const float* __restrict__ seg_l2, int num_seg, int64_t n, int mode) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float clip = 1.f;
    if (mode == 1 && clipnorm > 0.f) {
        const float norm = sqrtf(sumsq[0]);
        if (norm >= clipnorm) clip = clipnorm / norm;      // K.clip_norm
    }
    for (; i < n; i += stride) {
        int s = find_seg(seg_offset, num_seg, i);
        if (i >= seg_offset[s] + seg_numel[s]) continue;    // padding between segments
        if (mode == 0) {
            // gradient preparation: average over ranks, add the L2 term
            float g = grads[i] * grad_scale;
            float c = seg_l2 ? seg_l2[s] : 0.f;
            if (trainable && !trainable[s]) g = 0.f;
            else if (c != 0.f) g += c * params[i];
            grads[i] = g;
        } else {
            if (trainable && !trainable[s]) continue;
            float g = grads[i] * clip;
            float v = momentum * mom[i] - lr * g;
            mom[i] = v;
            params[i] += v;
        }
    }
}

extern "C" int mrcnn_sgd_momentum(float* params, float* momentum_buf, const float* grads, const float* sumsq,
                                  float clipnorm, float lr, float momentum, float grad_scale,
                                  const uint8_t* trainable_mask_per_seg, const int64_t* seg_offset,
                                  const int64_t* seg_numel, int num_seg, int64_t n, void* stream) {
    if (!params || !momentum_buf || !grads || !sumsq || !seg_offset || !seg_numel || num_seg <= 0 || n <= 0)
        return MRCNN_ERR_ARG;
    int64_t blocks = cdiv64(n, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, momentum_buf,
                       (float*)grads, sumsq, clipnorm, lr, momentum, grad_scale, trainable_mask_per_seg, seg_offset,
                       seg_numel, (const float*)nullptr, num_seg, n, 1);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_grad_prepare(float* grads, const float* params, float grad_scale,
                                  const uint8_t* trainable_mask_per_seg, const int64_t* seg_offset,
                                  const int64_t* seg_numel, const float* seg_l2, int num_seg, int64_t n,
                                  void* stream) {
    if (!grads || !params || !seg_offset || !seg_numel || num_seg <= 0 || n <= 0) return MRCNN_ERR_ARG;
    int64_t blocks = cdiv64(n, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)params,
                       (float*)nullptr, grads, (const float*)nullptr, 0.f, 0.f, 0.f, grad_scale,
                       trainable_mask_per_seg, seg_offset, seg_numel, seg_l2, num_seg, n, 0);
    return mrcnn_launch_status();
}
