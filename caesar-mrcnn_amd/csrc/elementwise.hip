// HBM-bound glue kernels of the Mask R-CNN graph: frozen-BN folding, the backward of the conv
// epilogue (BN affine + residual + activation), 3x3/s2 max pooling (TF SAME), P6 subsampling, the
// adjoint of nearest 2x upsampling, row softmax.  Reference ops: mrcnn/model.py:57-72, 117-130, 187,
// 2005-2022, 946, 1028.
#include "common.h"
#include <string.h>

// ---------------------------------------------------------------------------------------------
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               float* scale, float* shift, float* rstd, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 1.0f / sqrtf(var[i] + eps);
    float s = gamma[i] * r;
    scale[i] = s;
    shift[i] = beta[i] - mean[i] * s;
    if (rstd) rstd[i] = r;
}

extern "C" int mrcnn_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                             float eps, float* scale, float* shift, float* rstd, int64_t n, void* stream) {
    if (!gamma || !beta || !mean || !var || !scale || !shift || n <= 0) return MRCNN_ERR_ARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, mean, var, eps, scale, shift, rstd, n);
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Epilogue backward.  Each workgroup owns a contiguous range of rows of the dense [M, C] tensors and
// walks it element by element (coalesced); per-channel sums go through LDS atomics, then one global
// float atomic per channel per workgroup.
__global__ __launch_bounds__(256) void epilogue_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ z,
    const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* dy_out, float* dz_out, float* dgamma, float* dbeta, float* dbias, int64_t M, int C, int act,
    int64_t rows_per_block) {
    extern __shared__ float sacc[];   // [3][C]
    float* s_db = sacc;
    float* s_dg = sacc + C;
    float* s_dbias = sacc + 2 * C;
    for (int c = threadIdx.x; c < 3 * C; c += 256) sacc[c] = 0.f;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const int64_t e0 = r0 * C, e1 = r1 * C;
    const bool need_stats = dgamma || dbeta;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int c = (int)(e % C);
        float g = dout[e];
        if (act == MRCNN_ACT_RELU) g = out[e] > 0.f ? g : 0.f;
        else if (act == MRCNN_ACT_SIGMOID) { float o = out[e]; g = g * o * (1.f - o); }
        if (dy_out) dy_out[e] = g;
        float dz = scale ? g * scale[c] : g;
        if (dz_out) dz_out[e] = dz;
        if (need_stats) {
            atomicAdd(&s_db[c], g);
            if (dgamma) atomicAdd(&s_dg[c], g * (z[e] - mean[c]) * rstd[c]);
        }
        if (dbias) atomicAdd(&s_dbias[c], dz);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        if (dbeta) atomicAdd(&dbeta[c], s_db[c]);
        if (dgamma) atomicAdd(&dgamma[c], s_dg[c]);
        if (dbias) atomicAdd(&dbias[c], s_dbias[c]);
    }
}

// Vector form for C = 4 * 2^k (every BatchNorm layer of the graph).  A workgroup owns L = 2^lg float4 channel groups
// (blockIdx.y) and a range of rows (blockIdx.x); its 256/L row lanes walk the rows in batches of EPI_U.  Per-channel
// sums: registers -> LDS -> one global atomic per channel per workgroup; the narrow channel slice keeps the number of
// global atomics (the bottleneck of the wide form: 3*C per workgroup) 16x lower at the same number of workgroups.
#define EPI_U 4
__global__ __launch_bounds__(256) void epilogue_bwd_vec_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ z,
    const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* dy_out, float* dz_out, float* dgamma, float* dbeta, float* dbias, int64_t M, int C, int act,
    int64_t rows_per_block, int lg) {
    __shared__ float sacc[3 * 4 * 256];   // [3][4L]
    const int L = 1 << lg, R = 256 >> lg;
    for (int c = threadIdx.x; c < 12 * L; c += 256) sacc[c] = 0.f;
    __syncthreads();
    const int rsub = threadIdx.x >> lg, lane = threadIdx.x & (L - 1);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const bool need_stats = dgamma || dbeta;
    const int c = (blockIdx.y * L + lane) * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *(const f32x4*)(scale + c);
    if (dgamma) { mu = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c); }
    f32x4 a_db = {0.f, 0.f, 0.f, 0.f}, a_dg = a_db, a_bias = a_db;
    // every load of a batch is issued before its first store (the outputs may alias the inputs, so the compiler cannot
    // hoist them itself): one memory round trip per batch instead of one per row
    for (int64_t rb = r0 + rsub; rb < r1; rb += (int64_t)R * EPI_U) {
        f32x4 gg[EPI_U], oo[EPI_U], zz[EPI_U];
#pragma unroll
        for (int u = 0; u < EPI_U; ++u) {
            int64_t r = rb + (int64_t)u * R;
            if (r >= r1) r = r1 - 1;                       // clamped: in range, result discarded below
            const int64_t e = r * C + c;
            gg[u] = *(const f32x4*)(dout + e);
            if (act != MRCNN_ACT_NONE) oo[u] = *(const f32x4*)(out + e);
            if (dgamma) zz[u] = *(const f32x4*)(z + e);
        }
#pragma unroll
        for (int u = 0; u < EPI_U; ++u) {
            const int64_t r = rb + (int64_t)u * R;
            if (r >= r1) break;
            const int64_t e = r * C + c;
            f32x4 g = gg[u];
            if (act == MRCNN_ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = oo[u][k] > 0.f ? g[k] : 0.f;
            } else if (act == MRCNN_ACT_SIGMOID) {
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = g[k] * oo[u][k] * (1.f - oo[u][k]);
            }
            if (dy_out) *(f32x4*)(dy_out + e) = g;
            f32x4 dz;
#pragma unroll
            for (int k = 0; k < 4; ++k) dz[k] = g[k] * sc[k];
            if (dz_out) *(f32x4*)(dz_out + e) = dz;
            if (dgamma) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a_dg[k] += g[k] * (zz[u][k] - mu[k]) * rs[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a_db[k] += g[k]; a_bias[k] += dz[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (need_stats) atomicAdd(&sacc[lane * 4 + k], a_db[k]);
        if (dgamma) atomicAdd(&sacc[4 * L + lane * 4 + k], a_dg[k]);
        if (dbias) atomicAdd(&sacc[8 * L + lane * 4 + k], a_bias[k]);
    }
    __syncthreads();
    const int cb = blockIdx.y * 4 * L;
    for (int j = threadIdx.x; j < 4 * L; j += 256) {
        if (dbeta) atomicAdd(&dbeta[cb + j], sacc[j]);
        if (dgamma) atomicAdd(&dgamma[cb + j], sacc[4 * L + j]);
        if (dbias) atomicAdd(&dbias[cb + j], sacc[8 * L + j]);
    }
}

extern "C" int mrcnn_epilogue_bwd(const float* dout, const float* out, const float* z, const float* scale,
                                  const float* mean, const float* rstd, float* dy_out, float* dz_out,
                                  float* dgamma, float* dbeta, float* dbias, int64_t M, int C, int act,
                                  void* stream) {
    if (!dout || M <= 0 || C <= 0 || C > 4096) return MRCNN_ERR_ARG;
    if (act != MRCNN_ACT_NONE && !out) return MRCNN_ERR_ARG;
    if (dgamma && (!z || !mean || !rstd)) return MRCNN_ERR_ARG;
    const bool pow2 = C >= 16 && (C & (C - 1)) == 0;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (pow2 && al(dout) && al(out) && al(z) && al(scale) && al(mean) && al(rstd) && al(dy_out) && al(dz_out)) {
        const EpiGrid g = mrcnn_epilogue_grid(M, C);
        hipLaunchKernelGGL(epilogue_bwd_vec_kernel, dim3(g.row_blocks, g.chan_blocks), dim3(256), 0, (hipStream_t)stream,
                           dout, out, z, scale, mean, rstd, dy_out, dz_out, dgamma, dbeta, dbias, M, C, act,
                           g.rows_per_block, g.lg);
    } else {
        int64_t rows_per_block = cdiv64(M, 2048);
        // keep at least ~4K elements per workgroup so the per-channel atomics stay negligible
        int64_t min_rows = cdiv64(4096, C);
        if (rows_per_block < min_rows) rows_per_block = min_rows;
        unsigned grid = (unsigned)cdiv64(M, rows_per_block);
        hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(grid), dim3(256), 3 * C * sizeof(float), (hipStream_t)stream,
                           dout, out, z, scale, mean, rstd, dy_out, dz_out, dgamma, dbeta, dbias, M, C, act,
                           rows_per_block);
    }
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* out, int32_t* argmax, int N, int H,
                                   int W, int C, int OH, int OW, int pad_t, int pad_l) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)N * OH * OW * C;
    if (i >= total) return;
    int c = (int)(i % C);
    int64_t p = i / C;
    int ow = (int)(p % OW); p /= OW;
    int oh = (int)(p % OH);
    int n = (int)(p / OH);
    float best = -INFINITY;
    int bi = -1;
    for (int a = 0; a < 3; ++a) {
        int ih = oh * 2 - pad_t + a;
        if ((unsigned)ih >= (unsigned)H) continue;
        for (int b = 0; b < 3; ++b) {
            int iw = ow * 2 - pad_l + b;
            if ((unsigned)iw >= (unsigned)W) continue;
            float v = x[(((int64_t)n * H + ih) * W + iw) * C + c];
            if (bi < 0 || v > best) { best = v; bi = ih * W + iw; }
        }
    }
    out[i] = best;
    if (argmax) argmax[i] = bi;
}

extern "C" int mrcnn_maxpool3x3s2_fwd(const float* x, float* out, int32_t* argmax, int N, int H, int W, int C,
                                      int OH, int OW, int pad_t, int pad_l, void* stream) {
    if (!x || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return MRCNN_ERR_ARG;
    if ((OH - 1) * 2 - pad_t >= H || (OW - 1) * 2 - pad_l >= W) return MRCNN_ERR_ARG;
    int64_t total = (int64_t)N * OH * OW * C;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, out, argmax, N, H, W, C, OH, OW, pad_t, pad_l);
    return mrcnn_launch_status();
}

// gather form of the max-pool adjoint: every input pixel looks at the (at most 4) windows covering it
__global__ void maxpool_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ argmax,
                                   float* dx, int N, int H, int W, int C, int OH, int OW, int pad_t, int pad_l) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)N * H * W * C;
    if (i >= total) return;
    int c = (int)(i % C);
    int64_t p = i / C;
    int iw = (int)(p % W); p /= W;
    int ih = (int)(p % H);
    int n = (int)(p / H);
    const int me = ih * W + iw;
    float acc = 0.f;
    // windows with 2*oh - pad_t <= ih <= 2*oh - pad_t + 2
    int oh_lo = (ih + pad_t - 2 + 1) >> 1; if (oh_lo < 0) oh_lo = 0;
    int oh_hi = (ih + pad_t) >> 1; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_lo = (iw + pad_l - 2 + 1) >> 1; if (ow_lo < 0) ow_lo = 0;
    int ow_hi = (iw + pad_l) >> 1; if (ow_hi > OW - 1) ow_hi = OW - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
            int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
            if (argmax[o] == me) acc += dout[o];
        }
    dx[i] = acc;
}

extern "C" int mrcnn_maxpool3x3s2_bwd(const float* dout, const int32_t* argmax, float* dx, int N, int H, int W,
                                      int C, int OH, int OW, void* stream) {
    if (!dout || !argmax || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MRCNN_ERR_ARG;
    // TF SAME for k=3, s=2: total pad = max((OH-1)*2 + 3 - H, 0), before = total/2
    int tot_h = (OH - 1) * 2 + 3 - H; if (tot_h < 0) tot_h = 0;
    int tot_w = (OW - 1) * 2 + 3 - W; if (tot_w < 0) tot_w = 0;
    int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       dout, argmax, dx, N, H, W, C, OH, OW, tot_h / 2, tot_w / 2);
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------
__global__ void subsample2_kernel(const float* __restrict__ x, float* out, float* dx_acc,
                                  const float* __restrict__ dout, int N, int H, int W, int C) {
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)N * OH * OW * C;
    if (i >= total) return;
    int c = (int)(i % C);
    int64_t p = i / C;
    int ow = (int)(p % OW); p /= OW;
    int oh = (int)(p % OH);
    int n = (int)(p / OH);
    int64_t src = (((int64_t)n * H + 2 * oh) * W + 2 * ow) * C + c;
    if (out) out[i] = x[src];
    else dx_acc[src] += dout[i];
}

extern "C" int mrcnn_subsample2_fwd(const float* x, float* out, int N, int H, int W, int C, void* stream) {
    if (!x || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MRCNN_ERR_ARG;
    int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    hipLaunchKernelGGL(subsample2_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, out, (float*)nullptr, (const float*)nullptr, N, H, W, C);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_subsample2_bwd_acc(const float* dout, float* dx, int N, int H, int W, int C, void* stream) {
    if (!dout || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MRCNN_ERR_ARG;
    int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    hipLaunchKernelGGL(subsample2_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, (float*)nullptr, dx, dout, N, H, W, C);
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------
__global__ void upsample2_bwd_kernel(const float* __restrict__ dout, float* dsrc, int N, int H, int W, int C,
                                     int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)N * H * W * C;
    if (i >= total) return;
    int c = (int)(i % C);
    int64_t p = i / C;
    int w = (int)(p % W); p /= W;
    int h = (int)(p % H);
    int n = (int)(p / H);
    const int W2 = 2 * W;
    int64_t b = (((int64_t)n * 2 * H + 2 * h) * W2 + 2 * w) * C + c;
    float s = (dout[b] + dout[b + C]) + (dout[b + (int64_t)W2 * C] + dout[b + (int64_t)W2 * C + C]);
    dsrc[i] = accumulate ? dsrc[i] + s : s;
}

extern "C" int mrcnn_upsample2_bwd(const float* dout, float* dsrc, int N, int H, int W, int C, int accumulate,
                                   void* stream) {
    if (!dout || !dsrc || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MRCNN_ERR_ARG;
    int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       dout, dsrc, N, H, W, C, accumulate);
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------
__global__ void add_inplace_kernel(float* dst, const float* __restrict__ src, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] += src[i];
}

extern "C" int mrcnn_add_inplace(float* dst, const float* src, int64_t n, void* stream) {
    if (!dst || !src || n <= 0) return MRCNN_ERR_ARG;
    int64_t blocks = cdiv64(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    return mrcnn_launch_status();
}

__global__ void softmax_rows_kernel(const float* __restrict__ logits, float* probs, int64_t rows, int C) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* l = logits + r * C;
    float m = l[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(l[c] - m);
    for (int c = 0; c < C; ++c) probs[r * C + c] = expf(l[c] - m) / s;
}

extern "C" int mrcnn_softmax_rows(const float* logits, float* probs, int64_t rows, int C, void* stream) {
    if (!logits || !probs || rows <= 0 || C <= 0) return MRCNN_ERR_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, probs, rows, C);
    return mrcnn_launch_status();
}


// dst[n, h, w, (a*2+b)*C + c] = src[n, 2h+a, 2w+b, c]: regroups the gradient of the 2x2/s2 transposed
// convolution (mrcnn_mask_deconv, mrcnn/model.py:1087) into the column order of its GEMM.
__global__ void pixel_unshuffle2_kernel(const float* __restrict__ src, float* dst, int N, int H, int W, int C) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over float4 of dst
    const int c4n = C >> 2;
    int64_t total = (int64_t)N * H * W * 4 * c4n;
    if (i >= total) return;
    int c4 = (int)(i % c4n);
    int64_t p = i / c4n;
    int ab = (int)(p & 3); p >>= 2;
    int w = (int)(p % W); p /= W;
    int h = (int)(p % H);
    int n = (int)(p / H);
    const f32x4* s = (const f32x4*)(src + ((((int64_t)n * 2 * H + 2 * h + (ab >> 1)) * 2 * W) + 2 * w + (ab & 1)) * C);
    ((f32x4*)dst)[i] = s[c4];
}

extern "C" int mrcnn_pixel_unshuffle2(const float* src, float* dst, int N, int H, int W, int C, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MRCNN_ERR_ARG;
    int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, N, H, W, C);
    return mrcnn_launch_status();
}

// Fused backward of the mask-head output stage (build_fpn_mask_graph, mrcnn/model.py:1087-1090):
//   mask = sigmoid(conv1x1(up) + b_m),  up = relu(deconv2x2(x) + b_d)
// given dL/dmask.  One pass over `up` (the only large tensor) produces
//   dzg   [M, H/2, W/2, 4*Cd]  gradient w.r.t. the deconv GEMM output, ReLU-masked and regrouped to GEMM columns
//   dWm   [Cd, C]  += up^T . dz          dbm [C] += sum dz          dbd [Cd] += sum dzu
// instead of five passes (sigmoid epilogue, 1x1 wgrad, 1x1 dgrad, ReLU epilogue, pixel unshuffle).
// One thread per deconv channel, a workgroup walks `ppb` consecutive pixels of the [M, H, W] output grid.
#define MOB_MAXC 16
#define MOB_PPB 512
template <int CP>      // CP = C rounded up to 4 / 8 / 16
__global__ void mask_out_bwd_kernel(const float* __restrict__ dmask, const float* __restrict__ mask,
                                    const float* __restrict__ up, const float* __restrict__ wm, float* dzg, float* dWm,
                                    float* dbm, float* dbd, int64_t npix, int H, int W, int Cd, int C) {
    __shared__ __attribute__((aligned(16))) float sdz[MOB_PPB * CP];
    const int ci = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * MOB_PPB;
    const int np = (int)((npix - p0) < MOB_PPB ? (npix - p0) : MOB_PPB);
    // dz = dL/dmask * sigmoid' for this workgroup's pixels, once, into LDS
    for (int i = ci; i < MOB_PPB * CP; i += blockDim.x) {
        const int pl = i / CP, c = i - pl * CP;
        float v = 0.f;
        if (pl < np && c < C) {
            const float g = dmask[(p0 + pl) * C + c], q = mask[(p0 + pl) * C + c];
            v = g * q * (1.f - q);
        }
        sdz[i] = v;
    }
    float wrow[CP], aw[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) { wrow[c] = c < C ? wm[(int64_t)ci * C + c] : 0.f; aw[c] = 0.f; }
    float abd = 0.f;
    __syncthreads();
    const int hw = H * W, W2 = W >> 1, H2 = H >> 1;
    const float* upp = up + p0 * Cd + ci;
#pragma unroll 4
    for (int pl = 0; pl < np; ++pl) {
        const float u = upp[(int64_t)pl * Cd];
        float d = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < CP; c4 += 4) {
            const f32x4 z = *(const f32x4*)&sdz[pl * CP + c4];          // LDS broadcast
#pragma unroll
            for (int e = 0; e < 4; ++e) { d += z[e] * wrow[c4 + e]; aw[c4 + e] += u * z[e]; }
        }
        const float dzu = u > 0.f ? d : 0.f;
        abd += dzu;
        const int64_t pix = p0 + pl;
        const int64_t n = pix / hw;
        const int rem = (int)(pix - n * hw);
        const int y = rem / W, x = rem - y * W;
        dzg[(((n * H2 + (y >> 1)) * W2 + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1))) * Cd + ci] = dzu;
    }
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (c < C) atomicAdd(&dWm[(int64_t)ci * C + c], aw[c]);
    atomicAdd(&dbd[ci], abd);
    if (ci < C) {
        float s_ = 0.f;
        for (int pl = 0; pl < np; ++pl) s_ += sdz[pl * CP + ci];
        atomicAdd(&dbm[ci], s_);
    }
}

// The same pass with float4 accesses: a thread owns 4 deconv channels, 256 / (Cd / 4) pixels are in flight per workgroup step
// (4 for Cd = 256) and four steps are unrolled -- 16 KiB of loads in flight per workgroup instead of 4 (one dword per thread):
// 1.34 -> ~0.9 ms at 2048 ROIs (3.3 GB moved).  The per-thread partial sums of dWm / dbd meet in LDS (the pixel lanes of a channel
// group are different waves) before the one atomic per element and workgroup.  Cd % 4 == 0, Cd / 4 a power of two <= 256.
template <int CP>
__global__ __launch_bounds__(256) void mask_out_bwd_vec_kernel(const float* __restrict__ dmask, const float* __restrict__ mask,
                                                               const float* __restrict__ up, const float* __restrict__ wm, float* dzg,
                                                               float* dWm, float* dbm, float* dbd, int64_t npix, int H, int W, int Cd,
                                                               int C, int lg) {
    __shared__ __attribute__((aligned(16))) float sdz[MOB_PPB * CP];
    __shared__ float sred[256 * 4 * (CP + 1)];                    // [thread][4 channels][CP weights + 1 bias]
    const int tid = threadIdx.x;
    const int L = 1 << lg, PP = 256 >> lg;                        // L = Cd / 4 channel groups, PP pixels in flight
    const int cg = tid & (L - 1), pp = tid >> lg;
    const int ci = cg * 4;
    const int64_t p0 = (int64_t)blockIdx.x * MOB_PPB;
    const int np = (int)((npix - p0) < MOB_PPB ? (npix - p0) : MOB_PPB);
    for (int i = tid; i < MOB_PPB * CP; i += 256) {
        const int pl = i / CP, c = i - pl * CP;
        float v = 0.f;
        if (pl < np && c < C) {
            const float g = dmask[(p0 + pl) * C + c], q = mask[(p0 + pl) * C + c];
            v = g * q * (1.f - q);
        }
        sdz[i] = v;
    }
    float wrow[4][CP], aw[4][CP], abd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        abd[k] = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) { wrow[k][c] = c < C ? wm[(int64_t)(ci + k) * C + c] : 0.f; aw[k][c] = 0.f; }
    }
    __syncthreads();
    const int hw = H * W, W2 = W >> 1, H2 = H >> 1;
#pragma unroll 4
    for (int pl = pp; pl < np; pl += PP) {
        const f32x4 u = *(const f32x4*)(up + (p0 + pl) * Cd + ci);
        f32x4 dzu;
#pragma unroll
        for (int k = 0; k < 4; ++k) dzu[k] = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < CP; c4 += 4) {
            const f32x4 z = *(const f32x4*)&sdz[pl * CP + c4];          // LDS broadcast within a pixel lane
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) { dzu[k] += z[e] * wrow[k][c4 + e]; aw[k][c4 + e] += u[k] * z[e]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { dzu[k] = u[k] > 0.f ? dzu[k] : 0.f; abd[k] += dzu[k]; }
        const int64_t pix = p0 + pl;
        const int64_t n = pix / hw;
        const int rem = (int)(pix - n * hw);
        const int y = rem / W, x = rem - y * W;
        *(f32x4*)(dzg + (((n * H2 + (y >> 1)) * W2 + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1))) * Cd + ci) = dzu;
    }
    // partial sums of the PP pixel lanes -> LDS -> one atomic per element and workgroup
    float* mine = sred + tid * 4 * (CP + 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < CP; ++c) mine[k * (CP + 1) + c] = aw[k][c];
        mine[k * (CP + 1) + CP] = abd[k];
    }
    __syncthreads();
    for (int i = tid; i < L * 4 * (CP + 1); i += 256) {            // element i of channel group i / (4 (CP + 1))
        const int g = i / (4 * (CP + 1)), e = i - g * 4 * (CP + 1);
        float s_ = 0.f;
        for (int q = 0; q < PP; ++q) s_ += sred[(q * L + g) * 4 * (CP + 1) + e];
        const int k = e / (CP + 1), c = e - k * (CP + 1);
        if (c == CP) atomicAdd(&dbd[g * 4 + k], s_);
        else if (c < C) atomicAdd(&dWm[(int64_t)(g * 4 + k) * C + c], s_);
    }
    if (tid < C) {
        float s_ = 0.f;
        for (int pl = 0; pl < np; ++pl) s_ += sdz[pl * CP + tid];
        atomicAdd(&dbm[tid], s_);
    }
}

extern "C" int mrcnn_mask_out_bwd(const float* d_mask_out, const float* mask_out, const float* up, const float* w_mask,
                                  float* dzg, float* dw_mask, float* db_mask, float* db_deconv, int64_t M, int H, int W,
                                  int Cd, int C, void* stream) {
    if (!d_mask_out || !mask_out || !up || !w_mask || !dzg || !dw_mask || !db_mask || !db_deconv) return MRCNN_ERR_ARG;
    if (M <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cd < 64 || Cd > 1024 || (Cd & 63) || C < 1 || C > MOB_MAXC)
        return MRCNN_ERR_ARG;
    const int64_t npix = M * H * W;
    const dim3 grid((unsigned)cdiv64(npix, MOB_PPB)), block(Cd);
    hipStream_t s = (hipStream_t)stream;
    static const bool vec = !(getenv("MRCNN_MASK_OUT_BWD_VEC") && getenv("MRCNN_MASK_OUT_BWD_VEC")[0] == '0');
    const int L = Cd >> 2;
    if (vec && C <= 4 && !(L & (L - 1)) && L <= 256 && !((reinterpret_cast<uintptr_t>(up) | reinterpret_cast<uintptr_t>(dzg)) & 15)) {
        int lg = 0;
        while ((1 << lg) < L) ++lg;
        hipLaunchKernelGGL(mask_out_bwd_vec_kernel<4>, grid, dim3(256), 0, s, d_mask_out, mask_out, up, w_mask, dzg, dw_mask, db_mask,
                           db_deconv, npix, H, W, Cd, C, lg);
        return mrcnn_launch_status();
    }
    if (C <= 4)
        hipLaunchKernelGGL(mask_out_bwd_kernel<4>, grid, block, 0, s, d_mask_out, mask_out, up, w_mask, dzg, dw_mask,
                           db_mask, db_deconv, npix, H, W, Cd, C);
    else if (C <= 8)
        hipLaunchKernelGGL(mask_out_bwd_kernel<8>, grid, block, 0, s, d_mask_out, mask_out, up, w_mask, dzg, dw_mask,
                           db_mask, db_deconv, npix, H, W, Cd, C);
    else
        hipLaunchKernelGGL(mask_out_bwd_kernel<16>, grid, block, 0, s, d_mask_out, mask_out, up, w_mask, dzg, dw_mask,
                           db_mask, db_deconv, npix, H, W, Cd, C);
    return mrcnn_launch_status();
}

// Strided row copy / zero fill as KERNELS whenever sizes and pointers are 4-byte multiples (every use in this
// package): these calls may sit inside a HIP-graph capture (infer_graphed), and a captured memset node proved
// unreliable inside a large graph (stale counters on replay), so nothing on the captured path uses memcpy / memset nodes.
__global__ void copy2d_kernel(unsigned* dst, size_t dst_pitch_w, const unsigned* __restrict__ src, size_t src_pitch_w,
                              size_t row_w, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / row_w, c = i - r * row_w;
        dst[r * dst_pitch_w + c] = src[r * src_pitch_w + c];
    }
}

__global__ void fill_zero_kernel(f32x4* dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

__global__ void fill_zero_bytes_kernel(unsigned char* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 0;
}

extern "C" int mrcnn_copy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes,
                            size_t rows, void* stream) {
    if (!dst || !src || row_bytes == 0 || rows == 0 || dst_pitch < row_bytes || src_pitch < row_bytes) return MRCNN_ERR_ARG;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | dst_pitch | src_pitch | row_bytes;
    if ((bits & 3) == 0) {
        const size_t total = rows * (row_bytes / 4);
        size_t blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)dst, dst_pitch / 4,
                           (const unsigned*)src, src_pitch / 4, row_bytes / 4, total);
        return mrcnn_launch_status();
    }
    hipError_t e = hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, row_bytes, rows, hipMemcpyDeviceToDevice,
                                    (hipStream_t)stream);
    return e == hipSuccess ? MRCNN_OK : MRCNN_ERR_LAUNCH;
}

// GT instance masks cross PCIe bit-packed (mrcnn/model.py:1721-1904 feeds [B, H, W, MAX_GT_INSTANCES] bool per batch): byte
// b of a pixel carries instances 8b .. 8b + 7, least significant bit first (numpy.packbits(..., bitorder="little") along the
// instance axis).  Unpacked here to the uint8 planes the target kernels read; instances >= n_used (the padding up to G) are
// written as zeros, so the host never uploads them.
__global__ __launch_bounds__(256) void unpack_mask_bits_kernel(const unsigned char* __restrict__ packed, unsigned char* out,
                                                               long long npix, int nbytes, int n_used, int G) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix * G) return;
    const long long pix = i / G;
    const int g = (int)(i - pix * G);
    out[i] = g < n_used ? (unsigned char)((packed[pix * nbytes + (g >> 3)] >> (g & 7)) & 1u) : (unsigned char)0;
}

extern "C" int mrcnn_unpack_mask_bits(const void* packed, void* out, int64_t npix, int nbytes_per_pixel, int n_used, int G,
                                      void* stream) {
    if (!out || npix <= 0 || G <= 0 || n_used < 0 || n_used > G || nbytes_per_pixel < (n_used + 7) / 8 || (n_used > 0 && !packed))
        return MRCNN_ERR_ARG;
    const long long total = (long long)npix * G;
    hipLaunchKernelGGL(unpack_mask_bits_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)packed, (unsigned char*)out, (long long)npix, nbytes_per_pixel, n_used, G);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_fill_zero(void* dst, size_t bytes, void* stream) {
    if (!dst || bytes == 0) return MRCNN_ERR_ARG;
    if (((reinterpret_cast<uintptr_t>(dst) | bytes) & 15) == 0) {
        const size_t n4 = bytes / 16;
        size_t blocks = (n4 + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (f32x4*)dst, n4);
        return mrcnn_launch_status();
    }
    // unaligned pointer or size: a byte kernel -- never a memset node (captured memset nodes misbehave on replay, see above)
    size_t blocks = (bytes + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fill_zero_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, bytes);
    return mrcnn_launch_status();
}

int g_mrcnn_wgrad_lds_pad = 0;
int g_mrcnn_h16_phase = 1;
int g_mrcnn_h16_slab = -1;               // -1: MRCNN_H16_SLAB decides (default off: measured slower); 0 / 1: set by mrcnn_tuning_set (tests, A/B)
int g_mrcnn_proposal_skip_zero = 0;
int g_mrcnn_sk16 = 0;                    // 1 while the inference graph is issued (engine.infer): the 16 x 16-tile small-layer kernel may be picked

extern "C" int mrcnn_tuning_set(const char* key, long long value) {
    if (!key) return MRCNN_ERR_ARG;
    if (!strcmp(key, "wgrad_lds_pad")) {
        if (value < 0 || value > 32768) return MRCNN_ERR_ARG;
        g_mrcnn_wgrad_lds_pad = (int)value;
        return MRCNN_OK;
    }
    if (!strcmp(key, "h16_phase")) {
        if (value < 0 || value > 1) return MRCNN_ERR_ARG;
        g_mrcnn_h16_phase = (int)value;
        return MRCNN_OK;
    }
    if (!strcmp(key, "h16_slab")) {
        if (value < -1 || value > 1) return MRCNN_ERR_ARG;
        g_mrcnn_h16_slab = (int)value;
        return MRCNN_OK;
    }
    if (!strcmp(key, "sk16")) {
        if (value < 0 || value > 1) return MRCNN_ERR_ARG;
        g_mrcnn_sk16 = (int)value;
        return MRCNN_OK;
    }
    if (!strcmp(key, "proposal_skip_zero")) {
        if (value < 0 || value > 1) return MRCNN_ERR_ARG;
        g_mrcnn_proposal_skip_zero = (int)value;
        return MRCNN_OK;
    }
    return MRCNN_ERR_UNSUPPORTED;
}

extern "C" const char* mrcnn_hip_version(void) { return "mrcnn_hip 0.1 (gfx950)"; }
