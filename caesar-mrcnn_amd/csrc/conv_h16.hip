// 16-bit matrix-core path (BASELINE.json configs[4]: fp16 weights / activations on the CDNA4 fp16 MFMA),
// stage 1: the convolutions of the ROI heads.  Operands are float16 (or bfloat16), accumulation, bias,
// frozen-BN affine and activation are float32, the result is rounded once to 16 bits.
//
//   conv_fwd_h16_kernel   implicit-GEMM forward (also the data gradient, with the flipped weight image):
//                         256 x 128 output tile per workgroup, 4 waves of 128 x 64 (4 x 2 MFMA tiles of
//                         v_mfma_f32_32x32x16_{f16,bf16}), K-step 32, both operand tiles by buffer-addressed
//                         LDS-DMA exactly as conv_fwd_blds_kernel (32-bit lane offsets, scalar tap offsets,
//                         range-check zero fill), 64-byte LDS rows with the source-side XOR chunk swizzle.
//                         The B operand needs 8 consecutive k per lane, so weights are kept as W^T
//                         [Cout][KH*KW*Cin] (weights_to_h16_kernel writes it, and the flipped image
//                         [Cin][KH*KW*Cout] the data gradient uses, from the float32 HWIO master copy).
//   cast kernels          float32 <-> 16-bit, elementwise.
#include "common.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* h16_lds_ptr;

template <typename T> struct H16Traits;
template <> struct H16Traits<_Float16> {
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct H16Traits<__bf16> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

struct ConvH16Args {
    const void* x; const void* wt; const float* bias; const float* scale; const float* shift; void* out; void* z;
    int N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, OH, OW, act, M, Ktot;
    unsigned x_shift, x_records, w_records;
};

#define H16_OOB_OFFSET 0xFFFFFFF0u

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_fwd_h16_kernel(const ConvH16Args p) {
    typedef typename H16Traits<T>::v8 v8;
    constexpr int BM = 256, BN = 128, TM = 4, TN = 2;
    constexpr int ROWB = 64;                                    // bytes per LDS row (32 x 16-bit)
    constexpr int AB = BM * ROWB, BB = BN * ROWB;               // 16 KiB + 8 KiB per buffer
    __shared__ __attribute__((aligned(16))) char lds[2 * (AB + BB)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.Cout / BN;
    const int mtile = blockIdx.x / ntiles, ntile = blockIdx.x % ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int ohw = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - p.x_shift), 0, p.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.w_records, 0x00020000);

    // A: 16 pieces of 16 rows (this wave: pieces wave + 4 jj); lane (r, c) fetches logical 16-byte chunk c ^ ((r>>2)&3)
    unsigned a_voff[4];
    unsigned long long a_mask[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int r = (wave + jj * 4) * 16 + (lane >> 2);
        const int cl = (lane & 3) ^ ((r >> 2) & 3);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / ohw, rem = mm - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        const long long off = ((long long)n * p.H * p.W * p.Cin + ((long long)ih0 * p.W + iw0) * p.Cin + cl * 8) * 2 + p.x_shift;
        a_voff[jj] = (unsigned)off;
        unsigned long long mk = 0ull;
        if (ok)
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int th = t / p.KW, tw = t - th * p.KW;
                if ((unsigned)(ih0 + th) < (unsigned)p.H && (unsigned)(iw0 + tw) < (unsigned)p.W) mk |= 1ull << t;
            }
        a_mask[jj] = mk;
    }
    // B = W^T [Cout][Ktot]: 8 pieces of 16 output channels
    unsigned b_voff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int r = (wave + jj * 4) * 16 + (lane >> 2);
        const int cl = (lane & 3) ^ ((r >> 2) & 3);
        b_voff[jj] = (unsigned)((((long long)(n0 + r)) * p.Ktot + cl * 8) * 2);
    }

    int kh = 0, kw = 0, ci0 = 0, tap = 0;
    auto stage = [&](char* ab) {
        char* bb = ab + AB;
        const unsigned soff_a = (unsigned)(((kh * p.W + kw) * p.Cin + ci0) * 2);
        const unsigned soff_b = (unsigned)((tap * p.Cin + ci0) * 2);
        const unsigned long long bit = 1ull << tap;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const unsigned vo = (a_mask[jj] & bit) ? a_voff[jj] : H16_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (h16_lds_ptr)(ab + (wave + jj * 4) * 1024), 16, vo, soff_a, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (h16_lds_ptr)(bb + (wave + jj * 4) * 1024), 16, b_voff[jj], soff_b, 0, 0);
        ++tap;                                                  // channel-chunk outer, filter-tap inner
        if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; tap = 0; ci0 += 32; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    // operand k-group kk (16 k values) of a 32-row tile: lane (li, lh) reads logical chunk 2*kk + lh of row li
    const int arow = wm * 128 + li, brow = wn * 64 + li;
    const char* a_rd[2];
    const char* b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_rd[kk] = lds + arow * ROWB + (((2 * kk + lh) ^ ((arow >> 2) & 3)) << 4);
        b_rd[kk] = lds + AB + brow * ROWB + (((2 * kk + lh) ^ ((brow >> 2) & 3)) << 4);
    }

    auto compute = [&](auto curc) {
        constexpr int BO = decltype(curc)::value * (AB + BB);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = *(const v8*)(a_rd[kk] + BO + a * 32 * ROWB);
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = *(const v8*)(b_rd[kk] + BO + b * 32 * ROWB);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = H16Traits<T>::mfma(av[a], bv[b], acc[a][b]);
        }
    };

    const int nk = p.Ktot / 32;
    stage(lds);
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
        if (ks + 1 < nk) stage(lds + (AB + BB));
        compute(std::integral_constant<int, 0>{});
        __syncthreads();
        if (ks + 1 < nk) {
            if (ks + 2 < nk) stage(lds);
            compute(std::integral_constant<int, 1>{});
            __syncthreads();
        }
    }

    // ---- epilogue: bias, frozen-BN affine, activation in float32; one rounding to 16 bits --------------
    T* out = (T*)p.out;
    T* zo = (T*)p.z;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn * 64 + b * 32 + li;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float sc = p.scale ? p.scale[n] : 1.f, sh = p.scale ? p.shift[n] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int mb = m0 + wm * 128 + a * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) continue;
                const float zv = acc[a][b][r] + bias;
                const long long addr = (long long)m * p.Cout + n;
                if (zo) zo[addr] = (T)zv;
                float y = sc * zv + sh;
                if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
                else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
                out[addr] = (T)y;
            }
        }
    }
}

// W (float32, HWIO [tap][ci][co]) -> W^T [co][tap*Cin + ci] (forward operand) and, when wanted, the
// data-gradient operand [ci][tapT*Cout + co] with tapT the 180-degree rotated tap.
template <typename T>
__global__ void weights_to_h16_kernel(const float* __restrict__ w, T* wt_f, T* wt_d, int KH, int KW, int Cin, int Cout) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int kh = tap / KW, kw = tap % KW;
    const int tap_t = (KH - 1 - kh) * KW + (KW - 1 - kw);
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long Kf = (long long)KH * KW * Cin, Kd = (long long)KH * KW * Cout;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        const float v = (ci < Cin && co < Cout) ? w[((long long)tap * Cin + ci) * Cout + co] : 0.f;
        tile[r][tx] = v;
        if (wt_d && ci < Cin && co < Cout) wt_d[(long long)ci * Kd + (long long)tap_t * Cout + co] = (T)v;
    }
    __syncthreads();
    if (wt_f)
        for (int r = ty; r < 32; r += 8) {
            const int co = co0 + r, ci = ci0 + tx;
            if (co < Cout && ci < Cin) wt_f[(long long)co * Kf + (long long)tap * Cin + ci] = (T)tile[tx][r];
        }
}

template <typename T>
__global__ void cast_to_h16_kernel(const float* __restrict__ src, T* dst, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *(const f32x4*)(src + i);
        dst[i] = (T)v[0]; dst[i + 1] = (T)v[1]; dst[i + 2] = (T)v[2]; dst[i + 3] = (T)v[3];
    } else {
        for (long long j = i; j < n; ++j) dst[j] = (T)src[j];
    }
}

template <typename T>
__global__ void cast_from_h16_kernel(const T* __restrict__ src, float* dst, long long n, float mul) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (long long j = i; j < i + 4 && j < n; ++j) dst[j] = (float)src[j] * mul;
}

static int h16_dtype_ok(int dtype) { return dtype == MRCNN_DTYPE_F16 || dtype == MRCNN_DTYPE_BF16; }

extern "C" int mrcnn_conv2d_fwd_h16(const mrcnn_conv_desc* d, int dtype, const void* x, const void* w_t, const float* bias,
                                    const float* scale, const float* shift, void* out, void* z_out, void* stream) {
    if (!d || !x || !w_t || !out || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 ||
        d->OH <= 0 || d->OW <= 0 || d->KH * d->KW > 64)
        return MRCNN_ERR_ARG;
    if (d->Cin % 32 || d->Cout % 128 || d->res_mode != MRCNN_RES_NONE || d->out_mode != MRCNN_OUT_NHWC || d->cmod != d->Cout)
        return MRCNN_ERR_ARG;                                    // the shapes of the ROI heads; nothing else yet
    if (scale && !shift) return MRCNN_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w_t) & 15)) return MRCNN_ERR_ARG;
    const long long M = (long long)d->N * d->OH * d->OW;
    const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 2;
    const long long shift_b = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 2;
    const long long wbytes = (long long)d->KH * d->KW * d->Cin * d->Cout * 2;
    if (M >= (1LL << 31) || xbytes + shift_b >= 0x7FFFFFF0LL || wbytes >= 0x7FFFFFF0LL) return MRCNN_ERR_ARG;
    ConvH16Args a;
    a.x = x; a.wt = w_t; a.bias = bias; a.scale = scale; a.shift = shift; a.out = out; a.z = z_out;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.OH = d->OH; a.OW = d->OW; a.act = d->act;
    a.M = (int)M; a.Ktot = d->KH * d->KW * d->Cin;
    a.x_shift = (unsigned)shift_b; a.x_records = (unsigned)(xbytes + shift_b); a.w_records = (unsigned)wbytes;
    const unsigned blocks = (unsigned)(((M + 255) / 256) * (d->Cout / 128));
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(conv_fwd_h16_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(conv_fwd_h16_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_weights_to_h16(const float* w, void* wt_fwd, void* wt_dgrad, int KH, int KW, int Cin, int Cout, int dtype,
                                    void* stream) {
    if (!w || (!wt_fwd && !wt_dgrad) || KH <= 0 || KW <= 0 || Cin <= 0 || Cout <= 0 || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, KH * KW);
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(weights_to_h16_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, w, (_Float16*)wt_fwd,
                           (_Float16*)wt_dgrad, KH, KW, Cin, Cout);
    else
        hipLaunchKernelGGL(weights_to_h16_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wt_fwd,
                           (__bf16*)wt_dgrad, KH, KW, Cin, Cout);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_cast_to_h16(const float* src, void* dst, int64_t n, int dtype, void* stream) {
    if (!src || !dst || n < 0 || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (n == 0) return 0;
    if (reinterpret_cast<uintptr_t>(src) & 15) return MRCNN_ERR_ARG;
    const unsigned blocks = (unsigned)cdiv64(cdiv64(n, 4), 256);
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(cast_to_h16_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, (long long)n);
    else
        hipLaunchKernelGGL(cast_to_h16_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, (long long)n);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_cast_from_h16(const void* src, float* dst, int64_t n, int dtype, float multiplier, void* stream) {
    if (!src || !dst || n < 0 || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)cdiv64(cdiv64(n, 4), 256);
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(cast_from_h16_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, dst,
                           (long long)n, multiplier);
    else
        hipLaunchKernelGGL(cast_from_h16_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, dst,
                           (long long)n, multiplier);
    return mrcnn_launch_status();
}
