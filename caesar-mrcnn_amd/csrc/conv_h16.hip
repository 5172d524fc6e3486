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
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <utility>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* h16_lds_ptr;

template <typename T> struct H16Traits;
template <> struct H16Traits<_Float16> {
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct H16Traits<__bf16> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

static int h16_dtype_ok(int dtype) { return dtype == MRCNN_DTYPE_F16 || dtype == MRCNN_DTYPE_BF16; }

struct ConvH16Args {
    const void* x; const void* wt; const float* bias; const float* scale; const float* shift; void* out; void* z;
    int N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, OH, OW, act, M, Ktot;
    unsigned x_shift, x_records, w_records, out_records;
    unsigned mg_ohw, sh_ohw, mg_ow, sh_ow;           // m / (OH*OW) and rem / OW as multiply-high + shift (0 = divisor 1)
    int ptiles, mtile0;                              // phased kernel: tiles it owns; small-tile kernel: first 64-row tile (remainder launch)
    unsigned long long* dbg;                         // phased kernel: cycle stamps of workgroup 0 (tools/h16p_trace.py), else null
    int out_mode, cmod; long long ons, ohs, ows;     // MRCNN_OUT_DECONV2: pixel-shuffle store of the 2x2 transposed conv
    const void* res;                                 // small-tile kernel: 16-bit tensor added before the activation (strides of out)
    int dense;                                       // out is plain NHWC [M][Cout]
    // small-tile kernel as a data gradient fused with the epilogue backward of the layer below (mrcnn_conv2d_dgrad_ep_h16)
    const void* fb_out; const void* fb_z; const float* fb_scale; const float* fb_mean; const float* fb_rstd;
    float* fb_dgamma; float* fb_dbeta; float* fb_dbias; void* fb_dy; int fb_act; float fb_gmul;
    int ep_vec;                                      // small-tile kernel: 16-byte epilogue through LDS (pointers / strides allow it)
};

#define H16_OOB_OFFSET 0xFFFFFFF0u

template <class F, int... I>
__device__ __forceinline__ void h16_static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// s_waitcnt vmcnt(N) for a compile-time N (the immediate has to be in the instruction text)
template <int N> __device__ __forceinline__ void h16_wait_vmcnt() {
    static_assert(N == 0 || N == 4 || N == 6 || N == 8 || N == 12 || N == 16, "add the count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// Tile = (WAVES_M x 128) x (WAVES_N x 64), one wave per 128 x 64 (4 x 2 MFMA tiles of 32 x 32 x 16), K-step 32.
//   <T, 2, 2, 2>  256 x 128, 4 waves, double buffering; every wave waits for all its loads at each barrier and the DMA
//                 latency is hidden by 3 workgroups per CU (48 KiB LDS each).  24 KiB fetched per 2.1 MFLOP: at the
//                 matrix rate that is 47 B/clk/CU out of L2 -- the kernel is operand-starved (PMC: 32 % MFMA-busy).
//   <T, 4, 2, 4>  256 x 256, 8 waves (two per SIMD), ONE workgroup per CU: a ring of four 32 KiB stages with the loads
//                 issued three K-steps ahead and counted waits (s_waitcnt vmcnt(8 / 4 / 0): the two younger stages
//                 stay in flight across the raw s_barrier).  32 KiB per 4.2 MFLOP halves the L2 traffic per flop, and
//                 with one 256-row tile per CU the nine shifted tap reads of a channel chunk (K is walked chunk-outer,
//                 tap-inner) find their rows in the CU's own vector cache instead of evicting each other.
//   <T, 4, 2, 2, 4>  256 x 256 with FOUR waves of 128 x 128 (TN = 4: 4 x 4 MFMA tiles, 256 accumulator registers -- one wave per SIMD,
//                 so they fit): a wave reads 16 KiB of operands per K-step for 32 MFMAs where the 8-wave forms read 12 KiB for 16;
//                 per CU and K-step 64 + 32 KiB of LDS traffic against 1024 matrix cycles instead of 96 + 32.  Epilogue through
//                 LDS (16-byte stores), wave by wave.  MRCNN_H16_TILE=wave128.
template <typename T, int NBUF, int WAVES_M, int WAVES_N, int TN = 2>      // TN = 4 wants NBUF = 4 (the ring index is masked)
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, NBUF <= 3 ? 2 : (WAVES_M * WAVES_N) / 4) void conv_fwd_h16_kernel(const ConvH16Args p) {
    typedef typename H16Traits<T>::v8 v8;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * 128, BN = WAVES_N * 32 * TN, TM = 4;
    constexpr int ROWB = 64;                                    // bytes per LDS row (32 x 16-bit)
    constexpr int AB = BM * ROWB, BB = BN * ROWB;               // bytes per stage
    constexpr int APW = BM / 16 / NW, BPW = BN / 16 / NW;       // 1 KiB DMA pieces (16 rows) per wave and stage
    constexpr int DPS = APW + BPW;                              // DMA instructions per wave and stage
    __shared__ __attribute__((aligned(16))) char lds[NBUF * (AB + BB)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntiles = p.Cout / BN;
    const int mtile = blockIdx.x / ntiles, ntile = blockIdx.x % ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int ohw = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - p.x_shift), 0, p.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.w_records, 0x00020000);

    // A: BM / 16 pieces of 16 rows (this wave: pieces wave + NW jj); lane (r, c) fetches logical 16-byte chunk c ^ ((r>>2)&3)
    unsigned a_voff[4];                                         // [APW] in use (fixed bounds: a dependent bound here made
    unsigned long long a_mask[4];                               //  hipcc drop the kernel's host stub, see b_voff)
#pragma unroll
    for (int jj = 0; jj < APW; ++jj) {
        const int r = (wave + jj * NW) * 16 + (lane >> 2);
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
    // B = W^T [Cout][Ktot]: BN / 16 pieces of 16 output channels
    unsigned b_voff[4];                                         // [BPW] in use
    static_assert(APW <= 4 && BPW <= 4, "piece bookkeeping arrays");
#pragma unroll
    for (int jj = 0; jj < BPW; ++jj) {
        const int r = (wave + jj * NW) * 16 + (lane >> 2);
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
        for (int jj = 0; jj < APW; ++jj) {
            const unsigned vo = (a_mask[jj] & bit) ? a_voff[jj] : H16_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (h16_lds_ptr)(ab + (wave + jj * NW) * 1024), 16, vo, soff_a, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < BPW; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (h16_lds_ptr)(bb + (wave + jj * NW) * 1024), 16, b_voff[jj], soff_b, 0, 0);
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
    const int arow = wm * 128 + li, brow = wn * (32 * TN) + li;
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
    if constexpr (TN == 4) {
        // One wave per SIMD: nobody else hides this wave's operand reads, so they are software-pipelined against its own MFMAs.  Two
        // fragment sets; the reads of the NEXT half K-step (16 k) are issued before the 16 MFMAs of the current one (issuing those
        // takes the wave ~512 cycles, the reads land meanwhile), and the barrier / DMA issue of the next stage sit between the two
        // halves of a K-step instead of in front of it.
        constexpr int D = NBUF - 1;
        v8 fa[2][TM], fb[2][TN];
        // (a plain loop with a runtime stage offset: unrolled over the four stages, the allocator no longer kept the 256
        // accumulators in place and moved half of them between registers every K-step)
        auto rd = [&](auto setc, const int bo, auto kkc) {
            constexpr int F = decltype(setc)::value, kk = decltype(kkc)::value;
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[F][a] = *(const v8*)(a_rd[kk] + bo + a * 32 * ROWB);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[F][b] = *(const v8*)(b_rd[kk] + bo + b * 32 * ROWB);
        };
        auto mm = [&](auto setc) {
            constexpr int F = decltype(setc)::value;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = H16Traits<T>::mfma(fa[F][a], fb[F][b], acc[a][b]);
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        auto land = [&](int ks) {                               // stage ks has landed for this wave (younger stages stay in flight)
            const int younger = nk - 1 - ks;
            if (younger >= 2) h16_wait_vmcnt<2 * DPS>();
            else if (younger >= 1) h16_wait_vmcnt<DPS>();
            else h16_wait_vmcnt<0>();
        };
        for (int s0 = 0; s0 < D && s0 < nk; ++s0) stage(lds + s0 * (AB + BB));
        land(0);
        __builtin_amdgcn_s_barrier();
        if (D < nk) stage(lds + D * (AB + BB));
        rd(I0{}, 0, I0{});
        int cur = 0;                                            // ring slot of stage ks
#pragma nounroll
        for (int ks = 0; ks < nk; ++ks) {
            const int nxt = (cur + 1) & (NBUF - 1);
            if (!(p.mtile0 & 2)) rd(I1{}, cur * (AB + BB), I1{});
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.mtile0 & 1)) mm(I0{});
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < nk) {
                land(ks + 1);
                __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0), as a builtin so that the compiler's own wait bookkeeping sees it: this wave's reads of stage ks are done (issued 16 MFMAs ago)
                __builtin_amdgcn_s_barrier();                           // ... and everybody's: its slot may be refilled
                if (ks + 1 + D < nk) stage(lds + cur * (AB + BB));      // stage ks + 1 + D lives in slot (cur + 1 + D) % NBUF = cur
                if (!(p.mtile0 & 2)) rd(I0{}, nxt * (AB + BB), I0{});
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.mtile0 & 1)) mm(I1{});
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    } else if constexpr (NBUF == 2) {
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
    } else {
        constexpr int D = NBUF - 1;                             // prefetch distance in K-steps; DPS DMA instructions per stage and wave
        for (int s0 = 0; s0 < D && s0 < nk; ++s0) stage(lds + s0 * (AB + BB));
        for (int ks0 = 0; ks0 < nk; ks0 += NBUF) {
            h16_static_for([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                const int ks = ks0 + S;
                if (ks < nk) {
                    const int younger = nk - 1 - ks;            // stages issued after ks that may still be in flight (<= D - 1)
                    if (D >= 3 && younger >= 2) h16_wait_vmcnt<2 * DPS>();
                    else if (younger >= 1) h16_wait_vmcnt<DPS>();
                    else h16_wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();               // stage ks has landed for every wave; everyone is done with stage ks - 1
                    if (ks + D < nk) stage(lds + ((S + D) % NBUF) * (AB + BB));
                    compute(std::integral_constant<int, S>{});
                }
            }, std::make_integer_sequence<int, NBUF>{});
        }
    }

    // ---- epilogue: bias, frozen-BN affine, activation in float32; one rounding to 16 bits --------------
    // rows outermost: one output address per row (the transposed conv needs two divisions for it), then the wave's two
    // column tiles -- keeps the number of live registers small (128 x 64 accumulators per wave are already 128 of them)
    T* out = (T*)p.out;
    T* zo = (T*)p.z;
    const bool deconv = p.out_mode == MRCNN_OUT_DECONV2;        // column n = (a*2+b)*cmod + c -> pixel (2oh+a, 2ow+b), channel c
    if constexpr (TN == 4) {
        // 128 x 128 accumulators per wave: 256 two-byte stores per lane would cost more than the K loop.  Each 32-pixel block goes
        // through the wave's own LDS region in float32 (rows of 132 floats) and leaves as 16-byte pieces, a pixel's 256 bytes by 16
        // neighbouring lanes.  Dense NHWC output only (the host checks).
        typedef T t8 __attribute__((ext_vector_type(8)));
        constexpr int SST = 132;
        __syncthreads();                                        // every wave is done with the ring
        float* stg = (float*)lds + wave * (32 * SST);
        const int c8 = (lane & 15) * 8, nn = n0 + wn * 128 + c8;
        float cbv[8], csv[8], chv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cbv[e] = p.bias ? p.bias[nn + e] : 0.f;
            csv[e] = p.scale ? p.scale[nn + e] : 1.f;
            chv[e] = p.scale ? p.shift[nn + e] : 0.f;
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[(4 * lh + (r & 3) + 8 * (r >> 2)) * SST + b * 32 + li] = acc[a][b][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int m = m0 + wm * 128 + a * 32 + row;
                const f32x4 v0 = *(const f32x4*)&stg[row * SST + c8], v1 = *(const f32x4*)&stg[row * SST + c8 + 4];
                if (m >= p.M) continue;
                const long long addr = (long long)m * p.Cout + nn;
                t8 yo, zv8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float zv = (e < 4 ? v0[e & 3] : v1[e & 3]) + cbv[e];
                    zv8[e] = (T)zv;
                    float y = csv[e] * zv + chv[e];
                    if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
                    else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
                    yo[e] = (T)y;
                }
                if (zo) *(t8*)(zo + addr) = zv8;
                *(t8*)(out + addr) = yo;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return;
    }
    int cn[TN], cab[TN];
    float cbias[TN], csc[TN], csh[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn * (32 * TN) + b * 32 + li;
        cab[b] = deconv ? n / p.cmod : 0;
        cn[b] = deconv ? n - cab[b] * p.cmod : n;
        cbias[b] = p.bias ? p.bias[cn[b]] : 0.f;
        csc[b] = p.scale ? p.scale[cn[b]] : 1.f;
        csh[b] = p.scale ? p.shift[cn[b]] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int mb = m0 + wm * 128 + a * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m >= p.M) continue;
            long long rowaddr = (long long)m * p.Cout;
            int oh = 0, ow = 0;
            if (deconv) {
                const int ni = m / ohw, rem = m - ni * ohw;
                oh = rem / p.OW; ow = rem - oh * p.OW;
                rowaddr = (long long)ni * p.ons;
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const long long addr = deconv ? rowaddr + (long long)(2 * oh + (cab[b] >> 1)) * p.ohs +
                                                    (long long)(2 * ow + (cab[b] & 1)) * p.ows + cn[b]
                                              : rowaddr + cn[b];
                const float zv = acc[a][b][r] + cbias[b];
                if (zo) zo[addr] = (T)zv;
                float y = csc[b] * zv + csh[b];
                if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
                else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
                out[addr] = (T)y;
            }
        }
    }
}

// The 16-byte buffer stores of the phased kernel's epilogue and the vector unit.  gfx950 reads the data of a
// buffer_store_dwordx4 a few cycles after issue; a vector instruction that overwrites one of the four registers in the next
// slot wins the race for the lanes read last.  LLVM's hazard recogniser inserts the wait state only for the form with a
// CONSTANT soffset (the ISA manual exempts an SGPR soffset), yet tools/h16p_first_call.py caught exactly this on the SGPR
// form: the first dword of lanes 12-15 / 28-31 / 44-47 / 60-63 of a store that was followed by a v_mul into its first data
// register held the multiply's result -- about one first call in ten of a fresh process (cold instruction cache), and it
// is how a 16-bit training run picked up its first non-finite activation.  So the stores of a chunk sit between two
// scheduling barriers with s_nop 2 behind the last one: nothing but stores and scalar bookkeeping follows a store for three
// slots.  (The stores stay compiler builtins: as inline asm the recogniser no longer sees their SGPR operands, and the
// "vector write of an SGPR, then a memory instruction reads it" wait states go missing -- measured: wrong addresses.)
template <int N> __device__ __forceinline__ void h16p_wait() {        // s_waitcnt vmcnt(min(N, 63)): the field has 6 bits
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if constexpr (N == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else if constexpr (N == 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
    else { static_assert(N > 56, "add the count"); asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); }
}

// Phased 256 x 256 tile for the big layers (mask head: M = 401 408, N = 256, K = 2 304): 8 waves = two groups of four
// (wr = wave >> 2; waves w and w + 4 share a SIMD), each wave 128 pixels x 64 channels as 8 x 4 tiles of
// v_mfma_f32_16x16x32, K-step 64, two 64 KiB stages.  A K-step is four phases of
//     [operand reads for this phase | one quarter of a later stage by LDS-DMA | s_waitcnt vmcnt(8)]  s_barrier
//     [16 MFMAs = one quadrant of the wave's tile x K 64, raised priority]                            s_barrier
// and group 1 runs ONE barrier behind group 0 (it takes an extra s_barrier first): while one wave of a SIMD is in its
// MFMA half the other is in its read/DMA half, so the matrix pipe sees MFMAs back to back instead of both waves issuing
// loads in lockstep.  Same source for both groups -- no duplicated bodies, no spills.
//   quadrants  P1 reads X0 (pixel rows 0..63 of the wave) + W0 (channels 0..31), P2 W1, P3 X1, P4 nothing (W0 is kept):
//              (X0,W0) (X0,W1) (X1,W1) (X1,W0); 24 ds_read_b128 per wave and K-step.
//   staging    a stage is cut in the four quarters the phases consume -- A0 = X0 rows of both groups, B0 = W0 channels of
//              all four wave columns, B1, A1 -- each 16 KiB = 2 DMA pieces per wave.  With tile t (even stage) computed in
//              phases 1-4 and t+1 (odd stage) in 5-8, the quarter issued in phase 1..8 is (t+1).B1, (t+1).A1, (t+2).A0,
//              (t+2).B0, (t+2).B1, (t+2).A1, (t+3).A0, (t+3).B0: every quarter is re-staged >= 2 phases after its last
//              read and has >= 4 phases to land, and "everything but the 4 youngest quarters has landed" (vmcnt(8)) before a
//              phase's first barrier is exactly what the next phase reads.  Tiles past the end are issued out of range
//              (zero fill, no traffic) so that the count stays uniform; an odd number of K-steps computes one zero tile.
//   rows       128-byte LDS rows, logical 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7): source-side for the DMA,
//              conflict-free for the ds_read_b128 lane groups of the 16-row operand tiles.
//   output     the MFMA's A operand is the weight tile, so a lane ends with 4 CONSECUTIVE channels of one pixel: 8-byte
//              stores, float4 bias / scale / shift.
template <typename T, bool ZOUT, bool DECONV>
__global__ __launch_bounds__(512) void conv_fwd_h16p_kernel(const ConvH16Args p) {
    typedef typename H16Traits<T>::v8 v8;
    constexpr int STAGE = 65536, BREG = 32768, PRM = 2 * STAGE, MAXC = 512;
    // ONE LDS object (a second one beside a DMA-filled array makes hipcc drain vmcnt before every operand read): two
    // stages, then bias / scale / shift of all output channels -- the epilogue takes them from LDS because an ordinary
    // global load inside the persistent loop would make the compiler wait for vmcnt(0), i.e. drain the DMA stream
    constexpr int STG = PRM + 3 * MAXC * 4;                     // 2 KiB per wave: one 16-pixel x 64-channel output tile, transposed to rows
    constexpr int TRC = STG + 8 * 2048, NTRC = 512;             // cycle stamps (debug): waves 0 and 4, NTRC each
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE + 3 * MAXC * 4 + 8 * 2048 + 2 * NTRC * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = p.Cout >> 8;
    const int total = p.ptiles;                                 // the host may keep a last partial round for the small-tile kernel
    {   // y = max(acc * P0 + P1, floor), z = acc + P2
        float* prm = (float*)(lds + PRM);
        for (int c = tid; c < (DECONV ? p.cmod : p.Cout); c += 512) {       // DECONV: column (a*2+b)*cmod + c shares channel c's parameters
            const float bi = p.bias ? p.bias[c] : 0.f, sc = p.scale ? p.scale[c] : 1.f, sh = p.scale ? p.shift[c] : 0.f;
            prm[c] = sc;
            prm[MAXC + c] = sc * bi + sh;
            prm[2 * MAXC + c] = bi;
        }
        __syncthreads();
    }
    // taps as bit t = th * KW + tw (KH * KW <= 31): all rows' first columns, for the branch-free validity masks below
    unsigned tap_rows = 0u;
    for (int th = 0; th < p.KH; ++th) tap_rows |= 1u << (th * p.KW);
    const int ohw = p.OH * p.OW;
    const int nk = p.Ktot >> 6, nk2 = (nk + 1) & ~1;            // K-steps of a tile, padded to whole 8-phase rounds

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - p.x_shift), 0, p.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.w_records, 0x00020000);

    // ---- staging bookkeeping: quarter (A: qm, B: qn) x piece j; a piece = 8 tile rows x 128 bytes, lane (l >> 3, l & 7) ----
    // The workgroup is persistent (tiles blockIdx.x, + gridDim.x, ...) and the DMA stream runs on across tile borders:
    // while the last K-steps of a tile are computed the first ones of the next tile are already being staged, and the
    // epilogue's stores drain under the next tile's MFMAs.
    unsigned a_voff[2][2], a_mask[2][2], b_voff[2][2];
    int a_lds[2][2], b_lds[2][2];                               // wave-uniform LDS byte offsets of the pieces inside a stage
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int qr0 = (wave + 8 * j) * 8;                 // first row of the piece inside the quarter (0..120)
            a_lds[q][j] = ((qr0 >> 6) * 128 + q * 64 + (qr0 & 63)) * 128;
            b_lds[q][j] = BREG + ((qr0 >> 5) * 64 + q * 32 + (qr0 & 31)) * 128;
        }
    const int lane_ = lane;
    auto setup = [&](int tile) {                                // per-lane source offsets of output tile `tile` (none: all out of range)
        const int mtile = tile / ntiles, ntile = tile - mtile * ntiles;
        const bool live = tile < total;
        int lane = lane_;
        asm volatile("" : "+v"(lane));                          // as in the epilogue: nothing of this is worth a register in the K loop
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                {
                    const int r = a_lds[q][j] / 128 + (lane >> 3);
                    const int cl = (lane & 7) ^ ((r >> 1) & 7);
                    const int m = mtile * 256 + r;
                    const bool ok = live && m < p.M;
                    const int mm = ok ? m : 0;
                    const int n = p.mg_ohw ? (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw) : mm, rem = mm - n * ohw;
                    const int oh = p.mg_ow ? (int)(__umulhi((unsigned)rem, p.mg_ow) >> p.sh_ow) : rem, ow = rem - oh * p.OW;
                    const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
                    const long long off = ((long long)n * p.H * p.W * p.Cin + ((long long)ih0 * p.W + iw0) * p.Cin + cl * 8) * 2 + p.x_shift;
                    a_voff[q][j] = (unsigned)off;
                    // valid taps: rows th in [rlo, rhi), columns tw in [clo, chi) -- contiguous ranges, so the mask is a product
                    const int rlo = max(0, -ih0), rhi = min(p.KH, p.H - ih0), clo = max(0, -iw0), chi = min(p.KW, p.W - iw0);
                    unsigned mk = 0u;
                    if (ok && rhi > rlo && chi > clo) {
                        const unsigned cols = ((1u << chi) - 1u) & ~((1u << clo) - 1u);
                        const unsigned rows = tap_rows & ((1u << (rhi * p.KW)) - 1u) & ~((1u << (rlo * p.KW)) - 1u);
                        mk = cols * rows;
                    }
                    a_mask[q][j] = mk;
                }
                {
                    const int r = (b_lds[q][j] - BREG) / 128 + (lane >> 3);
                    const int cl = (lane & 7) ^ ((r >> 1) & 7);
                    b_voff[q][j] = live ? (unsigned)((((long long)(ntile * 256 + r)) * p.Ktot + cl * 8) * 2) : H16_OOB_OFFSET;
                }
            }
    };

    // the K-step whose quarters are being issued (A0, B0, B1, A1 in that order, then advance)
    int it_tile = blockIdx.x, it_kh = 0, it_kw = 0, it_tap = 0, it_ci0 = 0, it_kt = 0;
    auto issue_a = [&](auto qc, auto bufc) {
        constexpr int q = decltype(qc)::value;
        char* base = lds + decltype(bufc)::value * STAGE;
        const unsigned soff = (unsigned)(((it_kh * p.W + it_kw) * p.Cin + it_ci0) * 2);
        const unsigned bit = it_kt < nk ? (1u << it_tap) : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = (a_mask[q][j] & bit) ? a_voff[q][j] : H16_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (h16_lds_ptr)(base + a_lds[q][j]), 16, vo, soff, 0, 0);
        }
    };
    auto issue_b = [&](auto qc, auto bufc) {
        constexpr int q = decltype(qc)::value;
        char* base = lds + decltype(bufc)::value * STAGE;
        const unsigned soff = it_kt < nk ? (unsigned)((it_tap * p.Cin + it_ci0) * 2) : 0u;
        const unsigned dead = it_kt < nk ? 0u : H16_OOB_OFFSET;  // or-ed in: a select here becomes a branch around each load
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = b_voff[q][j] | dead;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (h16_lds_ptr)(base + b_lds[q][j]), 16, vo, soff, 0, 0);
        }
    };
    auto advance = [&]() {                                      // channel-chunk outer, filter-tap inner; then the next tile
        if (++it_kt == nk2) {
            it_kt = it_kh = it_kw = it_tap = it_ci0 = 0;
            it_tile += gridDim.x;
            setup(it_tile);
        } else {
            ++it_tap;
            if (++it_kw == p.KW) { it_kw = 0; if (++it_kh == p.KH) { it_kh = 0; it_tap = 0; it_ci0 += 64; } }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand reads: 16-row tile i, k half ks (32 k): lane (l & 15, l >> 4) reads logical chunk 4 ks + (l >> 4) of its row
    const int l15 = lane & 15, fq = lane >> 4;
    const int xrd = (wr * 128 + l15) * 128 + ((fq ^ ((l15 >> 1) & 7)) << 4);
    const int wrd = BREG + (wc * 64 + l15) * 128 + ((fq ^ ((l15 >> 1) & 7)) << 4);
    v8 xf[4][2], w0[2][2], w1[2][2];

    // ---- epilogue of one half of the wave's tile (4 pixel tiles x all 4 channel tiles, complete after phases 6 / 8 of a
    // tile's last K-step): bias, frozen-BN affine, activation in float32, one rounding.  A lane holds 4 consecutive channels
    // of one pixel -- stored like that, every store instruction touches 16 cache lines with 32 bytes each, and the write
    // path, not the arithmetic, set the cost of a tile border (4 400-7 700 cycles per chunk, tools/h16p_trace.py).  So each
    // 16 x 64 tile goes through 2 KiB of LDS (8-byte writes, chunk-swizzled rows) and leaves as whole 128-byte rows: two
    // dwordx4 buffer stores of 8 full lines each (rows past M fall outside the descriptor). ----
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_z = __builtin_amdgcn_make_buffer_rsrc(p.z, 0, ZOUT ? p.out_records : 0u, 0x00020000);
    const float act_floor = p.act == MRCNN_ACT_RELU ? 0.f : -INFINITY;  // no per-element branch on the activation
    constexpr int ES = ZOUT ? 16 : 8;                           // stores per chunk
    typedef T t4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto epilogue = [&](auto halfc, const int tile) {
        constexpr int PB = decltype(halfc)::value * 4;
        const int mtile = tile / ntiles, ntile = tile - mtile * ntiles;
        const int nb = ntile * 256 + wc * 64;
        // DECONV (MRCNN_OUT_DECONV2): the 64 columns are channels cnb .. cnb + 63 of output pixel (2 oh + a, 2 ow + b),
        // (a, b) = ab >> 1, ab & 1 -- still one 128-byte row per input pixel, only the row's address changes
        const int ab = DECONV ? nb / p.cmod : 0, cnb = DECONV ? nb - ab * p.cmod : nb;
        char* stg = lds + STG + wave * 2048;
        int ln = lane;
        asm volatile("" : "+v"(ln));                            // per-call lane arithmetic: hoisted out of the K loop it costs registers there
        const int l15 = ln & 15, fq = ln >> 4;
        const int wbase = l15 * 128 + (fq & 1) * 8, wsw = l15 & 7;
        const int r8 = ln >> 3;
        const int rbase = r8 * 128 + (((ln & 7) ^ (r8 & 7)) << 4);
        const unsigned voff = (unsigned)(((mtile * 256 + wr * 128 + r8) * p.Cout + nb) * 2 + (ln & 7) * 16);
        auto deconv_off = [&](int m) -> unsigned {              // byte offset of input pixel m's row of this chunk
            if (m >= p.M) return H16_OOB_OFFSET;
            const unsigned n = p.mg_ohw ? (__umulhi((unsigned)m, p.mg_ohw) >> p.sh_ohw) : (unsigned)m;
            const unsigned rem = (unsigned)m - n * (unsigned)ohw;
            const unsigned oh = p.mg_ow ? (__umulhi(rem, p.mg_ow) >> p.sh_ow) : rem;
            const unsigned ow = rem - oh * (unsigned)p.OW;
            return (n * (unsigned)p.ons + (2u * oh + (unsigned)(ab >> 1)) * (unsigned)p.ohs + (2u * ow + (unsigned)(ab & 1)) * (unsigned)p.ows +
                    (unsigned)cnb) * 2u + (unsigned)(ln & 7) * 16u;
        };
        f32x4 p0[4], p1[4], p2[4];                              // this lane's 16 channels: loaded once, not per pixel tile
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = cnb + c * 16 + fq * 4;
            p0[c] = *(const f32x4*)(lds + PRM + n * 4);
            p1[c] = *(const f32x4*)(lds + PRM + (MAXC + n) * 4);
            if constexpr (ZOUT) p2[c] = *(const f32x4*)(lds + PRM + (2 * MAXC + n) * 4);
        }
        // software pipeline over the 4 pixel tiles: LDS executes a wave's operations in order, so "write tile i, read it
        // back as rows, write tile i + 1, ..." needs no wait between them; the rows of tile i are stored while tile i + 1
        // is converted (a single wave has nobody else to hide the LDS round trip behind)
        u32x4 o0[4], o1[4], z0[4], z1[4];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i < 4) {
                t4 yv[4], zv4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) yv[c][j] = (T)fmaxf(acc[PB + i][c][j] * p0[c][j] + p1[c][j], act_floor);  // RELU / NONE only (the host checks)
                    if constexpr (ZOUT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) zv4[c][j] = (T)(acc[PB + i][c][j] + p2[c][j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[PB + i][c][j] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) *(t4*)(stg + wbase + (((c * 2 + (fq >> 1)) ^ wsw) << 4)) = yv[c];
                o0[i] = *(const u32x4*)(stg + rbase); o1[i] = *(const u32x4*)(stg + rbase + 1024);
                if constexpr (ZOUT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) *(t4*)(stg + wbase + (((c * 2 + (fq >> 1)) ^ wsw) << 4)) = zv4[c];
                    z0[i] = *(const u32x4*)(stg + rbase); z1[i] = *(const u32x4*)(stg + rbase + 1024);
                }
            }
            if (i > 0) {
                unsigned soff = (unsigned)(((PB + i - 1) * 16 * p.Cout) * 2), soff8 = soff + (unsigned)(8 * p.Cout * 2);
                unsigned vo0 = voff, vo8 = voff;
                if constexpr (DECONV) {
                    const int mrow = mtile * 256 + wr * 128 + (PB + i - 1) * 16 + r8;
                    vo0 = deconv_off(mrow); vo8 = deconv_off(mrow + 8);
                    soff = 0u; soff8 = 0u;
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_buffer_store_b128(o0[i - 1], rsrc_o, vo0, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1[i - 1], rsrc_o, vo8, soff8, 0);
                if constexpr (ZOUT) {
                    __builtin_amdgcn_raw_buffer_store_b128(z0[i - 1], rsrc_z, vo0, soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(z1[i - 1], rsrc_z, vo8, soff8, 0);
                }
                asm volatile("s_nop 2");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    int ntrc = 0;
    auto phase = [&](auto phc, const int t, const int tile) {
        constexpr int PH = decltype(phc)::value;                // 0..7
        constexpr int BUF = PH >> 2, Q = PH & 3;
        const char* sb = lds + BUF * STAGE;
        auto stamp = [&]() {
            if (p.dbg && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && ntrc < NTRC)
                ((unsigned long long*)(lds + TRC))[wr * NTRC + ntrc++] = __builtin_readcyclecounter();
        };
        stamp();
        // the epilogue of the half tile the previous two phases completed, in the last K-step of a tile (pixel tiles 0..3) and
        // in the first phase after it (4..7): here it runs beside the OTHER group's MFMAs
        const bool last = t + 2 >= nk2, after = t == 0 && tile != (int)blockIdx.x;
        if constexpr (PH == 6) { if (last) epilogue(I0{}, tile); }
        if constexpr (PH == 0) { if (after) epilogue(I1{}, tile - (int)gridDim.x); }
        __builtin_amdgcn_sched_barrier(0);                           // epilogue registers are free before the operands load
        if constexpr (Q == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w0[i][ks] = *(const v8*)(sb + (wrd ^ (ks << 6)) + i * 2048);
        }
        if constexpr (Q == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w1[i][ks] = *(const v8*)(sb + (wrd ^ (ks << 6)) + 4096 + i * 2048);
        }
        if constexpr (Q == 0 || Q == 2) {
            if constexpr (Q == 0) __builtin_amdgcn_sched_barrier(0);     // the 4 W reads first: they retire first
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) xf[i][ks] = *(const v8*)(sb + (xrd ^ (ks << 6)) + (Q == 2 ? 8192 : 0) + i * 2048);
        }
        // one quarter of a later stage
        if constexpr (PH == 0) issue_b(I1{}, I1{});
        if constexpr (PH == 1) { issue_a(I1{}, I1{}); advance(); }
        if constexpr (PH == 2) issue_a(I0{}, I0{});
        if constexpr (PH == 3) issue_b(I0{}, I0{});
        if constexpr (PH == 4) issue_b(I1{}, I0{});
        if constexpr (PH == 5) { issue_a(I1{}, I0{}); advance(); }
        if constexpr (PH == 6) issue_a(I0{}, I1{});
        if constexpr (PH == 7) issue_b(I0{}, I1{});
        // "everything but the 4 youngest quarters": the epilogue chunks' stores (ES each, issued before this phase's DMA in
        // phase 7 of the last K-step and in phase 1 after it) count too while they are among the 4 youngest phases
        if constexpr (PH >= 6) { if (last) h16p_wait<8 + ES>(); else h16p_wait<8>(); }
        else if constexpr (PH <= 1) { if (after) h16p_wait<8 + 2 * ES>(); else h16p_wait<8>(); }
        else if constexpr (PH <= 3) { if (after) h16p_wait<8 + ES>(); else h16p_wait<8>(); }
        else h16p_wait<8>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        constexpr int PB = (Q >= 2) ? 4 : 0;                     // pixel tiles of this quadrant
        constexpr int CB = (Q == 1 || Q == 2) ? 2 : 0;           // channel tiles of this quadrant
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[PB + i][CB + c] = H16Traits<T>::mfma16((CB ? w1 : w0)[c][ks], xf[i][ks], acc[PB + i][CB + c]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // prologue: K-step 0 whole, K-step 1's A0 and B0 (the state every 8-phase round starts from); group 1 drops one barrier behind
    setup(it_tile);
    issue_a(I0{}, I0{}); issue_b(I0{}, I0{}); issue_b(I1{}, I0{}); issue_a(I1{}, I0{}); advance();
    issue_a(I0{}, I1{}); issue_b(I0{}, I1{});
    if (wr == 1) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();

    int tile = blockIdx.x;
    for (; tile < total; tile += gridDim.x)
        for (int t = 0; t < nk2; t += 2)
            h16_static_for([&](auto phc) { phase(phc, t, tile); }, std::make_integer_sequence<int, 8>{});
    epilogue(I1{}, tile - (int)gridDim.x);                                  // grid <= tiles: every workgroup had one
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the out-of-range tail DMAs still target this workgroup's LDS
    if (p.dbg && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0)
        for (int i = 0; i < NTRC; ++i)
            p.dbg[wr * NTRC + i] = i < ntrc ? ((unsigned long long*)(lds + TRC))[wr * NTRC + i] : 0ull;
}

// ---- slab form of the phased kernel (round 3; 3 x 3, stride 1, pad 1 on maps with W <= 14: the mask head) --------------------
// MEASURED SLOWER than conv_fwd_h16p_kernel and therefore opt-in (MRCNN_H16_SLAB=1 / mrcnn_tuning_set("h16_slab", 1)); kept with its
// tests as the evidence that the number of LDS-DMA pieces is NOT what bounds the phased kernel.
// The phased kernel re-stages the pixels of a channel chunk for each of the nine taps: 1 152 of a tile's 2 304 LDS-DMA pieces,
// and what bounds it is the issue cost of those pieces.  Here a channel chunk's pixels are staged ONCE, as a slab
// of the tile's 256 pixel rows with a halo of 15 rows on either side (288 rows of 128 bytes, rows in the flat pixel order of the
// [M] tensor), and the nine taps are SHIFTED READS of it: tap (th, tw) reads row 15 + t + (th - 1) W + (tw - 1) for pixel t.  A
// pixel whose tap leaves its map (border of the 14 x 14 map, tile tail) reads a zero row instead: per lane a 9-bit mask for each of
// its eight 16-row operand tiles, made once per tile.  Rows are chunk-swizzled as before, c ^ ((row >> 1) & 7); a shift changes
// the row and with it the swizzle, but the term is the same for all eight tiles of a lane (they are 16 rows apart), so a tap costs
// one address per k half plus one select per tile.  Per K-step a wave issues 4 weight pieces + on average 0.5 slab pieces
// instead of 8; the weight stages, the two staggered wave groups, the barriers, the MFMA quadrants and the epilogue are the
// phased kernel's.  Two slabs (the next chunk's is staged five K-steps ahead, while this chunk's taps are computed), two 32 KiB
// weight stages, parameters, store staging and the zero row: 161 920 bytes of LDS.
//   waits     only weight quarters are counted: the quarter a phase reads next was issued 5 phases earlier and two quarters
//             (4 pieces) have been issued since -- s_waitcnt vmcnt(4) (+ the epilogue's stores while they are in that window); a
//             slab piece in the window only makes the wait stricter by one, and a chunk's slab is >= 20 pieces old at its first read
template <typename T, bool ZOUT>
__global__ __launch_bounds__(512) void conv_fwd_h16q_kernel(const ConvH16Args p) {
    typedef typename H16Traits<T>::v8 v8;
    constexpr int HALO = 15, SROWS = 288, SLAB = SROWS * 128, NPIECE = SROWS / 8;
    constexpr int BST = 2 * SLAB, BSZ = 32768, PRM = BST + 2 * BSZ, MAXC = 512;
    constexpr int STG = PRM + 3 * MAXC * 4, ZROW = STG + 8 * 2048;
    __shared__ __attribute__((aligned(16))) char lds[ZROW + 128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = p.Cout >> 8;
    const int total = p.ptiles;
    {
        float* prm = (float*)(lds + PRM);
        for (int c = tid; c < p.Cout; c += 512) {
            const float bi = p.bias ? p.bias[c] : 0.f, sc = p.scale ? p.scale[c] : 1.f, sh = p.scale ? p.shift[c] : 0.f;
            prm[c] = sc;
            prm[MAXC + c] = sc * bi + sh;
            prm[2 * MAXC + c] = bi;
        }
        if (tid < 32) ((float*)(lds + ZROW))[tid] = 0.f;
        __syncthreads();
    }
    const int ohw = p.OH * p.OW;
    const int nk = p.Ktot >> 6;                                 // 9 taps x Cin / 64 chunks; the host checks Cin % 128 == 0: nk is even

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - p.x_shift), 0, p.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.w_records, 0x00020000);

    // ---- staging bookkeeping ----------------------------------------------------------------------------------------------
    unsigned b_voff[2][2];
    int b_lds[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int qr0 = (wave + 8 * j) * 8;
            b_lds[q][j] = ((qr0 >> 5) * 64 + q * 32 + (qr0 & 31)) * 128;
        }
    const int lane_ = lane;
    auto setup_b = [&](int tile) {
        const int mtile = tile / ntiles, ntile = tile - mtile * ntiles;
        const bool live = tile < total;
        int lane = lane_;
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = b_lds[q][j] / 128 + (lane >> 3);
                const int cl = (lane & 7) ^ ((r >> 1) & 7);
                b_voff[q][j] = live ? (unsigned)((((long long)(ntile * 256 + r)) * p.Ktot + cl * 8) * 2) : H16_OOB_OFFSET;
            }
    };
    // slab pieces: piece pi covers slab rows 8 pi .. 8 pi + 7 = flat pixels mtile * 256 - 15 + 8 pi ..; lane (l >> 3, l & 7) fetches
    // logical chunk (l & 7) ^ ((row >> 1) & 7) of its row.  Nothing per piece is kept in registers: the offset is the lane's
    // constant part plus a wave-uniform pixel base, made when the piece is issued (one piece per K-step).
    auto slab_piece = [&](const int j, const int buf, const unsigned soff, const int mtile, const bool live) {
        const int pi = wave + 8 * j;                            // wave-uniform: pieces 36 .. 39 do not exist
        if (pi < NPIECE) {
            int ln = lane_;
            asm volatile("" : "+v"(ln));                        // recomputed per piece: a register kept across the K loop would spill (ZOUT form)
            const int g = mtile * 256 - HALO + pi * 8 + (ln >> 3);
            const bool ok = live && g >= 0 && g < p.M;
            const unsigned cl = (unsigned)((ln & 7) ^ (((ln >> 4) + 4 * (wave & 1)) & 7));
            const unsigned vo = ok ? p.x_shift + cl * 16u + (unsigned)g * (unsigned)(p.Cin * 2) : H16_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (h16_lds_ptr)(lds + buf * SLAB + pi * 1024), 16, vo, soff, 0, 0);
        }
    };

    int it_tile = blockIdx.x, it_tap = 0, it_ci0 = 0, it_kt = 0;
    auto issue_b = [&](auto qc, auto bufc) {
        constexpr int q = decltype(qc)::value;
        char* base = lds + BST + decltype(bufc)::value * BSZ;
        const unsigned soff = it_kt < nk ? (unsigned)((it_tap * p.Cin + it_ci0) * 2) : 0u;
        const unsigned dead = it_kt < nk ? 0u : H16_OOB_OFFSET;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = b_voff[q][j] | dead;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (h16_lds_ptr)(base + b_lds[q][j]), 16, vo, soff, 0, 0);
        }
    };
    int it_mtile = (int)blockIdx.x / ntiles, nx_mtile = ((int)blockIdx.x + (int)gridDim.x) / ntiles;      // row tile of the issue side's tile / of the next one
    auto issue_slab = [&]() {                                   // during taps 2 .. 6 of a chunk: one piece of the NEXT chunk's slab
        if (it_kt < nk && it_tap >= 2 && it_tap <= 6) {
            const bool wrap = it_ci0 + 64 == p.Cin;                 // the next chunk is chunk 0 of this workgroup's next tile
            const unsigned soff = wrap ? 0u : (unsigned)((it_ci0 + 64) * 2);
            slab_piece(it_tap - 2, ((it_ci0 >> 6) & 1) ^ 1, soff, wrap ? nx_mtile : it_mtile,
                       wrap ? it_tile + (int)gridDim.x < total : it_tile < total);
        }
    };
    auto advance = [&]() {
        if (++it_kt == nk) {
            it_kt = it_tap = it_ci0 = 0;
            it_tile += gridDim.x;
            it_mtile = nx_mtile;
            nx_mtile = (it_tile + (int)gridDim.x) / ntiles;
            setup_b(it_tile);
        } else if (++it_tap == 9) {
            it_tap = 0;
            it_ci0 += 64;
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, fq = lane >> 4;
    const int wrd = BST + (wc * 64 + l15) * 128 + ((fq ^ ((l15 >> 1) & 7)) << 4);
    v8 xf[4][2], w0[2][2], w1[2][2];
    unsigned vm0 = 0u, vm1 = 0u, vm2 = 0u;                      // tap masks of this lane's pixel in operand tiles 0-2, 3-5, 6-7 (9 bits each)

    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_z = __builtin_amdgcn_make_buffer_rsrc(p.z, 0, ZOUT ? p.out_records : 0u, 0x00020000);
    const float act_floor = p.act == MRCNN_ACT_RELU ? 0.f : -INFINITY;
    constexpr int ES = ZOUT ? 16 : 8;
    typedef T t4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto epilogue = [&](auto halfc, const int tile) {           // conv_fwd_h16p_kernel's, dense NHWC only
        constexpr int PB = decltype(halfc)::value * 4;
        const int mtile = tile / ntiles, ntile = tile - mtile * ntiles;
        const int nb = ntile * 256 + wc * 64;
        char* stg = lds + STG + wave * 2048;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int l15 = ln & 15, fq = ln >> 4;
        const int wbase = l15 * 128 + (fq & 1) * 8, wsw = l15 & 7;
        const int r8 = ln >> 3;
        const int rbase = r8 * 128 + (((ln & 7) ^ (r8 & 7)) << 4);
        const unsigned voff = (unsigned)(((mtile * 256 + wr * 128 + r8) * p.Cout + nb) * 2 + (ln & 7) * 16);
        f32x4 p0[4], p1[4], p2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = nb + c * 16 + fq * 4;
            p0[c] = *(const f32x4*)(lds + PRM + n * 4);
            p1[c] = *(const f32x4*)(lds + PRM + (MAXC + n) * 4);
            if constexpr (ZOUT) p2[c] = *(const f32x4*)(lds + PRM + (2 * MAXC + n) * 4);
        }
        u32x4 o0[4], o1[4], z0[4], z1[4];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i < 4) {
                t4 yv[4], zv4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) yv[c][j] = (T)fmaxf(acc[PB + i][c][j] * p0[c][j] + p1[c][j], act_floor);
                    if constexpr (ZOUT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) zv4[c][j] = (T)(acc[PB + i][c][j] + p2[c][j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[PB + i][c][j] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) *(t4*)(stg + wbase + (((c * 2 + (fq >> 1)) ^ wsw) << 4)) = yv[c];
                o0[i] = *(const u32x4*)(stg + rbase); o1[i] = *(const u32x4*)(stg + rbase + 1024);
                if constexpr (ZOUT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) *(t4*)(stg + wbase + (((c * 2 + (fq >> 1)) ^ wsw) << 4)) = zv4[c];
                    z0[i] = *(const u32x4*)(stg + rbase); z1[i] = *(const u32x4*)(stg + rbase + 1024);
                }
            }
            if (i > 0) {
                const unsigned soff = (unsigned)(((PB + i - 1) * 16 * p.Cout) * 2), soff8 = soff + (unsigned)(8 * p.Cout * 2);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_raw_buffer_store_b128(o0[i - 1], rsrc_o, voff, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1[i - 1], rsrc_o, voff, soff8, 0);
                if constexpr (ZOUT) {
                    __builtin_amdgcn_raw_buffer_store_b128(z0[i - 1], rsrc_z, voff, soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(z1[i - 1], rsrc_z, voff, soff8, 0);
                }
                asm volatile("s_nop 2");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // tap masks of the compute side's tile: bit t = th * 3 + tw set when pixel (oh, ow)'s tap stays inside the map and the row is < M
    auto tile_masks = [&](const int tile) {
        const int mtile = tile / ntiles;
        int ln = lane_;
        asm volatile("" : "+v"(ln));
        auto one = [&](const int gi) -> unsigned {
            const int m = mtile * 256 + wr * 128 + 16 * gi + (ln & 15);
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int n = p.mg_ohw ? (int)(__umulhi((unsigned)mm, p.mg_ohw) >> p.sh_ohw) : mm, rem = mm - n * ohw;
            const int oh = p.mg_ow ? (int)(__umulhi((unsigned)rem, p.mg_ow) >> p.sh_ow) : rem, ow = rem - oh * p.OW;
            const unsigned rows = (oh > 0 ? 0x007u : 0u) | 0x038u | (oh < p.H - 1 ? 0x1C0u : 0u);     // tap rows th = 0 / 1 / 2
            const unsigned cols = (ow > 0 ? 0x049u : 0u) | 0x092u | (ow < p.W - 1 ? 0x124u : 0u);     // tap columns tw = 0 / 1 / 2
            return ok ? (rows & cols) : 0u;
        };
        vm0 = one(0) | (one(1) << 9) | (one(2) << 18);
        vm1 = one(3) | (one(4) << 9) | (one(5) << 18);
        vm2 = one(6) | (one(7) << 9);
    };

    auto phase = [&](auto phc, const int t, const int tile) {
        constexpr int PH = decltype(phc)::value;                // 0..7
        constexpr int BUF = PH >> 2, Q = PH & 3;
        const char* sb = lds + BUF * BSZ;                        // weight stage (wrd carries BST)
        const bool last = t + 2 >= nk, after = t == 0 && tile != (int)blockIdx.x;
        if constexpr (PH == 6) { if (last) epilogue(I0{}, tile); }
        if constexpr (PH == 0) { if (after) epilogue(I1{}, tile - (int)gridDim.x); }
        if constexpr (PH == 0) { if (t == 0) tile_masks(tile); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (Q == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w0[i][ks] = *(const v8*)(sb + (wrd ^ (ks << 6)) + i * 2048);
        }
        if constexpr (Q == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w1[i][ks] = *(const v8*)(sb + (wrd ^ (ks << 6)) + 4096 + i * 2048);
        }
        if constexpr (Q == 0 || Q == 2) {
            if constexpr (Q == 0) __builtin_amdgcn_sched_barrier(0);
            // this K-step's tap as a shifted read of the chunk's slab
            const int kstep = t + BUF;
            const int chunk = kstep / 9, tap = kstep - chunk * 9;
            const int th = tap / 3, tw = tap - th * 3;
            const int shift = (th - 1) * p.W + (tw - 1);
            int ln = lane_;
            asm volatile("" : "+v"(ln));
            const int r0 = HALO + wr * 128 + (ln & 15) + shift;                       // slab row of operand tile 0's pixel (tiles: + 16 i)
            const int a0 = (chunk & 1) * SLAB + r0 * 128 + ((((ln >> 4)) ^ ((r0 >> 1) & 7)) << 4) + (Q == 2 ? 8192 : 0);
            const int a1 = a0 ^ 64;
            const int zr = ZROW + ((ln >> 4) << 4);
            if (p.dbg) {                                        // TIMING EXPERIMENT ONLY (MRCNN_H16P_TRACE set): no border masks -- wrong at map borders
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xf[i][0] = *(const v8*)(lds + a0 + i * 2048);
                    xf[i][1] = *(const v8*)(lds + a1 + i * 2048);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    constexpr int GI0 = (Q == 2 ? 4 : 0);
                    const int gi = GI0 + i;
                    const unsigned word = gi < 3 ? vm0 : (gi < 6 ? vm1 : vm2);
                    const bool ok = ((word >> (9 * (gi % 3) + tap)) & 1u) != 0u;
                    xf[i][0] = *(const v8*)(lds + (ok ? a0 + i * 2048 : zr));
                    xf[i][1] = *(const v8*)(lds + (ok ? a1 + i * 2048 : zr));
                }
            }
        }
        // staging: weight quarters as in the phased kernel; where that kernel staged pixel quarters, at most one slab piece
        if constexpr (PH == 0) issue_b(I1{}, I1{});
        if constexpr (PH == 1) advance();
        if constexpr (PH == 2) issue_slab();
        if constexpr (PH == 3) issue_b(I0{}, I0{});
        if constexpr (PH == 4) issue_b(I1{}, I0{});
        if constexpr (PH == 5) advance();
        if constexpr (PH == 6) issue_slab();
        if constexpr (PH == 7) issue_b(I0{}, I1{});
        if constexpr (PH == 7) { if (last) h16p_wait<4 + ES>(); else h16p_wait<4>(); }
        else if constexpr (PH == 0) { if (after) h16p_wait<4 + 2 * ES>(); else h16p_wait<4>(); }
        else if constexpr (PH == 3) { if (after) h16p_wait<4 + ES>(); else h16p_wait<4>(); }
        else if constexpr (PH == 4) h16p_wait<4>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        constexpr int PB = (Q >= 2) ? 4 : 0;
        constexpr int CB = (Q == 1 || Q == 2) ? 2 : 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[PB + i][CB + c] = H16Traits<T>::mfma16((CB ? w1 : w0)[c][ks], xf[i][ks], acc[PB + i][CB + c]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // prologue: the first chunk's slab whole, K-step 0's weight quarters, K-step 1's first; group 1 drops one barrier behind
    setup_b(it_tile);
#pragma unroll
    for (int j = 0; j < 5; ++j) slab_piece(j, 0, 0u, it_mtile, it_tile < total);
    issue_b(I0{}, I0{}); issue_b(I1{}, I0{}); advance();
    issue_b(I0{}, I1{});
    if (wr == 1) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();

    int tile = blockIdx.x;
    for (; tile < total; tile += gridDim.x)
        for (int t = 0; t < nk; t += 2)
            h16_static_for([&](auto phc) { phase(phc, t, tile); }, std::make_integer_sequence<int, 8>{});
    epilogue(I1{}, tile - (int)gridDim.x);
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Small-tile variant for the layers of the trunk (feature maps of 1 024 .. 16 384 pixels, Cin / Cout multiples of 64):
// 64 x 64 output tile, 4 waves of 32 x 32 (one MFMA tile each), K-step 64, a ring of four 16 KiB stages filled by LDS-DMA
// three steps ahead (counted vmcnt, raw barrier), two workgroups per CU.  No split-K and no second launch: a layer with
// M = 4 096, N = 256 is 256 workgroups whose K loop is 16-36 short steps -- what bounds these layers is the length of the
// dependent chain per workgroup, not the matrix rate.  LDS rows are 128 bytes (64 k); logical 16-byte chunk c of row r
// lives at chunk c ^ ((r >> 1) & 7) (rows r, r + 1 share a chunk but sit in different bank halves): applied on the DMA's
// source side and in the operand reads, conflict-free for the ds_read_b128 lane groups.
// Epilogue: bias, frozen-BN affine, optional 16-bit residual (a bottleneck block's shortcut, or the accumulating input of
// a data gradient), activation; output strides as in mrcnn_conv_desc (a stride-2 data gradient scatters into a zeroed
// tensor).
template <typename T>
__global__ __launch_bounds__(256, 2) void conv_fwd_h16s_kernel(const ConvH16Args p) {
    typedef typename H16Traits<T>::v8 v8;
    constexpr int BM = 64, BN = 64, NBUF = 4, D = 3;
    constexpr int ROWB = 128;                                   // bytes per LDS row (64 x 16-bit)
    constexpr int AB = BM * ROWB, BB = BN * ROWB;               // 8 KiB + 8 KiB per stage
    __shared__ __attribute__((aligned(16))) char lds[NBUF * (AB + BB)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.Cout / BN;
    const int mtile = blockIdx.x / ntiles + p.mtile0, ntile = blockIdx.x % ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int ohw = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - p.x_shift), 0, p.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.w_records, 0x00020000);

    // a 1 KiB DMA piece = 8 rows of 128 bytes: lane -> row (lane >> 3), physical chunk (lane & 7).  8 pieces per operand,
    // this wave stages pieces wave and wave + 4 of each.
    unsigned a_voff[2], b_voff[2];
    unsigned long long a_mask[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int r = (wave + jj * 4) * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / ohw, rem = mm - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        a_voff[jj] = (unsigned)(((long long)n * p.H * p.W * p.Cin + ((long long)ih0 * p.W + iw0) * p.Cin + cl * 8) * 2 + p.x_shift);
        unsigned long long mk = 0ull;
        if (ok)
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int th = t / p.KW, tw = t - th * p.KW;
                if ((unsigned)(ih0 + th) < (unsigned)p.H && (unsigned)(iw0 + tw) < (unsigned)p.W) mk |= 1ull << t;
            }
        a_mask[jj] = mk;
        b_voff[jj] = (unsigned)((((long long)(n0 + r)) * p.Ktot + cl * 8) * 2);
    }

    int kh = 0, kw = 0, ci0 = 0, tap = 0;
    auto stage = [&](char* ab) {
        char* bb = ab + AB;
        const unsigned soff_a = (unsigned)(((kh * p.W + kw) * p.Cin + ci0) * 2);
        const unsigned soff_b = (unsigned)((tap * p.Cin + ci0) * 2);
        const unsigned long long bit = 1ull << tap;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const unsigned vo = (a_mask[jj] & bit) ? a_voff[jj] : H16_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (h16_lds_ptr)(ab + (wave + jj * 4) * 1024), 16, vo, soff_a, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (h16_lds_ptr)(bb + (wave + jj * 4) * 1024), 16, b_voff[jj], soff_b, 0, 0);
        ++tap;                                                  // channel-chunk outer, filter-tap inner
        if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; tap = 0; ci0 += 64; } }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    // operand k-group kk (16 k values) of a 32-row tile: lane (li, lh) reads logical chunk 2*kk + lh of its row
    const int arow = wm * 32 + li, brow = wn * 32 + li;
    int a_rd[4], b_rd[4];                                       // byte offsets inside a stage
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_rd[kk] = arow * ROWB + (((2 * kk + lh) ^ ((arow >> 1) & 7)) << 4);
        b_rd[kk] = AB + brow * ROWB + (((2 * kk + lh) ^ ((brow >> 1) & 7)) << 4);
    }
    auto compute = [&](auto curc) {
        const char* buf = lds + decltype(curc)::value * (AB + BB);
        v8 av[4], bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { av[kk] = *(const v8*)(buf + a_rd[kk]); bv[kk] = *(const v8*)(buf + b_rd[kk]); }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = H16Traits<T>::mfma(av[kk], bv[kk], acc);
    };

    const int nk = p.Ktot / 64;
    for (int s0 = 0; s0 < D && s0 < nk; ++s0) stage(lds + s0 * (AB + BB));
    for (int ks0 = 0; ks0 < nk; ks0 += NBUF) {
        h16_static_for([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            const int ks = ks0 + S;
            if (ks < nk) {
                const int younger = nk - 1 - ks;                // stages issued after ks that may still be in flight (<= D - 1)
                if (younger >= 2) h16_wait_vmcnt<8>();          // 4 DMA instructions per stage and wave
                else if (younger >= 1) h16_wait_vmcnt<4>();
                else h16_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();                   // stage ks has landed for every wave; everyone is done with stage ks - 1
                if (ks + D < nk) stage(lds + ((S + D) % NBUF) * (AB + BB));
                compute(std::integral_constant<int, S>{});
            }
        }, std::make_integer_sequence<int, NBUF>{});
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------
    T* out = (T*)p.out;
    T* zo = (T*)p.z;
    const T* res = (const T*)p.res;
    if (p.ep_vec) {
        // The accumulators leave through LDS: the 64 x 64 tile is staged in float32 (rows of 68 floats) and every thread finishes
        // two 8-channel pieces of a pixel -- 16-byte loads of the residual / the layer below's tensors, 16-byte stores, a pixel's
        // 128 bytes by 8 neighbouring lanes -- instead of sixteen 2-byte accesses per lane that touch 64 bytes per row each.  The
        // expand layers of a bottleneck block (1 x 1 to 1024 channels: K is 4 steps, the tile's 8 KiB of output and 8 KiB of
        // residual are the work) were 13.4 us for M = 4096 alone against ~6 us of memory time.
        typedef T t8 __attribute__((ext_vector_type(8)));
        constexpr int SST = 68;
        __syncthreads();                                        // every wave is done with the ring
        float* stg = (float*)lds;                               // [64][SST]
        float* red = stg + 64 * SST;                            // [64 columns][3] (data gradient: channel sums)
        {
            const int rb = wm * 32 + 4 * lh, cb = wn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[(rb + (r & 3) + 8 * (r >> 2)) * SST + cb] = acc[r];
        }
        if (tid < 192) red[tid] = 0.f;
        __syncthreads();
        const int c8 = (tid & 7) * 8, nn = n0 + c8;
        if (p.fb_act < 0) {
            float cbv[8], csv[8], chv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cbv[e] = p.bias ? p.bias[nn + e] : 0.f;
                csv[e] = p.scale ? p.scale[nn + e] : 1.f;
                chv[e] = p.scale ? p.shift[nn + e] : 0.f;
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = (tid >> 3) + 32 * ps, m = m0 + row;
                if (m >= p.M) continue;
                long long addr;
                if (p.dense) {
                    addr = (long long)m * p.Cout + nn;
                } else {
                    const int ni = m / ohw, rem = m - ni * ohw;
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    addr = (long long)ni * p.ons + (long long)oh * p.ohs + (long long)ow * p.ows + nn;
                }
                const f32x4 v0 = *(const f32x4*)&stg[row * SST + c8], v1 = *(const f32x4*)&stg[row * SST + c8 + 4];
                t8 rv;
                if (res) rv = *(const t8*)(res + addr);
                t8 yo, zv8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float zv = (e < 4 ? v0[e & 3] : v1[e & 3]) + cbv[e];
                    zv8[e] = (T)zv;
                    float y = csv[e] * zv + chv[e];
                    if (res) y += (float)rv[e];
                    if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
                    else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
                    yo[e] = (T)y;
                }
                if (zo) *(t8*)(zo + addr) = zv8;
                *(t8*)(out + addr) = yo;
            }
            return;
        }
        // data gradient fused with the epilogue backward of the layer below (dense output)
        const T* bo = (const T*)p.fb_out;
        const T* bz = (const T*)p.fb_z;
        T* dyo = (T*)p.fb_dy;
        float sc[8], mu[8], rs[8], s0[8], s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = p.fb_scale ? p.fb_scale[nn + e] : 1.f;
            mu[e] = p.fb_dgamma ? p.fb_mean[nn + e] : 0.f;
            rs[e] = p.fb_dgamma ? p.fb_rstd[nn + e] : 0.f;
            s0[e] = s1[e] = s2[e] = 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = (tid >> 3) + 32 * ps, m = m0 + row;
            if (m >= p.M) continue;
            const long long addr = (long long)m * p.Cout + nn;
            const f32x4 v0 = *(const f32x4*)&stg[row * SST + c8], v1 = *(const f32x4*)&stg[row * SST + c8 + 4];
            t8 rv, bov, bzv;
            if (res) rv = *(const t8*)(res + addr);
            if (p.fb_act == MRCNN_ACT_RELU) bov = *(const t8*)(bo + addr);
            if (p.fb_dgamma) bzv = *(const t8*)(bz + addr);
            t8 dz8, g8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float g = e < 4 ? v0[e & 3] : v1[e & 3];
                if (res) g += (float)rv[e];
                if (p.fb_act == MRCNN_ACT_RELU) g = (float)bov[e] > 0.f ? g : 0.f;
                const float dz = g * sc[e];
                dz8[e] = (T)dz; g8[e] = (T)g;
                s0[e] += g;
                if (p.fb_dgamma) s1[e] += g * ((float)bzv[e] - mu[e]) * rs[e];
                s2[e] += dz;
            }
            *(t8*)(out + addr) = dz8;
            if (dyo) *(t8*)(dyo + addr) = g8;
        }
        // lanes l, l + 8, ..., l + 56 of a wave own the same 8 channels: fold them, then one LDS atomic per wave and value
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int d = 8; d < 64; d <<= 1) {
                s0[e] += __shfl_xor(s0[e], d, 64); s1[e] += __shfl_xor(s1[e], d, 64); s2[e] += __shfl_xor(s2[e], d, 64);
            }
        }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(&red[(c8 + e) * 3 + 0], s0[e]); atomicAdd(&red[(c8 + e) * 3 + 1], s1[e]); atomicAdd(&red[(c8 + e) * 3 + 2], s2[e]);
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int c = n0 + tid;
            const float gm = p.fb_gmul;
            if (p.fb_dbeta) atomicAdd(p.fb_dbeta + c, red[tid * 3 + 0] * gm);
            if (p.fb_dgamma) atomicAdd(p.fb_dgamma + c, red[tid * 3 + 1] * gm);
            if (p.fb_dbias) atomicAdd(p.fb_dbias + c, red[tid * 3 + 2] * gm);
        }
        return;
    }
    const int n = n0 + wn * 32 + li;
    if (p.fb_act >= 0) {
        // data gradient: y = acc (+ res) is d(loss)/d(activated output of the layer below), times the loss scale.
        //   g = y * act'(out_below);  dz = g * scale_below -> stored (and g as dy when asked);
        //   dbeta += sum g, dgamma += sum g (z - mean) rstd, dbias += sum dz   (times fb_gmul = 1 / loss scale)
        const T* bo = (const T*)p.fb_out;
        const T* bz = (const T*)p.fb_z;
        T* dyo = (T*)p.fb_dy;
        const float sc = p.fb_scale ? p.fb_scale[n] : 1.f;
        const float mu = p.fb_dgamma ? p.fb_mean[n] : 0.f, rs = p.fb_dgamma ? p.fb_rstd[n] : 0.f;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        const int mb = m0 + wm * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m >= p.M) continue;
            const long long addr = (long long)m * p.Cout + n;
            float g = acc[r];
            if (res) g += (float)res[addr];
            if (p.fb_act == MRCNN_ACT_RELU) g = (float)bo[addr] > 0.f ? g : 0.f;
            const float dz = g * sc;
            out[addr] = (T)dz;
            if (dyo) dyo[addr] = (T)g;
            s0 += g;
            if (p.fb_dgamma) s1 += g * ((float)bz[addr] - mu) * rs;
            s2 += dz;
        }
        s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        __syncthreads();                                        // the K loop's stages are dead
        float* red = (float*)lds;                               // [wm][64 columns][3]
        if (lh == 0) {
            red[(wm * 64 + wn * 32 + li) * 3 + 0] = s0;
            red[(wm * 64 + wn * 32 + li) * 3 + 1] = s1;
            red[(wm * 64 + wn * 32 + li) * 3 + 2] = s2;
        }
        __syncthreads();
        if (tid < 64) {
            const int c = n0 + tid;
            const float gm = p.fb_gmul;
            if (p.fb_dbeta) atomicAdd(p.fb_dbeta + c, (red[tid * 3 + 0] + red[(64 + tid) * 3 + 0]) * gm);
            if (p.fb_dgamma) atomicAdd(p.fb_dgamma + c, (red[tid * 3 + 1] + red[(64 + tid) * 3 + 1]) * gm);
            if (p.fb_dbias) atomicAdd(p.fb_dbias + c, (red[tid * 3 + 2] + red[(64 + tid) * 3 + 2]) * gm);
        }
        return;
    }
    const float cbias = p.bias ? p.bias[n] : 0.f;
    const float csc = p.scale ? p.scale[n] : 1.f, csh = p.scale ? p.shift[n] : 0.f;
    const int mb = m0 + wm * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m >= p.M) continue;
        long long addr;
        if (p.dense) {
            addr = (long long)m * p.Cout + n;
        } else {
            const int ni = m / ohw, rem = m - ni * ohw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            addr = (long long)ni * p.ons + (long long)oh * p.ohs + (long long)ow * p.ows + n;
        }
        const float zv = acc[r] + cbias;
        if (zo) zo[addr] = (T)zv;
        float y = csc * zv + csh;
        if (res) y += (float)res[addr];
        if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
        else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
        out[addr] = (T)y;
    }
}

// mrcnn_mask (1x1 conv to C <= 16 class maps) + sigmoid on the 16-bit deconvolution output: one wave per pixel group,
// a lane owns 4 of the Cd channels (8-byte load), dot products reduced across the wave, float32 result [npix, C].
template <typename T>
__global__ __launch_bounds__(256) void mask_out_fwd_h16_kernel(const T* __restrict__ up, const float* __restrict__ wm,
                                                               const float* __restrict__ bm, float* mask, long long npix, int Cd, int C) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[4][16];
    const int nseg = Cd / 256;                                   // 256 channels per wave pass
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e][c] = (c < C && nseg == 1) ? wm[(long long)(lane * 4 + e) * C + c] : 0.f;
    const long long p0 = ((long long)blockIdx.x * 4 + wave) * 16;
    for (int q = 0; q < 16; ++q) {
        const long long pix = p0 + q;
        if (pix >= npix) break;
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.f;
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const t4 u = *(const t4*)(up + pix * Cd + sgm * 256 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float uf = (float)u[e];
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < C) acc[c] += uf * (nseg == 1 ? w[e][c] : wm[(long long)(sgm * 256 + lane * 4 + e) * C + c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < C) acc[c] = wave_sum(acc[c]);
        if (lane < C) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) if (c == lane) v = acc[c];
            v += bm ? bm[lane] : 0.f;
            mask[pix * C + lane] = 1.f / (1.f + expf(-v));
        }
    }
}

// C <= 4 (the repo's 3 classes + background), Cd = 256: a 16-lane group owns one pixel (16 channels = 32 bytes per lane),
// four pixels per wave at a time; the 4 dot products are reduced inside the group (4 xor steps).
template <typename T>
__global__ __launch_bounds__(256) void mask_out_fwd_h16_c4_kernel(const T* __restrict__ up, const float* __restrict__ wm,
                                                                  const float* __restrict__ bm, float* mask, long long npix, int C) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    float w[16][4];
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[e][c] = c < C ? wm[(long long)(sub * 16 + e) * C + c] : 0.f;
    const float bias = (sub < C && bm) ? bm[sub] : 0.f;
    const long long wave0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;       // 64 pixels per wave
    for (int it = 0; it < 16; ++it) {
        const long long pix = wave0 + it * 4 + grp;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (pix < npix) {
            const t8 u0 = *(const t8*)(up + pix * 256 + sub * 16);
            const t8 u1 = *(const t8*)(up + pix * 256 + sub * 16 + 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (float)u0[e], b = (float)u1[e];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += a * w[e][c] + b * w[8 + e][c];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
        if (pix < npix && sub < C) {
            const float v = (sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3]) + bias;
            mask[pix * C + sub] = 1.f / (1.f + expf(-v));
        }
    }
}

// mask_out_bwd_kernel on a 16-bit `up`: dzg is written in 16 bits, multiplied by the loss scale; sums stay float32.
template <typename T, int CP>
__global__ void mask_out_bwd_h16_kernel(const float* __restrict__ dmask, const float* __restrict__ mask, const T* __restrict__ up,
                                        const float* __restrict__ wm, T* dzg, float* dWm, float* dbm, float* dbd, long long npix,
                                        int H, int W, int Cd, int C, float lscale) {
    constexpr int PPB = 512;                    // 128 made the per-workgroup atomics (Cd*C + Cd + C of them) the bottleneck
    __shared__ __attribute__((aligned(16))) float sdz[PPB * CP];
    const int ci = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * PPB;
    const int np = (int)((npix - p0) < PPB ? (npix - p0) : PPB);
    for (int i = ci; i < PPB * CP; i += blockDim.x) {
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
    for (int c = 0; c < CP; ++c) { wrow[c] = c < C ? wm[(long long)ci * C + c] : 0.f; aw[c] = 0.f; }
    float abd = 0.f;
    __syncthreads();
    const int hw = H * W, W2 = W >> 1, H2 = H >> 1;
    const T* upp = up + p0 * Cd + ci;
#pragma unroll 4
    for (int pl = 0; pl < np; ++pl) {
        const float u = (float)upp[(long long)pl * Cd];
        float d = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < CP; c4 += 4) {
            const f32x4 z = *(const f32x4*)&sdz[pl * CP + c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { d += z[e] * wrow[c4 + e]; aw[c4 + e] += u * z[e]; }
        }
        const float dzu = u > 0.f ? d : 0.f;
        abd += dzu;
        const long long pix = p0 + pl;
        const long long n = pix / hw;
        const int rem = (int)(pix - n * hw);
        const int y = rem / W, x = rem - y * W;
        dzg[(((n * H2 + (y >> 1)) * W2 + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1))) * Cd + ci] = (T)(dzu * lscale);
    }
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (c < C) atomicAdd(&dWm[(long long)ci * C + c], aw[c]);
    atomicAdd(&dbd[ci], abd);
    if (ci < C) {
        float s_ = 0.f;
        for (int pl = 0; pl < np; ++pl) s_ += sdz[pl * CP + ci];
        atomicAdd(&dbm[ci], s_);
    }
}

// The same pass for Cd == 256 with 16-byte accesses: a pixel's 256 channels are 32 lanes x 8 channels, a workgroup walks
// 8 pixels at a time (slot = thread >> 5), 4 loads in flight per thread.  The one-thread-per-channel kernel above moves 2
// bytes per lane and instruction and keeps 4 x 128 bytes per wave in flight: 1.0 ms for 1.64 GB at 2048 ROIs (1.6 TB/s,
// latency bound); this one is bound by HBM.  Per-channel sums: registers per (slot, channel), LDS atomics across the 8
// slots, then the same global atomics per workgroup as before.
template <typename T, int CP>
__global__ __launch_bounds__(256) void mask_out_bwd_h16_v8_kernel(const float* __restrict__ dmask, const float* __restrict__ mask,
                                                                  const T* __restrict__ up, const float* __restrict__ wm, T* dzg,
                                                                  float* dWm, float* dbm, float* dbd, long long npix, int H, int W,
                                                                  int C, float lscale) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    constexpr int PPB = 512, Cd = 256;
    __shared__ __attribute__((aligned(16))) float sdz[PPB * CP];
    __shared__ float ssum[Cd * (CP + 1)];
    const int tid = threadIdx.x, cg = tid & 31, slot = tid >> 5;
    const long long p0 = (long long)blockIdx.x * PPB;
    const int np = (int)((npix - p0) < PPB ? (npix - p0) : PPB);
    for (int i = tid; i < PPB * CP; i += 256) {
        const int pl = i / CP, c = i - pl * CP;
        float v = 0.f;
        if (pl < np && c < C) {
            const float g = dmask[(p0 + pl) * C + c], q = mask[(p0 + pl) * C + c];
            v = g * q * (1.f - q);
        }
        sdz[i] = v;
    }
    for (int i = tid; i < Cd * (CP + 1); i += 256) ssum[i] = 0.f;
    float wrow[8][CP], aw[8][CP], abd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        abd[e] = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) { wrow[e][c] = c < C ? wm[(long long)(cg * 8 + e) * C + c] : 0.f; aw[e][c] = 0.f; }
    }
    __syncthreads();
    const int hw = H * W, W2 = W >> 1, H2 = H >> 1;
    const T* upp = up + p0 * Cd + cg * 8;
#pragma unroll 4
    for (int pl = slot; pl < np; pl += 8) {
        const t8 u = *(const t8*)(upp + (long long)pl * Cd);
        float z[CP];
#pragma unroll
        for (int c4 = 0; c4 < CP; c4 += 4) {
            const f32x4 zz = *(const f32x4*)&sdz[pl * CP + c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) z[c4 + e] = zz[e];
        }
        t8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float uf = (float)u[e];
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c) { d += z[c] * wrow[e][c]; aw[e][c] += uf * z[c]; }
            const float dzu = uf > 0.f ? d : 0.f;
            abd[e] += dzu;
            o[e] = (T)(dzu * lscale);
        }
        const long long pix = p0 + pl;
        const long long n = pix / hw;
        const int rem = (int)(pix - n * hw);
        const int y = rem / W, x = rem - y * W;
        *(t8*)(dzg + (((n * H2 + (y >> 1)) * W2 + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1))) * Cd + cg * 8) = o;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int c = 0; c < CP; ++c) atomicAdd(&ssum[(cg * 8 + e) * (CP + 1) + c], aw[e][c]);
        atomicAdd(&ssum[(cg * 8 + e) * (CP + 1) + CP], abd[e]);
    }
    __syncthreads();
    {
        const int ci = tid;                                     // 256 threads = 256 channels
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) atomicAdd(&dWm[(long long)ci * C + c], ssum[ci * (CP + 1) + c]);
        atomicAdd(&dbd[ci], ssum[ci * (CP + 1) + CP]);
        if (ci < C) {
            float s_ = 0.f;
            for (int pl = 0; pl < np; ++pl) s_ += sdz[pl * CP + ci];
            atomicAdd(&dbm[ci], s_);
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
__global__ void cast_to_h16_kernel(const float* __restrict__ src, T* dst, long long n, float mul) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *(const f32x4*)(src + i);
        t4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (T)(v[k] * mul);
        if ((reinterpret_cast<uintptr_t>(dst + i) & 7) == 0) *(t4*)(dst + i) = o;
        else { dst[i] = o[0]; dst[i + 1] = o[1]; dst[i + 2] = o[2]; dst[i + 3] = o[3]; }
    } else {
        for (long long j = i; j < n; ++j) dst[j] = (T)(src[j] * mul);
    }
}

// Backward of  out = act(scale * (conv + bias) + shift)  on 16-bit tensors (mrcnn_epilogue_bwd in 16 bits): reads the
// upstream gradient, the activated output and the pre-BN value z, writes dz (16 bit) and accumulates the channel sums
// dgamma / dbeta / dbias in float32, multiplied by `gmul` (1 / loss scale) -- dz itself stays scaled.
// A workgroup owns 2^lg groups of 8 channels (16-byte loads; blockIdx.y) and a range of rows (blockIdx.x); its 256 >> lg
// row lanes walk the rows four at a time with all loads of a batch issued before the first store (dz_out may alias
// dout).  Channel sums: registers -> LDS -> one global atomic per channel per workgroup (a narrow channel slice keeps
// those few).
#define EPI16_U 4
template <typename T>
__global__ __launch_bounds__(256) void epilogue_bwd_h16_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                               const T* __restrict__ z, const float* __restrict__ scale,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               T* dz_out, float* dgamma, float* dbeta, float* dbias, long long M,
                                                               int C, int act, long long rows_per_block, int lg, float gmul, T* dy_out) {
    __shared__ float sacc[3 * 8 * 256];   // [3][8L]
    const int L = 1 << lg, R = 256 >> lg;
    for (int c = threadIdx.x; c < 24 * L; c += 256) sacc[c] = 0.f;
    __syncthreads();
    const int rsub = threadIdx.x >> lg, lane = threadIdx.x & (L - 1);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    typedef T t8 __attribute__((ext_vector_type(8)));
    const int c = (blockIdx.y * L + lane) * 8;
    float sc[8], mu[8], rs[8], a_db[8], a_dg[8], a_bias[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = scale ? scale[c + k] : 1.f;
        mu[k] = dgamma ? mean[c + k] : 0.f;
        rs[k] = dgamma ? rstd[c + k] : 0.f;
        a_db[k] = a_dg[k] = a_bias[k] = 0.f;
    }
    for (long long rb = r0 + rsub; rb < r1; rb += (long long)R * EPI16_U) {
        t8 gg[EPI16_U], oo[EPI16_U], zz[EPI16_U];
#pragma unroll
        for (int u = 0; u < EPI16_U; ++u) {
            long long r = rb + (long long)u * R;
            if (r >= r1) r = r1 - 1;                           // clamped: in range, result discarded below
            const long long e = r * C + c;
            gg[u] = *(const t8*)(dout + e);
            if (act == MRCNN_ACT_RELU) oo[u] = *(const t8*)(out + e);
            if (dgamma) zz[u] = *(const t8*)(z + e);
        }
#pragma unroll
        for (int u = 0; u < EPI16_U; ++u) {
            const long long r = rb + (long long)u * R;
            if (r >= r1) break;
            const long long e = r * C + c;
            t8 dzv, dyv;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float g = (float)gg[u][k];
                if (act == MRCNN_ACT_RELU) g = (float)oo[u][k] > 0.f ? g : 0.f;
                const float dz = g * sc[k];
                dzv[k] = (T)dz;
                dyv[k] = (T)g;
                if (dgamma) a_dg[k] += g * ((float)zz[u][k] - mu[k]) * rs[k];
                a_db[k] += g;
                a_bias[k] += dz;
            }
            *(t8*)(dz_out + e) = dzv;
            if (dy_out) *(t8*)(dy_out + e) = dyv;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (dbeta || dgamma) atomicAdd(&sacc[lane * 8 + k], a_db[k]);
        if (dgamma) atomicAdd(&sacc[8 * L + lane * 8 + k], a_dg[k]);
        if (dbias) atomicAdd(&sacc[16 * L + lane * 8 + k], a_bias[k]);
    }
    __syncthreads();
    const int cb = blockIdx.y * 8 * L;
    for (int j = threadIdx.x; j < 8 * L; j += 256) {
        if (dbeta) atomicAdd(&dbeta[cb + j], sacc[j] * gmul);
        if (dgamma) atomicAdd(&dgamma[cb + j], sacc[8 * L + j] * gmul);
        if (dbias) atomicAdd(&dbias[cb + j], sacc[16 * L + j] * gmul);
    }
}

// 8 elements per thread when n % 8 == 0 and both pointers are 16-byte aligned (every tensor of the engine); ACC: dst += src * mul
template <typename T, bool ACC>
__global__ void cast_from_h16_kernel(const T* __restrict__ src, float* dst, long long n, float mul, int vec) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const long long i = t * 8;
        if (i >= n) return;
        const t8 v = *(const t8*)(src + i);
        f32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] = (float)v[k] * mul; hi[k] = (float)v[4 + k] * mul; }
        if (ACC) {
            const f32x4 a = *(const f32x4*)(dst + i), b = *(const f32x4*)(dst + i + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo[k] += a[k]; hi[k] += b[k]; }
        }
        *(f32x4*)(dst + i) = lo;
        *(f32x4*)(dst + i + 4) = hi;
        return;
    }
    for (long long j = t * 8; j < t * 8 + 8 && j < n; ++j) dst[j] = (ACC ? dst[j] : 0.f) + (float)src[j] * mul;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient in 16 bits: dW[(tap, ci), co] = sum_m X[m, (tap, ci)] * dY[m, co], float32 accumulation and
// float32 output (slabs per pixel split, summed in fixed order into the float32 gradient buffer).
// The contraction runs over pixels, but both operands are stored pixel-major, so each MFMA operand (8
// consecutive pixels of one channel per lane) is a TRANSPOSED read of the LDS tile: ds_read_b64_tr_b16 delivers
// a 4-pixel x 16-channel block column-major to a 16-lane group.  Tiles: 256 (tap, ci) x 128 co per workgroup,
// 4 waves of 128 x 64, 32 pixels per step; X [32][256] (512-byte rows) and dY [32][128] (256-byte rows) arrive
// by buffer-addressed LDS-DMA.  The four pixel rows one transposed read touches are 512 / 256 bytes apart and
// would share banks: logical 16-byte chunk c of pixel row r is stored at chunk c ^ ((r & 3) << 2) (applied on
// the SOURCE side of the DMA, and in the read addresses), which makes the reads conflict-free.
// Per-pixel input offsets and tap-validity masks come from a table built once per geometry
// (pixel_table_kernel): 16 bytes per pixel, prefetched one step ahead -- no divisions in the loop.
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4* h16_tr_ptr;
#define H16_WG_OOB PIXEL_TABLE_OOB

struct WgradH16Args {
    const void* x; const void* dy; float* out; const PixelEntry* table;
    int N, H, W, Cin, Cout, KH, KW, OH, OW, M, Ktot, splits, chunk, xcd_order;
    unsigned x_shift, x_records;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_h16_kernel(const WgradH16Args p) {
    typedef typename H16Traits<T>::v8 v8;
    constexpr int BI = 256, BN = 128, BP = 32, TM = 4, TN = 2;
    constexpr int XROW = BI * 2, YROW = BN * 2;                 // bytes per pixel row
    constexpr int XB = BP * XROW, YB = BP * YROW;               // 16 KiB + 8 KiB per buffer
    // two separately declared buffers: the compiler then knows that the LDS-DMA into one cannot alias the transposed
    // reads of the other and leaves the DMA in flight during the MFMAs (with one array it waited vmcnt(0) before the reads)
    __shared__ __attribute__((aligned(16))) char lds0[XB + YB];
    __shared__ __attribute__((aligned(16))) char lds1[XB + YB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.Cout / BN, itiles = p.Ktot / BI;
    // all (tap, channel-tile) workgroups of one pixel split read the same dY rows and -- shifted by a tap -- the same X
    // rows: keep them on one XCD (PMC: FETCH_SIZE 2.57 GB per launch on the mask-head shape with the plain order, 6.2 x
    // the algorithmic bytes, profiles/r02_pmc_h16_kernels.txt)
    int bid = p.xcd_order ? (int)mrcnn_xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int ntile = bid % ntiles; bid /= ntiles;
    const int itile = bid % itiles;
    const int split = bid / itiles;
    const int i0 = itile * BI, n0 = ntile * BN;
    const int m_begin = split * p.chunk;
    const int m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
    const int tap = i0 / p.Cin, ci0 = i0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - p.x_shift), 0, p.x_records, 0x00020000);
    const unsigned soff_x = (unsigned)(((kh * p.W + kw) * p.Cin + ci0) * 2);
    const unsigned tapbit = 1u << (tap & 31);
    const bool tap_hi = tap >= 32;

    // X piece j = rows 2j, 2j+1 (wave: j = wave + 4 jj): lane -> row 2j + (lane>>5), physical chunk lane&31
    const int xr = (2 * wave + (lane >> 5)) & 3;                // (row & 3), the same for all four pieces of this wave
    const unsigned x_lane = (unsigned)((((lane & 31) ^ (xr << 2)) & 31) * 16);
    // dY piece j = rows 4j .. 4j+3 (wave: j = wave + 4 jj): lane -> row 4j + (lane>>4), physical chunk lane&15
    const int yr = (lane >> 4) & 3;
    const unsigned y_voff = (unsigned)((lane >> 4) * p.Cout * 2 + (((lane & 15) ^ (yr << 2)) & 15) * 16);

    const PixelEntry* tab = p.table + (lane >> 5);
    PixelEntry ent[4];
    auto fetch = [&](int mb) {                                   // table rows exist up to M + BP - 1
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ent[jj] = tab[mb + 2 * (wave + 4 * jj)];
    };
    auto stage = [&](char* xb, int mb) {
        char* yb = xb + XB;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const unsigned word = tap_hi ? ent[jj].mask_hi : ent[jj].mask_lo;
            const unsigned vo = ((word & tapbit) ? ent[jj].off : H16_WG_OOB) + x_lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (h16_lds_ptr)(xb + (wave + 4 * jj) * 1024), 16, vo, soff_x, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = wave + 4 * jj;
            const int row0 = mb + 4 * j;
            const int left = m_end - row0;
            const unsigned rec = left > 0 ? (unsigned)(((left - 1) * p.Cout + BN) * 2) : 0u;
            const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)p.dy + ((long long)row0 * p.Cout + n0) * 2), 0, rec, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (h16_lds_ptr)(yb + j * 1024), 16, y_voff, 0, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposed-read addresses: lane supplies pixel row 8*(lane>>5) + q (+ 4t + 16kk), q = (lane&15)>>2, and the 8 bytes of
    // columns 4*(lane&3) .. +3 of its 16-lane group's 16 channels; logical chunk -> physical chunk ^ (q << 2)
    const int q = (lane & 15) >> 2;
    const int xcol = wm * 128 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);          // channel (+ a * 32)
    const int ycol = wn * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int prow = 8 * (lane >> 5) + q;
    // a * 32 channels = 4 chunks: the XOR touches chunk bits 2..3 only when q != 0, so "+ a * 64 bytes" is not uniform;
    // keep one base per a (4) and per b (2) instead of immediates
    int x_rda[TM];                                            // byte offsets inside a buffer
    int y_rdb[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int c = xcol + a * 32;
        x_rda[a] = prow * XROW + ((((c >> 3) ^ (q << 2)) & 31) << 4) + (c & 7) * 2;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int c = ycol + b * 32;
        y_rdb[b] = XB + prow * YROW + ((((c >> 3) ^ (q << 2)) & 15) << 4) + (c & 7) * 2;
    }

    auto compute = [&](auto curc) {
        const char* const buf = decltype(curc)::value ? lds1 : lds0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                union { s16x4 h[2]; v8 v; } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_tr_ptr)(buf + x_rda[a] + kk * 16 * XROW));
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_tr_ptr)(buf + x_rda[a] + kk * 16 * XROW + 4 * XROW));
                av[a] = u.v;
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                union { s16x4 h[2]; v8 v; } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_tr_ptr)(buf + y_rdb[b] + kk * 16 * YROW));
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_tr_ptr)(buf + y_rdb[b] + kk * 16 * YROW + 4 * YROW));
                bv[b] = u.v;
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = H16Traits<T>::mfma(av[a], bv[b], acc[a][b]);
        }
    };

    if (m_begin < m_end) {
        // unconditional, clamped table prefetch (see conv_wgrad_blds_body: a conditional fetch costs a vmcnt(0) wait
        // right behind every DMA issue)
        auto fetch_at = [&](int mb) { fetch(mb < p.M ? mb : p.M); };
        fetch(m_begin);
        stage(lds0, m_begin);
        fetch_at(m_begin + BP);
        __syncthreads();
        for (int mb = m_begin; mb < m_end; mb += 2 * BP) {
            const bool has1 = mb + BP < m_end;
            if (has1) stage(lds1, mb + BP);
            fetch_at(mb + 2 * BP);
            compute(std::integral_constant<int, 0>{});
            __syncthreads();
            if (has1 && mb + 2 * BP < m_end) stage(lds0, mb + 2 * BP);
            fetch_at(mb + 3 * BP);
            if (has1) {
                compute(std::integral_constant<int, 1>{});
                __syncthreads();
            }
        }
    }

    float* dst = p.out + (long long)split * p.Ktot * p.Cout;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int ib = i0 + wm * 128 + a * 32 + 4 * lh, n = n0 + wn * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(long long)(ib + (r & 3) + 8 * (r >> 2)) * p.Cout + n] = acc[a][b][r];
        }
}

// Phased weight gradient for the big stride-1 "same" layers (mask head: M = 401 408 pixels, 9 taps x 256 x 256): the
// schedule of conv_fwd_h16p_kernel turned to the pixel contraction.  A workgroup owns one 256 (tap, ci) x 256 co tile of dW
// over one pixel split (tiles x splits ~ one workgroup per CU, one round); 8 waves = two staggered groups of four
// (wr = wave >> 2: 128 input channels, wc = wave & 3: 64 output channels), each wave 8 x 4 tiles of v_mfma_f32_16x16x32
// (A = dY^T, B = X: a lane ends with 4 consecutive co of one (tap, ci) row), K-step = 64 pixels, two 64 KiB stages.
//   phases     as in the forward kernel: four per K-step, [operand reads | one quarter of a later stage by LDS-DMA |
//              s_waitcnt vmcnt(8)] s_barrier [16 MFMAs] s_barrier, group 1 one barrier behind group 0; quadrants
//              (X0,Y0) (X0,Y1) (X1,Y1) (X1,Y0) with X0 / X1 = the wave's input channels 0..63 / 64..127 and Y0 / Y1 = its
//              output channels 0..31 / 32..63; quarter issue order (t+1).B1 (t+1).A1 (t+2).A0 (t+2).B0 (t+2).B1 (t+2).A1
//              (t+3).A0 (t+3).B0 -- every quarter is re-staged >= 2 phases after its last read, lands >= 4 phases later.
//   staging    both operands are pixel-major in memory, so a quarter is [64 pixels][128 channels] (256-byte rows): A0 / A1 =
//              the X0 / X1 channels of both groups, B0 / B1 = the Y0 / Y1 channels of the four wave columns; a DMA piece =
//              4 pixel rows, 2 pieces per wave and quarter.  The contraction index is the ROW index of both tiles: every MFMA
//              operand is two ds_read_b64_tr_b16 (4 pixels x 16 channels each).  One such read touches pixel rows
//              8 fq + q (fq = lane >> 4, q = (lane & 15) >> 2), 32 bytes in each; rows are 256 bytes = all 64 banks apart, so
//              logical 16-byte chunk c of pixel row r is stored at chunk c ^ (((r & 3) | ((r >> 3) & 1) << 2) << 1) -- the 16
//              rows of a read then cover every 32-byte bank group exactly twice (the minimum for 512 bytes), applied on the
//              DMA's source side and in the read addresses.
//   addresses  stride 1 and OH == H, OW == W make the input row of (pixel m, tap) linear: m + kh * W + kw rows behind the
//              shifted base; only the tap's validity needs (oh, ow), two multiply-high divisions per piece.  No table, no
//              ordinary global load in the loop (the counted vmcnt waits see DMA pieces only).  Rows past the split's end
//              fall outside the dY descriptor (zeros); X rows there are switched off too (0 x NaN).
static void h16_magic(unsigned d, unsigned* mg, unsigned* sh);

struct WgradH16PArgs {
    const void* x; const void* dy; float* out;
    int H, W, Cin, Cout, KW, pad_t, pad_l, M, Ktot, chunk, nk, itiles, ntiles, ohw, xcd_order;
    unsigned x_shift, x_records, mg_ohw, sh_ohw, mg_ow, sh_ow;
};

template <typename T>
__global__ __launch_bounds__(512) void conv_wgrad_h16p_kernel(const WgradH16PArgs p) {
    typedef typename H16Traits<T>::v8 v8;
    constexpr int QBYTES = 16384, ROW = 256;
    // EIGHT LDS objects, one per (stage, quarter): hipcc orders an LDS read after every LDS-DMA it cannot tell apart from the
    // read's object -- with one array it put s_waitcnt vmcnt(0) in front of the transposed reads of every phase.  A phase
    // never reads the quarter it stages, so with separate objects the only waits the compiler adds are for the DMA that
    // filled the quarter being read (>= 4 phases old: weaker than the kernel's own vmcnt(8)).
    __shared__ __attribute__((aligned(16))) char q_a0_s0[QBYTES];
    __shared__ __attribute__((aligned(16))) char q_b0_s0[QBYTES];
    __shared__ __attribute__((aligned(16))) char q_b1_s0[QBYTES];
    __shared__ __attribute__((aligned(16))) char q_a1_s0[QBYTES];
    __shared__ __attribute__((aligned(16))) char q_a0_s1[QBYTES];
    __shared__ __attribute__((aligned(16))) char q_b0_s1[QBYTES];
    __shared__ __attribute__((aligned(16))) char q_b1_s1[QBYTES];
    __shared__ __attribute__((aligned(16))) char q_a1_s1[QBYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // the (tap, channel-tile) workgroups of one pixel split read the same dY rows and -- shifted by a tap -- the same X rows:
    // logical ids that are neighbours share an XCD and with it an L2 (PMC before: FETCH_SIZE 3.2 GB per launch on the
    // mask-head layer, 7.7 x the algorithmic bytes at ~7 TB/s -- the kernel was fabric-bound)
    int bid = p.xcd_order ? (int)mrcnn_xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int ntile = bid % p.ntiles; bid /= p.ntiles;
    const int itile = bid % p.itiles;
    const int split = bid / p.itiles;
    const int i0 = itile * 256, n0 = ntile * 256;
    const int tap = i0 / p.Cin, ci0 = i0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dh = kh - p.pad_t, dw_ = kw - p.pad_l;
    const int m_begin = split * p.chunk;
    const int m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
    const int nk = p.nk, nk2 = (nk + 1) & ~1;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - p.x_shift), 0, p.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)((long long)m_end * p.Cout * 2), 0x00020000);
    const unsigned soff_x = (unsigned)(((kh * p.W + kw) * p.Cin + ci0) * 2);
    const unsigned soff_y = (unsigned)(n0 * 2);

    // ---- DMA pieces: piece pc = wave + 8 j of a quarter = pixel rows 4 pc .. 4 pc + 3; lane -> (row r4, physical chunk) ----
    const int r4 = lane >> 4, pchunk = lane & 15;
    unsigned a_chan[2][2], b_chan[2][2];            // [quarter 0/1][piece j]: byte offset of this lane's 8 channels inside the pixel row
    int row_in_step[2];                             // pixel row of the piece inside the K-step
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 4 * (wave + 8 * j) + r4;
        row_in_step[j] = r;
        const int lc = pchunk ^ ((((r & 3) | (((r >> 3) & 1) << 2)) << 1));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a_chan[q][j] = (unsigned)(((lc >> 3) * 128 + q * 64 + (lc & 7) * 8) * 2);
            b_chan[q][j] = (unsigned)(((lc >> 2) * 64 + q * 32 + (lc & 3) * 8) * 2);
        }
    }
    const int piece_lds0 = wave * 1024, piece_lds1 = (wave + 8) * 1024;

    int it_kt = 0;                                  // the K-step whose quarters are being issued
    auto issue_a = [&](auto qc, auto bufc) {
        constexpr int q = decltype(qc)::value;
        char* base = decltype(bufc)::value ? (q ? q_a1_s1 : q_a0_s1) : (q ? q_a1_s0 : q_a0_s0);
        const int mb = m_begin + it_kt * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = mb + row_in_step[j];
            const unsigned n = p.mg_ohw ? (__umulhi((unsigned)m, p.mg_ohw) >> p.sh_ohw) : (unsigned)m;
            const unsigned pos = (unsigned)m - n * (unsigned)p.ohw;
            const unsigned oh = p.mg_ow ? (__umulhi(pos, p.mg_ow) >> p.sh_ow) : pos;
            const unsigned ow = pos - oh * (unsigned)p.W;
            const bool ok = (unsigned)((int)oh + dh) < (unsigned)p.H && (unsigned)((int)ow + dw_) < (unsigned)p.W && m < m_end &&
                            it_kt < nk;
            const unsigned vo = ok ? (unsigned)m * (unsigned)(p.Cin * 2) + a_chan[q][j] : H16_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (h16_lds_ptr)(base + (j ? piece_lds1 : piece_lds0)), 16, vo, soff_x, 0, 0);
        }
    };
    auto issue_b = [&](auto qc, auto bufc) {
        constexpr int q = decltype(qc)::value;
        char* base = decltype(bufc)::value ? (q ? q_b1_s1 : q_b0_s1) : (q ? q_b1_s0 : q_b0_s0);
        const int mb = m_begin + it_kt * 64;
        const unsigned dead = it_kt < nk ? 0u : H16_OOB_OFFSET;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = ((unsigned)(mb + row_in_step[j]) * (unsigned)(p.Cout * 2) + b_chan[q][j]) | dead;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (h16_lds_ptr)(base + (j ? piece_lds1 : piece_lds0)), 16, vo, soff_y, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- transposed operand reads: lane (g = lane & 15, fq = lane >> 4) addresses pixel row 8 fq + (g >> 2) (+ 4, + 32 ks) and
    // the 8 bytes of channels 4 (g & 3) .. + 3 of the 16-channel tile; it receives channel g of those four pixels ----
    const int g = lane & 15, fq = lane >> 4, q4 = g >> 2, sb = g & 3;
    const int fsw = (q4 | ((fq & 1) << 2)) << 1;
    const int rd_row = (8 * fq + q4) * ROW + (sb & 1) * 8;
    int x_rd[4], y_rd[2];                           // per 16-channel tile of a quarter: (logical chunk ^ swizzle) * 16, + row part
#pragma unroll
    for (int a = 0; a < 4; ++a) x_rd[a] = rd_row + ((((8 * wr + 2 * a) ^ fsw) + (sb >> 1)) << 4);
#pragma unroll
    for (int b = 0; b < 2; ++b) y_rd[b] = rd_row + ((((4 * wc + 2 * b) ^ fsw) + (sb >> 1)) << 4);
    auto rd8 = [&](const char* ptr) -> v8 {
        union { s16x4 h[2]; v8 v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_tr_ptr)ptr);
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_tr_ptr)(ptr + 4 * ROW));
        return u.v;
    };
    v8 xf[4][2], y0[2][2], y1[2][2];

    auto phase = [&](auto phc) {
        constexpr int PH = decltype(phc)::value;                // 0..7
        constexpr int BUF = PH >> 2, Q = PH & 3;
        if constexpr (Q == 0) {
            const char* yq = BUF ? q_b0_s1 : q_b0_s0;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) y0[b][ks] = rd8(yq + y_rd[b] + ks * 32 * ROW);
        }
        if constexpr (Q == 1) {
            const char* yq = BUF ? q_b1_s1 : q_b1_s0;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) y1[b][ks] = rd8(yq + y_rd[b] + ks * 32 * ROW);
        }
        if constexpr (Q == 0 || Q == 2) {
            if constexpr (Q == 0) __builtin_amdgcn_sched_barrier(0);
            const char* xq = BUF ? (Q == 2 ? q_a1_s1 : q_a0_s1) : (Q == 2 ? q_a1_s0 : q_a0_s0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) xf[a][ks] = rd8(xq + x_rd[a] + ks * 32 * ROW);
        }
        // one quarter of a later stage
        if constexpr (PH == 0) issue_b(I1{}, I1{});
        if constexpr (PH == 1) { issue_a(I1{}, I1{}); ++it_kt; }
        if constexpr (PH == 2) issue_a(I0{}, I0{});
        if constexpr (PH == 3) issue_b(I0{}, I0{});
        if constexpr (PH == 4) issue_b(I1{}, I0{});
        if constexpr (PH == 5) { issue_a(I1{}, I0{}); ++it_kt; }
        if constexpr (PH == 6) issue_a(I0{}, I1{});
        if constexpr (PH == 7) issue_b(I0{}, I1{});
        h16p_wait<8>();                                         // everything but the 4 youngest quarters has landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        constexpr int AB = (Q >= 2) ? 4 : 0;                     // input-channel tiles of this quadrant
        constexpr int CB = (Q == 1 || Q == 2) ? 2 : 0;           // output-channel tiles of this quadrant
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[AB + a][CB + b] = H16Traits<T>::mfma16((CB ? y1 : y0)[b][ks], xf[a][ks], acc[AB + a][CB + b]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // prologue: K-step 0 whole, K-step 1's A0 and B0; group 1 drops one barrier behind
    issue_a(I0{}, I0{}); issue_b(I0{}, I0{}); issue_b(I1{}, I0{}); issue_a(I1{}, I0{}); ++it_kt;
    issue_a(I0{}, I1{}); issue_b(I0{}, I1{});
    if (wr == 1) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();

    for (int t = 0; t < nk2; t += 2)
        h16_static_for([&](auto phc) { phase(phc); }, std::make_integer_sequence<int, 8>{});
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the out-of-range tail DMAs still target this workgroup's LDS

    // slab [split][Ktot][Cout]: lane (g, fq) holds rows (tap, ci) = .. + g, columns co = .. + 4 fq .. + 3
    float* dst = p.out + (long long)split * p.Ktot * p.Cout;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const long long row = (long long)(i0 + wr * 128 + a * 16 + g) * p.Cout + n0 + wc * 64 + 4 * fq;
#pragma unroll
        for (int b = 0; b < 4; ++b) *(f32x4*)(dst + row + b * 16) = acc[a][b];
    }
}

__global__ void wgrad_h16_reduce_kernel(const float* __restrict__ slabs, float* dw, long long n, int splits, int acc, float mul) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = mrcnn_slab_sum<float>(0.f, slabs, n, i, splits) * mul;
    dw[i] = acc ? dw[i] + s : s;
}

// n % 4 == 0, 16-byte aligned buffers: four elements per thread
__global__ void wgrad_h16_reduce_vec_kernel(const float* __restrict__ slabs, float* dw, long long n4, int splits, int acc, float mul) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s = mrcnn_slab_sum<f32x4>(zero4, slabs, 4 * n4, 4 * i, splits);
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] *= mul;
    if (acc) {
        const f32x4 o = *(const f32x4*)(dw + 4 * i);
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = o[q] + s[q];
    }
    *(f32x4*)(dw + 4 * i) = s;
}

struct WgradH16Plan { int splits, chunk; size_t table_bytes, slab_bytes; };

static WgradH16Plan plan_wgrad_h16(const mrcnn_conv_desc* d) {
    WgradH16Plan pl;
    const long long M = (long long)d->N * d->OH * d->OW;
    const long long tiles = (long long)(d->KH * d->KW * d->Cin / 256) * (d->Cout / 128);
    long long splits = 768 / (tiles > 0 ? tiles : 1);           // 3 workgroups per CU (48 KiB LDS each)
    const long long max_splits = (M + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    long long chunk = ((M + splits - 1) / splits + 31) / 32 * 32;
    splits = (M + chunk - 1) / chunk;
    pl.splits = (int)splits; pl.chunk = (int)chunk;
    pl.table_bytes = (size_t)(M + 32) * sizeof(PixelEntry);
    pl.slab_bytes = (size_t)splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
    return pl;
}

// Phased kernel (conv_wgrad_h16p_kernel): stride 1, output as large as the input, 256-channel tiles on both sides, one
// workgroup per CU in one round, enough K-steps per split for the pipeline to matter.
struct WgradH16PPlan { int ok, splits, chunk, nk, itiles, ntiles; size_t slab_bytes; };

static WgradH16PPlan plan_wgrad_h16p(const mrcnn_conv_desc* d) {
    WgradH16PPlan pl;
    memset(&pl, 0, sizeof(pl));
    // MRCNN_WGRAD_H16_PHASE (read per call: tests and A/B timings switch it): 0 never, 1 where it pays (default), 2 on every
    // shape the kernel can take (tests: short and ragged pixel ranges)
    const char* env = getenv("MRCNN_WGRAD_H16_PHASE");
    const int enabled = env ? atoi(env) : 1;
    const long long M = (long long)d->N * d->OH * d->OW;
    if (!enabled || d->stride != 1 || d->OH != d->H || d->OW != d->W || d->Cin % 256 || d->Cout % 256 || d->KH * d->KW > 64 ||
        d->pad_t < 0 || d->pad_l < 0 || d->pad_t >= d->KH || d->pad_l >= d->KW || M * d->Cout * 2 >= 0xFFFFFF00LL)
        return pl;
    const int cus = mrcnn_num_cus();
    pl.itiles = d->KH * d->KW * d->Cin / 256; pl.ntiles = d->Cout / 256;
    const long long tiles = (long long)pl.itiles * pl.ntiles;
    long long splits = cus / tiles;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    const long long chunk = ((M + splits - 1) / splits + 63) / 64 * 64;
    splits = (M + chunk - 1) / chunk;
    if (enabled < 2 && (chunk < 64 * 32 || tiles * splits * 2 < cus)) return pl;   // short K loops / half-empty chip: the table kernel's case
    pl.ok = 1; pl.splits = (int)splits; pl.chunk = (int)chunk; pl.nk = (int)(chunk / 64);
    pl.slab_bytes = (size_t)splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
    return pl;
}

static int wgrad_h16_shape_ok(const mrcnn_conv_desc* d) {
    return d && d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 &&
           d->OH > 0 && d->OW > 0 && d->KH * d->KW <= 64 && d->Cin % 256 == 0 && d->Cout % 128 == 0;
}

extern "C" size_t mrcnn_conv2d_wgrad_h16_workspace(const mrcnn_conv_desc* d) {
    if (!wgrad_h16_shape_ok(d)) return 0;
    const WgradH16Plan pl = plan_wgrad_h16(d);
    const WgradH16PPlan pp = plan_wgrad_h16p(d);
    const size_t slabs = pl.slab_bytes > pp.slab_bytes ? pl.slab_bytes : pp.slab_bytes;
    return pl.table_bytes + slabs + 512;
}

extern "C" int mrcnn_conv2d_wgrad_h16(const mrcnn_conv_desc* d, int dtype, const void* x, const void* dy, float* dw,
                                      void* workspace, size_t workspace_bytes, int beta_acc, float multiplier, void* stream) {
    if (!wgrad_h16_shape_ok(d) || !x || !dy || !dw || !workspace || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return MRCNN_ERR_ARG;
    const long long M = (long long)d->N * d->OH * d->OW;
    const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 2;
    const long long shift_b = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 2;
    if (M >= (1LL << 30) || xbytes + shift_b >= 0x7FFFFF00LL || M * d->Cout * 2 >= 0x7FFFFF00LL) return MRCNN_ERR_ARG;
    const WgradH16Plan pl = plan_wgrad_h16(d);
    if (workspace_bytes < mrcnn_conv2d_wgrad_h16_workspace(d)) return MRCNN_ERR_WORKSPACE;
    char* ws = (char*)((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    PixelEntry* table = (PixelEntry*)ws;
    float* slabs = (float*)(ws + ((pl.table_bytes + 255) & ~(size_t)255));
    hipStream_t s = (hipStream_t)stream;
    const WgradH16PPlan pp = plan_wgrad_h16p(d);
    if (pp.ok && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0) {
        WgradH16PArgs b;
        b.x = x; b.dy = dy; b.out = slabs;
        b.H = d->H; b.W = d->W; b.Cin = d->Cin; b.Cout = d->Cout; b.KW = d->KW; b.pad_t = d->pad_t; b.pad_l = d->pad_l;
        b.M = (int)M; b.Ktot = d->KH * d->KW * d->Cin; b.chunk = pp.chunk; b.nk = pp.nk; b.itiles = pp.itiles; b.ntiles = pp.ntiles;
        b.ohw = d->OH * d->OW;
        { const char* xe = getenv("MRCNN_WGRAD_XCD"); b.xcd_order = xe ? atoi(xe) : 1; }
        b.x_shift = (unsigned)shift_b; b.x_records = (unsigned)(xbytes + shift_b);
        h16_magic((unsigned)(d->OH * d->OW), &b.mg_ohw, &b.sh_ohw);
        h16_magic((unsigned)d->OW, &b.mg_ow, &b.sh_ow);
        const unsigned pblocks = (unsigned)(pp.itiles * pp.ntiles * pp.splits);
        if (dtype == MRCNN_DTYPE_F16) hipLaunchKernelGGL(conv_wgrad_h16p_kernel<_Float16>, dim3(pblocks), dim3(512), 0, s, b);
        else hipLaunchKernelGGL(conv_wgrad_h16p_kernel<__bf16>, dim3(pblocks), dim3(512), 0, s, b);
        const long long n = (long long)b.Ktot * b.Cout;                  // Cout % 256 == 0: the vector reduction always fits
        hipLaunchKernelGGL(wgrad_h16_reduce_vec_kernel, dim3((unsigned)cdiv64(n / 4, 256)), dim3(256), 0, s, slabs, dw, n / 4,
                           pp.splits, beta_acc, multiplier);
        return mrcnn_launch_status();
    }
    const int rows = (int)M + 32;
    hipLaunchKernelGGL(pixel_table_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, table, d->N, d->H, d->W, d->Cin,
                       d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->OH, d->OW, (int)M, rows, (unsigned)shift_b, 2);
    WgradH16Args a;
    a.x = x; a.dy = dy; a.out = slabs; a.table = table;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.OH = d->OH; a.OW = d->OW;
    a.M = (int)M; a.Ktot = d->KH * d->KW * d->Cin; a.splits = pl.splits; a.chunk = pl.chunk;
    a.x_shift = (unsigned)shift_b; a.x_records = (unsigned)(xbytes + shift_b);
    static const int xcd_env = getenv("MRCNN_WGRAD_XCD") ? atoi(getenv("MRCNN_WGRAD_XCD")) : 1;
    a.xcd_order = xcd_env;
    const unsigned blocks = (unsigned)((a.Ktot / 256) * (a.Cout / 128) * pl.splits);
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(conv_wgrad_h16_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(conv_wgrad_h16_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, a);
    const long long n = (long long)a.Ktot * a.Cout;
    if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0)
        hipLaunchKernelGGL(wgrad_h16_reduce_vec_kernel, dim3((unsigned)cdiv64(n / 4, 256)), dim3(256), 0, s, slabs, dw, n / 4,
                           pl.splits, beta_acc, multiplier);
    else
        hipLaunchKernelGGL(wgrad_h16_reduce_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, slabs, dw, n, pl.splits,
                           beta_acc, multiplier);
    return mrcnn_launch_status();
}


// x / d for x < 2^31 as (umulhi(x, mg) >> sh): mg = ceil(2^(31+s) / d), s = ceil(log2 d), sh = s - 1; d = 1 -> mg = 0 (no division)
static void h16_magic(unsigned d, unsigned* mg, unsigned* sh) {
    if (d <= 1) { *mg = 0; *sh = 0; return; }
    unsigned sft = 0;
    while ((1ull << sft) < d) ++sft;
    *mg = (unsigned)(((1ull << (31 + sft)) + d - 1) / d);
    *sh = sft - 1;
}

static thread_local const mrcnn_bwd_epilogue_h16* g_h16_fb = nullptr;   // set around the call by mrcnn_conv2d_dgrad_ep_h16

// Which 16-bit forward kernel a shape takes: 2 = small tile (64 x 64), 1 = large tile (256 x 128 / 256 x 256), 0 = none.
static int h16_fwd_kernel_for(const mrcnn_conv_desc* d, const void* res) {
    if (!d || d->Cin % 32 || d->KH * d->KW > 64) return 0;
    const long long M = (long long)d->N * d->OH * d->OW;
    const bool dense = d->out_mode == MRCNN_OUT_NHWC && d->out_w_stride == d->Cout && d->out_h_stride == (int64_t)d->OW * d->Cout &&
                       d->out_n_stride == (int64_t)d->OH * d->OW * d->Cout;
    const bool large_ok = d->Cout % 128 == 0 && !res && (dense || d->out_mode == MRCNN_OUT_DECONV2);
    const bool small_ok = d->Cin % 64 == 0 && d->Cout % 64 == 0 && d->out_mode == MRCNN_OUT_NHWC && d->cmod == d->Cout;
    const char* force = getenv("MRCNN_H16_SMALL");                   // A/B: 0 = never, 1 = whenever possible
    if (force && force[0] == '0') return large_ok ? 1 : 0;
    if (force && force[0] == '1') return small_ok ? 2 : (large_ok ? 1 : 0);
    // the large tiles need a few hundred workgroups to fill the chip; below that the short dependent chains of the small
    // tiles win (and they are the only ones with a residual port and strided output)
    const long long large_wgs = ((M + 255) / 256) * (d->Cout / 128);
    if (large_ok && (large_wgs >= 384 || !small_ok)) return 1;
    return small_ok ? 2 : 0;
}

extern "C" int mrcnn_conv2d_fwd_h16_supported(const mrcnn_conv_desc* d, int has_res) {
    return h16_fwd_kernel_for(d, has_res ? (const void*)d : nullptr) != 0;
}

extern "C" int mrcnn_conv2d_fwd_h16_res(const mrcnn_conv_desc* d, int dtype, const void* x, const void* w_t, const float* bias,
                                        const float* scale, const float* shift, const void* res, void* out, void* z_out,
                                        void* stream) {
    if (!d || !x || !w_t || !out || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 ||
        d->OH <= 0 || d->OW <= 0 || d->KH * d->KW > 64)
        return MRCNN_ERR_ARG;
    if (d->res_mode == MRCNN_RES_UP2 || (d->res_mode == MRCNN_RES_SAME) != (res != nullptr)) return MRCNN_ERR_ARG;
    int which = h16_fwd_kernel_for(d, res);
    if (!which) return MRCNN_ERR_ARG;
    if (d->out_mode == MRCNN_OUT_DECONV2) {
        if (d->Cout != 4 * d->cmod || d->cmod % 128 || z_out) return MRCNN_ERR_ARG;
    } else if (d->out_mode != MRCNN_OUT_NHWC || d->cmod != d->Cout) {
        return MRCNN_ERR_ARG;
    }
    if (scale && !shift) return MRCNN_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w_t) & 15)) return MRCNN_ERR_ARG;
    const long long M = (long long)d->N * d->OH * d->OW;
    const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 2;
    const long long shift_b = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 2;
    const long long wbytes = (long long)d->KH * d->KW * d->Cin * d->Cout * 2;
    if (M >= (1LL << 31) || xbytes + shift_b >= 0x7FFFFFF0LL || wbytes >= 0x7FFFFFF0LL) return MRCNN_ERR_ARG;
    ConvH16Args a;
    a.x = x; a.wt = w_t; a.bias = bias; a.scale = scale; a.shift = shift; a.out = out; a.z = z_out; a.res = res;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.OH = d->OH; a.OW = d->OW; a.act = d->act;
    a.M = (int)M; a.Ktot = d->KH * d->KW * d->Cin;
    a.x_shift = (unsigned)shift_b; a.x_records = (unsigned)(xbytes + shift_b); a.w_records = (unsigned)wbytes;
    a.out_records = (unsigned)std::min<long long>(M * d->Cout * 2, 0xFFFFFFF0LL);
    a.out_mode = d->out_mode; a.cmod = d->cmod; a.ons = d->out_n_stride; a.ohs = d->out_h_stride; a.ows = d->out_w_stride;
    a.dense = d->out_mode == MRCNN_OUT_NHWC && d->out_w_stride == d->Cout && d->out_h_stride == (int64_t)d->OW * d->Cout &&
              d->out_n_stride == (int64_t)d->OH * d->OW * d->Cout;
    a.ptiles = 0; a.mtile0 = 0;
    a.dbg = nullptr;
    h16_magic((unsigned)(d->OH * d->OW), &a.mg_ohw, &a.sh_ohw);
    h16_magic((unsigned)d->OW, &a.mg_ow, &a.sh_ow);
    if (const char* t = getenv("MRCNN_H16P_TRACE")) a.dbg = (unsigned long long*)strtoull(t, nullptr, 0);   // device buffer of 1024 x u64
    a.fb_act = -1; a.fb_out = a.fb_z = nullptr; a.fb_scale = a.fb_mean = a.fb_rstd = nullptr;
    a.fb_dgamma = a.fb_dbeta = a.fb_dbias = nullptr; a.fb_dy = nullptr; a.fb_gmul = 1.f;
    if (g_h16_fb) {                              // mrcnn_conv2d_dgrad_ep_h16: small-tile kernel only, dense, plain store
        if (which != 2 && !(d->Cin % 64 == 0 && d->Cout % 64 == 0)) return MRCNN_ERR_UNSUPPORTED;
        if (!a.dense || bias || scale || z_out || d->act != MRCNN_ACT_NONE || d->stride != 1) return MRCNN_ERR_UNSUPPORTED;
        which = 2;
        const mrcnn_bwd_epilogue_h16* ep = g_h16_fb;
        a.fb_act = ep->act; a.fb_out = ep->out; a.fb_z = ep->z; a.fb_scale = ep->scale; a.fb_mean = ep->mean; a.fb_rstd = ep->rstd;
        a.fb_dgamma = ep->dgamma; a.fb_dbeta = ep->dbeta; a.fb_dbias = ep->dbias; a.fb_dy = ep->dy; a.fb_gmul = ep->grad_multiplier;
    }
    {   // 16-byte epilogue of the small-tile kernel: every tensor it touches 16-byte aligned, pixel strides multiples of 8 elements
        static const bool on = !(getenv("MRCNN_H16S_EP_VEC") && getenv("MRCNN_H16S_EP_VEC")[0] == '0');
        auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        a.ep_vec = on && d->Cout % 64 == 0 && al(a.out) && al(a.z) && al(a.res) && al(a.fb_out) && al(a.fb_z) && al(a.fb_dy) &&
                   a.ons % 8 == 0 && a.ohs % 8 == 0 && a.ows % 8 == 0;
    }
    hipStream_t s = (hipStream_t)stream;
    if (which == 2) {
        const unsigned blocks = (unsigned)(((M + 63) / 64) * (d->Cout / 64));
        if (dtype == MRCNN_DTYPE_F16) hipLaunchKernelGGL(conv_fwd_h16s_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv_fwd_h16s_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, a);
        return mrcnn_launch_status();
    }
    // Large tiles (read per call so that tests and A/B timings can switch: MRCNN_H16_TILE = big | small | ring):
    //   big    256 x 256, 8 waves, 4-stage ring, one workgroup per CU -- Cout % 256 == 0
    //   phase  256 x 256, 8 waves in two staggered groups, persistent (default from one full round of tiles on)
    //   small  256 x 128, 4 waves, double buffered, 3 workgroups per CU (the round-1 kernel; the default below that)
    //   ring   256 x 128 with a 3-slot ring (round-1 experiment)
    const char* tile = getenv("MRCNN_H16_TILE");
    const long long big_tiles = ((M + 255) / 256) * (d->Cout / 256);
    // small and big stop at ~30 % of the matrix peak (745-810 TFLOP/s on the mask-head shape): per K-step a wave spends
    // about as long issuing its LDS-DMA pieces and waiting for operand reads as the matrix pipe needs for its MFMAs, and
    // the two waves of a SIMD do it in lockstep.  De-phasing them with duplicated bodies or a selector loop made hipcc spill
    // the accumulators; the phased kernel gets the stagger from one extra barrier instead (1.0-1.06 PFLOP/s here, 1.15-1.2
    // where the tiles are whole rounds; measured shader clock under it: 1.96 GHz, tools/h16p_trace.py).
    bool big = false;
    if (tile && !strcmp(tile, "big")) big = d->Cout % 256 == 0;
    if (tile && !strcmp(tile, "wave128") && d->Cout % 256 == 0 && a.dense && !res && d->Cin % 32 == 0 &&
        !((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(z_out)) & 15)) {
        const unsigned blocks = (unsigned)big_tiles;
        if (const char* dbg = getenv("MRCNN_H16W_DBG")) a.mtile0 = atoi(dbg);      // timing experiments: 1 = no MFMAs, 2 = no operand reads
        if (dtype == MRCNN_DTYPE_F16) hipLaunchKernelGGL((conv_fwd_h16_kernel<_Float16, 4, 2, 2, 4>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_fwd_h16_kernel<__bf16, 4, 2, 2, 4>), dim3(blocks), dim3(256), 0, s, a);
        return mrcnn_launch_status();
    }
    // the transposed convolution's pixel-shuffle store (MRCNN_OUT_DECONV2) is the same kernel with another row address
    const bool deconv_p = d->out_mode == MRCNN_OUT_DECONV2 && d->cmod % 64 == 0 && d->cmod <= 512 && d->Cout == 4 * d->cmod && !z_out;
    const bool phased_ok = d->Cout % 256 == 0 && (deconv_p || (d->Cout <= 512 && a.dense)) && d->Cin % 64 == 0 && d->KH * d->KW <= 31 &&
                           !res && (d->act == MRCNN_ACT_NONE || d->act == MRCNN_ACT_RELU) && M * d->Cout * 2 < 0xFFFFFFF0LL;
    // phase: persistent 256 x 256 tiles (see conv_fwd_h16p_kernel), the default where a shape has at least one full round of
    // them; a last partial round of at most half the CUs goes to the small-tile kernel instead (a 256-row tile costs a full
    // tile time however few there are; 64-row tiles spread the same rows over the whole chip)
    const int cus = mrcnn_num_cus();
    const bool want_phase = tile ? !strcmp(tile, "phase") : (big_tiles >= cus && g_mrcnn_h16_phase);
    if (phased_ok && want_phase) {
        const int ntn = d->Cout / 256;
        long long own = big_tiles;
        const long long full = big_tiles / cus * cus / ntn * ntn;
        if (full > 0 && big_tiles - full > 0 && (big_tiles - full) * 2 <= cus && !getenv("MRCNN_H16P_NO_SPLIT") && !deconv_p) own = full;
        a.ptiles = (int)own;
        unsigned blocks = (unsigned)std::min<long long>(own, cus);   // one workgroup per CU (LDS)
        if (const char* g = getenv("MRCNN_H16P_GRID")) blocks = (unsigned)std::min<long long>(own, atoi(g));   // experiments
        // slab form (conv_fwd_h16q_kernel): 3 x 3 / stride 1 / pad 1 on maps no wider than 14 pixels (halo of W + 1 <= 15 rows), an
        // even number of 64-channel chunks (the two slab buffers alternate by chunk parity across tiles): the mask head's layers
        // OFF by default: correct (test_conv_fwd_h16_slab) and 8-12 % SLOWER than the per-tap staging on the mask-head layer (0.419 ->
        // 0.474 ms; with the border masks knocked out 0.422-0.431 ms: halving the pixel pieces buys nothing, the masks cost) --
        // tools/h16_slab_probe.py, DESIGN 4.1c
        static const bool slab_env = getenv("MRCNN_H16_SLAB") && getenv("MRCNN_H16_SLAB")[0] == '1';
        const bool slab_on = g_mrcnn_h16_slab < 0 ? slab_env : g_mrcnn_h16_slab != 0;
        const bool slab = slab_on && !deconv_p && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
                          d->OH == d->H && d->OW == d->W && d->W <= 14 && d->Cin % 128 == 0 && a.dense && d->Cout <= 512;
        if (slab) {
            if (dtype == MRCNN_DTYPE_F16) {
                if (z_out) hipLaunchKernelGGL((conv_fwd_h16q_kernel<_Float16, true>), dim3(blocks), dim3(512), 0, s, a);
                else hipLaunchKernelGGL((conv_fwd_h16q_kernel<_Float16, false>), dim3(blocks), dim3(512), 0, s, a);
            } else {
                if (z_out) hipLaunchKernelGGL((conv_fwd_h16q_kernel<__bf16, true>), dim3(blocks), dim3(512), 0, s, a);
                else hipLaunchKernelGGL((conv_fwd_h16q_kernel<__bf16, false>), dim3(blocks), dim3(512), 0, s, a);
            }
        } else if (deconv_p) {
            if (dtype == MRCNN_DTYPE_F16) hipLaunchKernelGGL((conv_fwd_h16p_kernel<_Float16, false, true>), dim3(blocks), dim3(512), 0, s, a);
            else hipLaunchKernelGGL((conv_fwd_h16p_kernel<__bf16, false, true>), dim3(blocks), dim3(512), 0, s, a);
        } else if (dtype == MRCNN_DTYPE_F16) {
            if (z_out) hipLaunchKernelGGL((conv_fwd_h16p_kernel<_Float16, true, false>), dim3(blocks), dim3(512), 0, s, a);
            else hipLaunchKernelGGL((conv_fwd_h16p_kernel<_Float16, false, false>), dim3(blocks), dim3(512), 0, s, a);
        } else {
            if (z_out) hipLaunchKernelGGL((conv_fwd_h16p_kernel<__bf16, true, false>), dim3(blocks), dim3(512), 0, s, a);
            else hipLaunchKernelGGL((conv_fwd_h16p_kernel<__bf16, false, false>), dim3(blocks), dim3(512), 0, s, a);
        }
        if (own < big_tiles) {
            const long long mdone = own / ntn * 256;
            a.mtile0 = (int)(mdone / 64);
            const unsigned rblocks = (unsigned)(((M - mdone + 63) / 64) * (d->Cout / 64));
            if (dtype == MRCNN_DTYPE_F16) hipLaunchKernelGGL(conv_fwd_h16s_kernel<_Float16>, dim3(rblocks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(conv_fwd_h16s_kernel<__bf16>, dim3(rblocks), dim3(256), 0, s, a);
        }
        return mrcnn_launch_status();
    }
    const bool ring = (tile && !strcmp(tile, "ring")) || getenv("MRCNN_H16_RING") != nullptr;
    if (big) {
        const unsigned blocks = (unsigned)big_tiles;
        if (dtype == MRCNN_DTYPE_F16) hipLaunchKernelGGL((conv_fwd_h16_kernel<_Float16, 4, 2, 4>), dim3(blocks), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv_fwd_h16_kernel<__bf16, 4, 2, 4>), dim3(blocks), dim3(512), 0, s, a);
        return mrcnn_launch_status();
    }
    const unsigned blocks = (unsigned)(((M + 255) / 256) * (d->Cout / 128));
    if (ring) {
        if (dtype == MRCNN_DTYPE_F16) hipLaunchKernelGGL((conv_fwd_h16_kernel<_Float16, 3, 2, 2>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_fwd_h16_kernel<__bf16, 3, 2, 2>), dim3(blocks), dim3(256), 0, s, a);
    } else if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL((conv_fwd_h16_kernel<_Float16, 2, 2, 2>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv_fwd_h16_kernel<__bf16, 2, 2, 2>), dim3(blocks), dim3(256), 0, s, a);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_conv2d_fwd_h16(const mrcnn_conv_desc* d, int dtype, const void* x, const void* w_t, const float* bias,
                                    const float* scale, const float* shift, void* out, void* z_out, void* stream) {
    return mrcnn_conv2d_fwd_h16_res(d, dtype, x, w_t, bias, scale, shift, nullptr, out, z_out, stream);
}

extern "C" int mrcnn_conv2d_dgrad_ep_h16(const mrcnn_conv_desc* d, int dtype, const void* dz, const void* w_t, const void* res,
                                         void* dz_below, const mrcnn_bwd_epilogue_h16* ep, void* stream) {
    if (!ep) return MRCNN_ERR_ARG;
    if ((ep->act != MRCNN_ACT_NONE && ep->act != MRCNN_ACT_RELU) || (ep->act == MRCNN_ACT_RELU && !ep->out)) return MRCNN_ERR_ARG;
    if (ep->dgamma && (!ep->z || !ep->mean || !ep->rstd)) return MRCNN_ERR_ARG;
    g_h16_fb = ep;
    const int rc = mrcnn_conv2d_fwd_h16_res(d, dtype, dz, w_t, nullptr, nullptr, nullptr, res, dz_below, nullptr, stream);
    g_h16_fb = nullptr;
    return rc;
}

extern "C" int mrcnn_mask_out_fwd_h16(int dtype, const void* up, const float* w_mask, const float* b_mask, float* mask_out,
                                      int64_t npix, int Cd, int C, void* stream) {
    if (!up || !w_mask || !mask_out || npix <= 0 || Cd < 256 || Cd % 256 || C < 1 || C > 16 || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (Cd == 256 && C <= 4) {
        const unsigned blocks4 = (unsigned)cdiv64(npix, 256);
        if (dtype == MRCNN_DTYPE_F16)
            hipLaunchKernelGGL(mask_out_fwd_h16_c4_kernel<_Float16>, dim3(blocks4), dim3(256), 0, (hipStream_t)stream,
                               (const _Float16*)up, w_mask, b_mask, mask_out, (long long)npix, C);
        else
            hipLaunchKernelGGL(mask_out_fwd_h16_c4_kernel<__bf16>, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, (const __bf16*)up,
                               w_mask, b_mask, mask_out, (long long)npix, C);
        return mrcnn_launch_status();
    }
    const unsigned blocks = (unsigned)cdiv64(npix, 64);
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(mask_out_fwd_h16_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)up,
                           w_mask, b_mask, mask_out, (long long)npix, Cd, C);
    else
        hipLaunchKernelGGL(mask_out_fwd_h16_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)up, w_mask,
                           b_mask, mask_out, (long long)npix, Cd, C);
    return mrcnn_launch_status();
}

template <typename T>
static void launch_mask_out_bwd_h16(const float* dm, const float* m, const void* up, const float* wm, void* dzg, float* dw, float* dbm,
                                    float* dbd, long long npix, int H, int W, int Cd, int C, float ls, hipStream_t s) {
    const dim3 grid((unsigned)cdiv64(npix, 512)), block(Cd);
    static const int v8 = getenv("MRCNN_MASK_OUT_BWD_V8") ? atoi(getenv("MRCNN_MASK_OUT_BWD_V8")) : 1;
    if (v8 && Cd == 256 && C <= 4 && ((reinterpret_cast<uintptr_t>(up) | reinterpret_cast<uintptr_t>(dzg)) & 15) == 0) {
        hipLaunchKernelGGL((mask_out_bwd_h16_v8_kernel<T, 4>), grid, dim3(256), 0, s, dm, m, (const T*)up, wm, (T*)dzg, dw, dbm, dbd, npix,
                           H, W, C, ls);
        return;
    }
    if (C <= 4)
        hipLaunchKernelGGL((mask_out_bwd_h16_kernel<T, 4>), grid, block, 0, s, dm, m, (const T*)up, wm, (T*)dzg, dw, dbm, dbd, npix, H, W, Cd, C, ls);
    else if (C <= 8)
        hipLaunchKernelGGL((mask_out_bwd_h16_kernel<T, 8>), grid, block, 0, s, dm, m, (const T*)up, wm, (T*)dzg, dw, dbm, dbd, npix, H, W, Cd, C, ls);
    else
        hipLaunchKernelGGL((mask_out_bwd_h16_kernel<T, 16>), grid, block, 0, s, dm, m, (const T*)up, wm, (T*)dzg, dw, dbm, dbd, npix, H, W, Cd, C, ls);
}

extern "C" int mrcnn_mask_out_bwd_h16(int dtype, const float* d_mask_out, const float* mask_out, const void* up, const float* w_mask,
                                      void* dzg, float* dw_mask, float* db_mask, float* db_deconv, int64_t M, int H, int W, int Cd,
                                      int C, float loss_scale, void* stream) {
    if (!d_mask_out || !mask_out || !up || !w_mask || !dzg || !dw_mask || !db_mask || !db_deconv || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (M <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cd < 64 || Cd > 1024 || (Cd & 63) || C < 1 || C > 16) return MRCNN_ERR_ARG;
    const long long npix = (long long)M * H * W;
    if (dtype == MRCNN_DTYPE_F16)
        launch_mask_out_bwd_h16<_Float16>(d_mask_out, mask_out, up, w_mask, dzg, dw_mask, db_mask, db_deconv, npix, H, W, Cd, C, loss_scale,
                                          (hipStream_t)stream);
    else
        launch_mask_out_bwd_h16<__bf16>(d_mask_out, mask_out, up, w_mask, dzg, dw_mask, db_mask, db_deconv, npix, H, W, Cd, C, loss_scale,
                                        (hipStream_t)stream);
    return mrcnn_launch_status();
}

// Every 16-bit weight image of the model in ONE launch: table[l] = {offset of the float32 HWIO kernel in the flat parameter
// buffer (floats), W^T image pointer, data-gradient image pointer (or 0), KH, KW, Cin, Cout, first tile}; a tile is 32 ci x
// 32 co of one tap, as in weights_to_h16_kernel.  The images are refreshed once per optimiser step; layer by layer that was
// 77 launches of ~5 us at the head of the ResNet-101 step (knock-out: 0.6 ms of the 19 ms 512 x 512 step).
struct H16ImageEntry { long long off; unsigned long long wf, wd; int KH, KW, Cin, Cout, first_tile, pad; };

template <typename T>
__global__ void weights_to_h16_batched_kernel(const float* __restrict__ params, const H16ImageEntry* __restrict__ table, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;                         // last entry with first_tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const H16ImageEntry e = table[lo];
    int t = blockIdx.x - e.first_tile;
    const int cot = (e.Cout + 31) / 32, cit = (e.Cin + 31) / 32;
    const int co0 = (t % cot) * 32; t /= cot;
    const int ci0 = (t % cit) * 32;
    const int tap = t / cit;
    const int kh = tap / e.KW, kw = tap % e.KW;
    const int tap_t = (e.KH - 1 - kh) * e.KW + (e.KW - 1 - kw);
    const float* w = params + e.off;
    T* wt_f = (T*)e.wf;
    T* wt_d = (T*)e.wd;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long Kf = (long long)e.KH * e.KW * e.Cin, Kd = (long long)e.KH * e.KW * e.Cout;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        const float v = (ci < e.Cin && co < e.Cout) ? w[((long long)tap * e.Cin + ci) * e.Cout + co] : 0.f;
        tile[r][tx] = v;
        if (wt_d && ci < e.Cin && co < e.Cout) wt_d[(long long)ci * Kd + (long long)tap_t * e.Cout + co] = (T)v;
    }
    __syncthreads();
    if (wt_f)
        for (int r = ty; r < 32; r += 8) {
            const int co = co0 + r, ci = ci0 + tx;
            if (co < e.Cout && ci < e.Cin) wt_f[(long long)co * Kf + (long long)tap * e.Cin + ci] = (T)tile[tx][r];
        }
}

extern "C" int mrcnn_weights_to_h16_batched(const float* params, const void* table, int n_layers, int total_tiles, int dtype,
                                            void* stream) {
    if (!params || !table || n_layers <= 0 || total_tiles <= 0 || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(weights_to_h16_batched_kernel<_Float16>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                           params, (const H16ImageEntry*)table, n_layers);
    else
        hipLaunchKernelGGL(weights_to_h16_batched_kernel<__bf16>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                           params, (const H16ImageEntry*)table, n_layers);
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

extern "C" int mrcnn_epilogue_bwd_h16_dy(int dtype, const void* dout, const void* out, const void* z, const float* scale,
                                         const float* mean, const float* rstd, void* dz_out, void* dy_out, float* dgamma, float* dbeta,
                                         float* dbias, int64_t M, int C, int act, float grad_multiplier, void* stream) {
    if (!dout || !dz_out || M <= 0 || C < 16 || C > 4096 || (C & (C - 1)) || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if ((act != MRCNN_ACT_NONE && act != MRCNN_ACT_RELU) || (act == MRCNN_ACT_RELU && !out)) return MRCNN_ERR_ARG;
    if (dgamma && (!z || !mean || !rstd)) return MRCNN_ERR_ARG;
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al(dout) || !al(out) || !al(z) || !al(dz_out) || !al(dy_out)) return MRCNN_ERR_ARG;
    const int c8n = C >> 3;                                   // C is a power of two >= 16
    int lg = 0;
    while ((2 << lg) <= (c8n < 16 ? c8n : 16)) ++lg;           // 16 lanes x 8 channels per row at most
    const long long R = 256 >> lg;
    long long rows_per_block = cdiv64(M, 2048);
    rows_per_block = cdiv64(rows_per_block, R * EPI16_U) * R * EPI16_U;   // whole batches
    const dim3 grid((unsigned)cdiv64(M, rows_per_block), (unsigned)(c8n >> lg));
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(epilogue_bwd_h16_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)dout, (const _Float16*)out, (const _Float16*)z, scale, mean, rstd, (_Float16*)dz_out,
                           dgamma, dbeta, dbias, (long long)M, C, act, rows_per_block, lg, grad_multiplier, (_Float16*)dy_out);
    else
        hipLaunchKernelGGL(epilogue_bwd_h16_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)dout, (const __bf16*)out, (const __bf16*)z, scale, mean, rstd, (__bf16*)dz_out, dgamma,
                           dbeta, dbias, (long long)M, C, act, rows_per_block, lg, grad_multiplier, (__bf16*)dy_out);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_epilogue_bwd_h16(int dtype, const void* dout, const void* out, const void* z, const float* scale,
                                      const float* mean, const float* rstd, void* dz_out, float* dgamma, float* dbeta,
                                      float* dbias, int64_t M, int C, int act, float grad_multiplier, void* stream) {
    return mrcnn_epilogue_bwd_h16_dy(dtype, dout, out, z, scale, mean, rstd, dz_out, nullptr, dgamma, dbeta, dbias, M, C, act,
                                     grad_multiplier, stream);
}

extern "C" int mrcnn_cast_to_h16(const float* src, void* dst, int64_t n, int dtype, float multiplier, void* stream) {
    if (!src || !dst || n < 0 || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (n == 0) return 0;
    if (reinterpret_cast<uintptr_t>(src) & 15) return MRCNN_ERR_ARG;
    const unsigned blocks = (unsigned)cdiv64(cdiv64(n, 4), 256);
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL(cast_to_h16_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst,
                           (long long)n, multiplier);
    else
        hipLaunchKernelGGL(cast_to_h16_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst,
                           (long long)n, multiplier);
    return mrcnn_launch_status();
}

template <bool ACC>
static int launch_cast_from_h16(const void* src, float* dst, int64_t n, int dtype, float multiplier, void* stream) {
    if (!src || !dst || n < 0 || !h16_dtype_ok(dtype)) return MRCNN_ERR_ARG;
    if (n == 0) return 0;
    const int vec = (n % 8 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const unsigned blocks = (unsigned)cdiv64(cdiv64(n, 8), 256);
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL((cast_from_h16_kernel<_Float16, ACC>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src,
                           dst, (long long)n, multiplier, vec);
    else
        hipLaunchKernelGGL((cast_from_h16_kernel<__bf16, ACC>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, dst,
                           (long long)n, multiplier, vec);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_cast_from_h16(const void* src, float* dst, int64_t n, int dtype, float multiplier, void* stream) {
    return launch_cast_from_h16<false>(src, dst, n, dtype, multiplier, stream);
}

extern "C" int mrcnn_axpy_from_h16(const void* src, float* dst, int64_t n, int dtype, float multiplier, void* stream) {
    return launch_cast_from_h16<true>(src, dst, n, dtype, multiplier, stream);
}
