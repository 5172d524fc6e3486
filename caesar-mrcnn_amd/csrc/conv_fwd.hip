// Implicit-GEMM convolution forward for gfx950 on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32).  Stands in for KL.Conv2D / Dense / Conv2DTranspose(2x2,s2) + frozen
// BatchNorm + Add + Activation of the reference graph (mrcnn/model.py:99-210, 916-957, 986-1091,
// 2005-2022).  GEMM view: M = N*OH*OW pixels, N = Cout, K = KH*KW*Cin; A is gathered on the fly
// from the NHWC input (never materialised), B is the HWIO weight matrix as stored by Keras.
//
// Tiling: BM x BN output tile per workgroup, K-step 32, WM x WN waves, each wave owns
// (BM/WM) x (BN/WN) as TM x TN MFMA tiles of 32x32.  Within a K-step the two half-waves take
// k in [0,16) and [16,32) (the MFMA's two k-slices), so a lane reads 16 contiguous k of its A row
// with four ds_read_b128 and its B column with conflict-free ds_read_b32.  Next K-step's global loads
// are issued into registers before the MFMAs of the current one (register-staged prefetch).
#include "common.h"

struct ConvArgs {
    const float* x; const float* w; const float* bias; const float* scale; const float* shift;
    const float* res; float* out; float* z;
    int N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, OH, OW;
    int act, res_mode, out_mode, cmod;
    long long ons, ohs, ows;
    int M, Ktot, nk, fastA, vecB, dense;
    FastDiv d_ohw, d_ow;               // exact division of output-pixel indices (non-dense stores: deconv, concat, UP2)
    FastDiv d_c4;                      // Cout / 4 (vector split-K epilogue)
    float* slab; int ksplit, ksteps;   // split-K: partial sums [ksplit][M][Cout], K-steps per split
    int bat_rows;                      // LDS-DMA kernel as a batched GEMM (Winograd domain): rows [b * bat_rows, (b + 1) * bat_rows) use the
                                       // weight matrix w + b * Ktot * Cout; 0 = one weight matrix (every convolution)
    int mtile0;                        // LDS-DMA split-K kernel as the TAIL launch of a large layer: first 128-row tile it owns; slab rows
                                       // are relative to it ([ksplit][M - 128 mtile0][Cout])
    // fused backward epilogue (mrcnn_conv2d_dgrad_ep, LDS-DMA kernel only): the result y is the gradient w.r.t. the
    // activated output of the layer below; that layer's epilogue backward is applied before the store
    const float* fb_out; const float* fb_z; const float* fb_scale; const float* fb_mean; const float* fb_rstd;
    float* fb_dgamma; float* fb_dbeta; float* fb_dbias; int fb_act;
    float* fb_dy;                      // optional: y * act' as a second output (split-K reduction only)
};

// Padding taps, rows past M and columns past Cout fetch from here: loads stay unconditional (no divergent branch,
// no conservative wait at the join).
__device__ __attribute__((aligned(64))) float g_conv_zero_page[16];

// One 32x32 accumulator tile: lane holds column n, rows mbase + (r&3) + 8*(r>>2).
__device__ __forceinline__ void conv_epilogue_tile(const ConvArgs& p, const f32x16& acc, int mbase, int n) {
    if (n >= p.Cout) return;
    const int ab_tile = n / p.cmod;              // (a*2+b) of the transposed-conv column; 0 otherwise (cmod == Cout)
    const int c = n - ab_tile * p.cmod;
    const float bias = p.bias ? p.bias[c] : 0.f;
    const float sc = p.scale ? p.scale[c] : 1.f;
    const float sh = p.scale ? p.shift[c] : 0.f;
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (m >= p.M) continue;
        float zv = acc[r] + bias;
        long long addr, raddr;
        if (p.dense && p.res_mode != MRCNN_RES_UP2) {
            addr = (long long)m * p.Cout + n;
            raddr = addr;
        } else {
            const int ni = (int)fast_div((unsigned)m, p.d_ohw), rem = m - ni * ohw;
            const int oh = (int)fast_div((unsigned)rem, p.d_ow), ow = rem - oh * p.OW;
            if (p.out_mode == MRCNN_OUT_DECONV2) {
                const int ab = ab_tile;
                addr = (long long)ni * p.ons + (long long)(2 * oh + (ab >> 1)) * p.ohs +
                       (long long)(2 * ow + (ab & 1)) * p.ows + c;
            } else {
                addr = (long long)ni * p.ons + (long long)oh * p.ohs + (long long)ow * p.ows + n;
            }
            raddr = addr;
            if (p.res_mode == MRCNN_RES_UP2)
                raddr = (((long long)ni * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1)) * p.Cout + n;
        }
        if (p.z) p.z[addr] = zv;
        float y = sc * zv + sh;
        if (p.res_mode != MRCNN_RES_NONE) y += p.res[raddr];
        if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
        else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
        p.out[addr] = y;
    }
}

// Fused-backward form of conv_epilogue_tile (dense NHWC only): y = acc (+ res) is d(loss)/d(out_below);
//   g = y * act'(out_below),  dz = g * scale_below  -> stored;  sums: dbeta += g, dgamma += g*(z-mean)*rstd, dbias += dz
// (mrcnn_epilogue_bwd applied in the producer).  s[0..2] accumulate this lane's column sums.
__device__ __forceinline__ void conv_epilogue_tile_bwd(const ConvArgs& p, const f32x16& acc, int mbase, int n, float* s) {
    const float sc = p.fb_scale ? p.fb_scale[n] : 1.f;
    const float mu = p.fb_dgamma ? p.fb_mean[n] : 0.f, rs = p.fb_dgamma ? p.fb_rstd[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (m >= p.M) continue;
        const long long addr = (long long)m * p.Cout + n;
        float g = acc[r];
        if (p.res_mode != MRCNN_RES_NONE) g += p.res[addr];
        if (p.fb_act == MRCNN_ACT_RELU) g = p.fb_out[addr] > 0.f ? g : 0.f;
        const float dz = g * sc;
        p.out[addr] = dz;
        s[0] += g;
        if (p.fb_dgamma) s[1] += g * (p.fb_z[addr] - mu) * rs;
        s[2] += dz;
    }
}

// K-steps of operand tiles held in registers ahead of the LDS stores.  1 = classic double buffering.  4 was measured
// (whole split-K slices issued up front): no change on detect (3.96 vs 3.98 ms) or the training step -- these launches
// sit on the ~4.8 us dependent-launch floor, not on the memory round trips of their 4-5 K-steps.
#ifndef CONV_PREFETCH_STEPS
#define CONV_PREFETCH_STEPS 1
#endif
template <int BM, int BN, int WM, int WN, bool FAST>
__device__ __forceinline__ void conv_fwd_body(const ConvArgs& p, const int block) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AST = 36;          // A row stride (floats): 32 + 4 pad -> conflict-free b128 reads
    constexpr int BST = BN + 4;
    constexpr int AV = BM * 8 / NT;  // float4 loads of A per thread per K-step
    constexpr int BV = 8 * BN / NT;  // float4 loads of B per thread per K-step
    constexpr int AROWSTEP = NT / 8;
    __shared__ __attribute__((aligned(16))) float lds[BM * AST + 32 * BST];
    float* As = lds;
    float* Bs = lds + BM * AST;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = (p.Cout + BN - 1) / BN;
    const int mtiles = (p.M + BM - 1) / BM;
    const int kz = block / (mtiles * ntiles);
    const int tile = block - kz * (mtiles * ntiles);
    const int mtile = tile / ntiles, ntile = tile % ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int ks_begin = kz * p.ksteps;
    const int ks_end = (ks_begin + p.ksteps < p.nk) ? ks_begin + p.ksteps : p.nk;

    // ---- per-thread tile-load bookkeeping --------------------------------------------------------
    // FAST (Cin % 32 == 0, Cout % 4 == 0, 16-byte aligned x/w): every K-step lies inside one filter tap,
    // so a lane's A address is  row_ptr + uniform_offset(tap, ci0)  and its validity one bit of a per-row
    // tap mask; B is  col_ptr + uniform_offset(k).  Two 64-bit adds per load, no per-step index math.
    const int a_c4 = tid & 7;
    const int ohw = p.OH * p.OW;
    const float* a_ptr[AV];
    unsigned long long a_mask[AV];
    int a_ih0[AV], a_iw0[AV];
    long long a_nb[AV];
    bool a_ok[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        int m = m0 + (tid >> 3) + i * AROWSTEP;
        a_ok[i] = m < p.M;
        int mm = a_ok[i] ? m : 0;
        int n = mm / ohw, rem = mm - n * ohw;
        int oh = rem / p.OW, ow = rem - oh * p.OW;
        a_ih0[i] = oh * p.stride - p.pad_t;
        a_iw0[i] = ow * p.stride - p.pad_l;
        a_nb[i] = (long long)n * p.H * p.W * p.Cin;
        if (FAST) {
            a_ptr[i] = p.x + a_nb[i] + ((long long)a_ih0[i] * p.W + a_iw0[i]) * p.Cin + a_c4 * 4;
            unsigned long long mk = 0ull;
            if (a_ok[i])
                for (int t = 0; t < p.KH * p.KW; ++t) {
                    int th = t / p.KW, tw = t - th * p.KW;
                    if ((unsigned)(a_ih0[i] + th) < (unsigned)p.H && (unsigned)(a_iw0[i] + tw) < (unsigned)p.W) mk |= 1ull << t;
                }
            a_mask[i] = mk;
        }
    }
    const float* b_ptr[BV];
    bool b_ok[BV];
#pragma unroll
    for (int i = 0; i < BV; ++i) {
        const int idx = tid + i * NT;
        const int krow = idx / (BN / 4), c4 = idx % (BN / 4);
        b_ok[i] = n0 + c4 * 4 < p.Cout;
        b_ptr[i] = p.w + (long long)krow * p.Cout + n0 + c4 * 4;
    }

    constexpr int PF = (BM * BN > 64 * 128 && CONV_PREFETCH_STEPS > 2) ? 2 : CONV_PREFETCH_STEPS;   // big tiles: register budget
    f32x4 ra[PF][AV], rb[PF][BV];
    int kh = 0, kw = 0, ci0 = 0, tap = 0;   // fast-path K-step position
    if (FAST && ks_begin > 0) {
        const int k0 = ks_begin * 32;
        tap = k0 / p.Cin;
        ci0 = k0 - tap * p.Cin;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }

    auto load_tiles = [&](int ks, f32x4 (&ra)[AV], f32x4 (&rb)[BV]) {
        if (FAST) {
            const long long aoff = ((long long)kh * p.W + kw) * p.Cin + ci0;
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                const bool v = (a_mask[i] >> tap) & 1ull;
                ra[i] = *(const f32x4*)(v ? a_ptr[i] + aoff : g_conv_zero_page);
            }
            ci0 += 32;
            if (ci0 >= p.Cin) { ci0 = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }
            const long long boff = (long long)ks * 32 * p.Cout;
#pragma unroll
            for (int i = 0; i < BV; ++i) {
                if (p.vecB) {
                    rb[i] = *(const f32x4*)(b_ok[i] ? b_ptr[i] + boff : g_conv_zero_page);
                } else {                 // ragged Cout (RPN heads: 6 / 12 columns): scalar weight loads
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    const int nb = n0 + ((tid + i * NT) % (BN / 4)) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nb + e < p.Cout) v[e] = b_ptr[i][boff + e];
                    rb[i] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = ks * 32 + a_c4 * 4 + e;
                    if (a_ok[i] && k < p.Ktot) {
                        int tp = k / p.Cin, ci = k - tp * p.Cin;
                        int tkh = tp / p.KW, tkw = tp - tkh * p.KW;
                        int ih = a_ih0[i] + tkh, iw = a_iw0[i] + tkw;
                        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                            v[e] = p.x[a_nb[i] + ((long long)ih * p.W + iw) * p.Cin + ci];
                    }
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < BV; ++i) {
                int idx = tid + i * NT;
                int krow = idx / (BN / 4), c4 = idx % (BN / 4);
                int k = ks * 32 + krow, n = n0 + c4 * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.Ktot) {
                    const float* ptr = p.w + (long long)k * p.Cout + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Cout) v[e] = ptr[e];
                }
                rb[i] = v;
            }
        }
    };
    auto store_tiles = [&](const f32x4 (&ra)[AV], const f32x4 (&rb)[BV]) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            int r = (tid >> 3) + i * AROWSTEP;
            *(f32x4*)&As[r * AST + a_c4 * 4] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BV; ++i) {
            int idx = tid + i * NT;
            int krow = idx / (BN / 4), c4 = idx % (BN / 4);
            *(f32x4*)&Bs[krow * BST + c4 * 4] = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (ks_begin + j < ks_end) load_tiles(ks_begin + j, ra[j], rb[j]);
    for (int ks = ks_begin; ks < ks_end; ks += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (ks + j < ks_end) {                                  // uniform
                store_tiles(ra[j], rb[j]);                          // waits for step ks+j only: younger loads stay in flight
                __syncthreads();
                if (ks + j + PF < ks_end) load_tiles(ks + j + PF, ra[j], rb[j]);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    f32x4 av[TM];
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        av[a] = *(const f32x4*)&As[(wm * TM * 32 + a * 32 + li) * AST + lh * 16 + t4 * 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float bv[TN];
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            bv[b] = Bs[(lh * 16 + t4 * 4 + e) * BST + wn * TN * 32 + b * 32 + li];
#pragma unroll
                        for (int a = 0; a < TM; ++a)
#pragma unroll
                            for (int b = 0; b < TN; ++b)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][e], bv[b], acc[a][b], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
    }

    // ---- epilogue: bias, frozen-BN affine, residual, activation ---------------------------------
    const int mw0 = m0 + wm * TM * 32 + 4 * lh, nw0 = n0 + wn * TN * 32 + li;
    if (p.ksplit > 1) {
        float* slab = p.slab + (long long)kz * p.M * p.Cout;
        auto put = [&](const f32x16& c, int mbase, int n) {
            if (n >= p.Cout) return;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < p.M) slab[(long long)m * p.Cout + n] = c[r];
            }
        };
        if constexpr (TM >= 1 && TN >= 1) put(acc[0][0], mw0, nw0);
        if constexpr (TM >= 1 && TN >= 2) put(acc[0][1], mw0, nw0 + 32);
        if constexpr (TM >= 2 && TN >= 1) put(acc[1][0], mw0 + 32, nw0);
        if constexpr (TM >= 2 && TN >= 2) put(acc[1][1], mw0 + 32, nw0 + 32);
        return;
    }
    if constexpr (TM >= 1 && TN >= 1) conv_epilogue_tile(p, acc[0][0], mw0, nw0);
    if constexpr (TM >= 1 && TN >= 2) conv_epilogue_tile(p, acc[0][1], mw0, nw0 + 32);
    if constexpr (TM >= 2 && TN >= 1) conv_epilogue_tile(p, acc[1][0], mw0 + 32, nw0);
    if constexpr (TM >= 2 && TN >= 2) conv_epilogue_tile(p, acc[1][1], mw0 + 32, nw0 + 32);
}

template <int BM, int BN, int WM, int WN, bool FAST>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_fwd_kernel(const ConvArgs p) {
    conv_fwd_body<BM, BN, WM, WN, FAST>(p, (int)blockIdx.x);
}

// Up to CONV_MULTI_MAX independent convolutions in one launch (mrcnn_conv2d_fwd_multi): the workgroups of problem g
// are blocks first[g] .. first[g+1]-1; the split-K reductions of all problems likewise share one launch (efirst).
#define CONV_MULTI_MAX 5
struct ConvMultiArgs {
    ConvArgs a[CONV_MULTI_MAX];
    int first[CONV_MULTI_MAX + 1];
    int efirst[CONV_MULTI_MAX + 1];
    int evec[CONV_MULTI_MAX];
    int n;
};

template <int BM, int BN, int WM, int WN, bool FAST>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_fwd_multi_kernel(const ConvMultiArgs mp) {
    int g = 0;
    while (g + 1 < mp.n && (int)blockIdx.x >= mp.first[g + 1]) ++g;
    conv_fwd_body<BM, BN, WM, WN, FAST>(mp.a[g], (int)blockIdx.x - mp.first[g]);
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant of the 128x128 tile for the large layers (no split-K, Cin % 32 == 0, Cout % 128 == 0):
// tiles go global -> LDS with `global_load_lds_dwordx4` (no staging VGPRs, no ds_write), double buffered,
// K-step 16, one barrier per step.  The DMA writes LDS lane-linearly (wave base + 16 B * lane), so
//   A [128 rows][16 k] is stored row-major with 64-byte rows and its four 16-byte chunks XOR-swizzled on the
//     SOURCE side (lane (r, c) fetches logical chunk c ^ ((r >> 2) & 3)); the MFMA operand read
//     (ds_read_b128 of logical chunk q of row r at physical chunk q ^ ((r >> 2) & 3)) is then conflict-free
//     for the ds_read_b128 lane groups;
//   B [16 k][128 n] is stored as is (512-byte rows) and read with conflict-free ds_read_b32 (lane = n).
// Padding taps / rows past M fetch from a zero page.

__device__ __forceinline__ void conv_glds16(const float* gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void conv_fwd_glds_kernel(const ConvArgs p) {
    constexpr int BM = 128, BN = 128, BK = 16, TM = 2, TN = 2;
    constexpr int AF = BM * BK, BF = BK * BN;                  // floats per tile (8 KiB each)
    __shared__ __attribute__((aligned(16))) float lds[2 * (AF + BF)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.Cout / BN;
    const int mtile = blockIdx.x / ntiles, ntile = blockIdx.x % ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int ohw = p.OH * p.OW;

    // this wave stages A pieces {wave, wave+4} (16 rows each) and B pieces {wave, wave+4} (2 k-rows each)
    const float* a_ptr[2];
    unsigned long long a_mask[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int r = (wave + jj * 4) * 16 + (lane >> 2);
        const int cl = (lane & 3) ^ ((r >> 2) & 3);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / ohw, rem = mm - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        a_ptr[jj] = p.x + (long long)n * p.H * p.W * p.Cin + ((long long)ih0 * p.W + iw0) * p.Cin + cl * 4;
        unsigned long long mk = 0ull;
        if (ok)
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int th = t / p.KW, tw = t - th * p.KW;
                if ((unsigned)(ih0 + th) < (unsigned)p.H && (unsigned)(iw0 + tw) < (unsigned)p.W) mk |= 1ull << t;
            }
        a_mask[jj] = mk;
    }
    const float* b_ptr[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
        b_ptr[jj] = p.w + (long long)((wave + jj * 4) * 2 + (lane >> 5)) * p.Cout + n0 + (lane & 31) * 4;

    int kh = 0, kw = 0, ci0 = 0, tap = 0;
    auto stage = [&](int buf, int ks) {
        float* ab = lds + buf * (AF + BF);
        float* bb = ab + AF;
        const long long aoff = ((long long)kh * p.W + kw) * p.Cin + ci0;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const bool v = (a_mask[jj] >> tap) & 1ull;
            conv_glds16(v ? a_ptr[jj] + aoff : g_conv_zero_page, ab + (wave + jj * 4) * 256);
        }
        // K is walked channel-chunk outer / filter-tap inner: the KH*KW shifted reads of one 64-byte
        // channel chunk follow each other in time, so all but the first hit L2 (the weights just follow)
        const long long boff = ((long long)tap * p.Cin + ci0) * p.Cout;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) conv_glds16(b_ptr[jj] + boff, bb + (wave + jj * 4) * 256);
        ++tap;
        if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; tap = 0; ci0 += BK; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int nk = p.Ktot / BK;
    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) stage(cur ^ 1, ks + 1);
        const float* ab = lds + cur * (AF + BF);
        const float* bb = ab + AF;
#pragma unroll
        for (int q = 0; q < 2; ++q) {              // logical chunk 2*lh + q of this half-wave's 8 k values
            f32x4 av[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = wm * 64 + a * 32 + li;
                av[a] = *(const f32x4*)&ab[row * BK + (((2 * lh + q) ^ ((row >> 2) & 3)) << 2)];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = lh * 8 + q * 4 + e;
                float bv[TN];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[b] = bb[k * BN + wn * 64 + b * 32 + li];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][e], bv[b], acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    const int mw0 = m0 + wm * 64 + 4 * lh, nw0 = n0 + wn * 64 + li;
    conv_epilogue_tile(p, acc[0][0], mw0, nw0);
    conv_epilogue_tile(p, acc[0][1], mw0, nw0 + 32);
    conv_epilogue_tile(p, acc[1][0], mw0 + 32, nw0);
    conv_epilogue_tile(p, acc[1][1], mw0 + 32, nw0 + 32);
}

// Same tile, buffer addressing (the default; the kernel above stays for inputs >= 2 GiB): the per-step
// address arithmetic of the flat variant (64-bit pointer adds, zero-page selects, M0 through a VGPR) costs
// ~1.5 vector instructions per MFMA, and those do not overlap the matrix pipe of the same SIMD (PMC:
// SQ_VALU_MFMA_COEXEC_CYCLES = 0).  Here every lane keeps ONE 32-bit byte offset per piece for the whole
// loop; the filter-tap / channel-chunk displacement is the instruction's scalar offset; a padded tap sets
// the lane's offset out of range and the hardware range check of the buffer descriptor writes zeros into
// LDS (no zero page); the base pointer is moved back by the largest negative tap displacement so that lane
// offsets are never negative; the loop is unrolled over the two LDS buffers so that every LDS address is
// base register + immediate.  Left per step: 3-4 vector instructions per A piece (tap-mask test), none for B.
typedef __attribute__((address_space(3))) void* conv_lds_ptr;
#define CONV_OOB_OFFSET 0xFFFFFFF0u

// SPLIT (mid-size layers: K cut into slices, partial sums to slabs) is a compile-time variant so that the unsplit
// kernel keeps its 96 VGPRs (5 workgroups per CU); with the slice bookkeeping in the same body it grew to 123.
template <bool SPLIT>
__device__ __forceinline__ void conv_fwd_blds_body(const ConvArgs& p, const unsigned x_shift, const unsigned x_records) {
    constexpr int BM = 128, BN = 128, BK = 16, TM = 2, TN = 2;
    constexpr int AF = BM * BK, BF = BK * BN;                  // floats per tile (8 KiB each)
    __shared__ __attribute__((aligned(16))) float lds[2 * (AF + BF)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.Cout / BN;
    const int mt0 = SPLIT ? p.mtile0 : 0;                               // tail launch of a large layer: first tile it owns
    const int tiles = ((p.M + BM - 1) / BM - mt0) * ntiles;
    const int kz = SPLIT ? (int)blockIdx.x / tiles : 0;                 // split-K slice (mid-size layers)
    const int tile = (int)blockIdx.x - kz * tiles;
    const int mtile = tile / ntiles + mt0, ntile = tile % ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int ohw = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - x_shift), 0, x_records, 0x00020000);
    const long long wbat = (!SPLIT && p.bat_rows) ? (long long)(m0 / p.bat_rows) * p.Ktot * p.Cout : 0;   // batched GEMM: the tile's weight matrix
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + wbat + n0), 0, (unsigned)(((long long)p.Ktot * p.Cout - n0) * 4), 0x00020000);

    unsigned a_voff[2];
    unsigned long long a_mask[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int r = (wave + jj * 4) * 16 + (lane >> 2);
        const int cl = (lane & 3) ^ ((r >> 2) & 3);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / ohw, rem = mm - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        const long long off = ((long long)n * p.H * p.W * p.Cin + ((long long)ih0 * p.W + iw0) * p.Cin + cl * 4) * 4 + x_shift;
        a_voff[jj] = (unsigned)off;
        unsigned long long mk = 0ull;
        if (ok)
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int th = t / p.KW, tw = t - th * p.KW;
                if ((unsigned)(ih0 + th) < (unsigned)p.H && (unsigned)(iw0 + tw) < (unsigned)p.W) mk |= 1ull << t;
            }
        a_mask[jj] = mk;
    }
    unsigned b_voff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) b_voff[jj] = (unsigned)((((wave + jj * 4) * 2 + (lane >> 5)) * p.Cout + (lane & 31) * 4) * 4);

    int kh = 0, kw = 0, ci0 = 0, tap = 0;
    const int ks_first = SPLIT ? kz * p.ksteps : 0;             // in K-steps of BK
    if (SPLIT && ks_first > 0) {
        const int taps = p.KH * p.KW;
        const int chunk = ks_first / taps;
        tap = ks_first - chunk * taps;
        ci0 = chunk * BK;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }
    auto stage = [&](float* ab) {                               // ab: LDS buffer (A tile, then B tile)
        float* bb = ab + AF;
        const unsigned soff_a = (unsigned)(((kh * p.W + kw) * p.Cin + ci0) * 4);
        const unsigned soff_b = (unsigned)((tap * p.Cin + ci0) * p.Cout * 4);
        const unsigned long long bit = 1ull << tap;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const unsigned vo = (a_mask[jj] & bit) ? a_voff[jj] : CONV_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (conv_lds_ptr)(ab + (wave + jj * 4) * 256), 16, vo, soff_a, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (conv_lds_ptr)(bb + (wave + jj * 4) * 256), 16, b_voff[jj], soff_b, 0, 0);
        // K is walked channel-chunk outer / filter-tap inner (see the flat variant)
        ++tap;
        if (++kw == p.KW) { kw = 0; if (++kh == p.KH) { kh = 0; tap = 0; ci0 += BK; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    // per-lane LDS read bases (floats): A chunk q of row (wm*64 + li) [+32 rows = +512 floats], B column
    const int row0 = wm * 64 + li;
    const float* a_rd0 = lds + row0 * BK + (((2 * lh + 0) ^ ((row0 >> 2) & 3)) << 2);
    const float* a_rd1 = lds + row0 * BK + (((2 * lh + 1) ^ ((row0 >> 2) & 3)) << 2);
    const float* b_rd = lds + AF + lh * 8 * BN + wn * 64 + li;

    auto compute = [&](auto curc) {
        constexpr int CUR = decltype(curc)::value;
        constexpr int BO = CUR * (AF + BF);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 av[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = *(const f32x4*)((q ? a_rd1 : a_rd0) + BO + a * 32 * BK);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bv[TN];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[b] = b_rd[BO + (q * 4 + e) * BN + b * 32];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][e], bv[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    const int nk_all = p.Ktot / BK;
    const int nk = SPLIT ? (ks_first + p.ksteps < nk_all ? p.ksteps : nk_all - ks_first) : nk_all;
    stage(lds);
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
        if (ks + 1 < nk) stage(lds + (AF + BF));
        compute(std::integral_constant<int, 0>{});
        __syncthreads();
        if (ks + 1 < nk) {
            if (ks + 2 < nk) stage(lds);
            compute(std::integral_constant<int, 1>{});
            __syncthreads();
        }
    }
    const int mw0 = m0 + wm * 64 + 4 * lh, nw0 = n0 + wn * 64 + li;
    if constexpr (SPLIT) {                                      // partial sums -> slab kz; the reduction kernel applies the epilogue
        const int mrel0 = mt0 * BM;                               // slab rows are relative to the launch's first tile
        float* slab = p.slab + (long long)kz * (p.M - mrel0) * p.Cout;
        auto put = [&](const f32x16& c, int mbase, int n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < p.M) slab[(long long)(m - mrel0) * p.Cout + n] = c[r];
            }
        };
        put(acc[0][0], mw0, nw0);
        put(acc[0][1], mw0, nw0 + 32);
        put(acc[1][0], mw0 + 32, nw0);
        put(acc[1][1], mw0 + 32, nw0 + 32);
        return;
    }
    if (p.fb_act >= 0) {
        // fused backward epilogue: column sums of this workgroup (2 row waves x 2 lane halves per column) meet in LDS,
        // then one atomic per channel and sum
        float s0[3] = {0.f, 0.f, 0.f}, s1[3] = {0.f, 0.f, 0.f};
        conv_epilogue_tile_bwd(p, acc[0][0], mw0, nw0, s0);
        conv_epilogue_tile_bwd(p, acc[1][0], mw0 + 32, nw0, s0);
        conv_epilogue_tile_bwd(p, acc[0][1], mw0, nw0 + 32, s1);
        conv_epilogue_tile_bwd(p, acc[1][1], mw0 + 32, nw0 + 32, s1);
#pragma unroll
        for (int k = 0; k < 3; ++k) { s0[k] += __shfl_xor(s0[k], 32, 64); s1[k] += __shfl_xor(s1[k], 32, 64); }
        __syncthreads();                                        // the K loop's LDS tiles are dead
        float* red = lds;                                       // [wm][wn*64 + col][3]
        if (lh == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                red[(wm * 128 + wn * 64 + li) * 3 + k] = s0[k];
                red[(wm * 128 + wn * 64 + 32 + li) * 3 + k] = s1[k];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int n = n0 + tid;
            const float db = red[tid * 3 + 0] + red[(128 + tid) * 3 + 0];
            const float dg = red[tid * 3 + 1] + red[(128 + tid) * 3 + 1];
            const float dbi = red[tid * 3 + 2] + red[(128 + tid) * 3 + 2];
            if (p.fb_dbeta) atomicAdd(p.fb_dbeta + n, db);
            if (p.fb_dgamma) atomicAdd(p.fb_dgamma + n, dg);
            if (p.fb_dbias) atomicAdd(p.fb_dbias + n, dbi);
        }
        return;
    }
    conv_epilogue_tile(p, acc[0][0], mw0, nw0);
    conv_epilogue_tile(p, acc[0][1], mw0, nw0 + 32);
    conv_epilogue_tile(p, acc[1][0], mw0 + 32, nw0);
    conv_epilogue_tile(p, acc[1][1], mw0 + 32, nw0 + 32);
}

__global__ __launch_bounds__(256, 2) void conv_fwd_blds_kernel(const ConvArgs p, const unsigned x_shift, const unsigned x_records) {
    conv_fwd_blds_body<false>(p, x_shift, x_records);
}

__global__ __launch_bounds__(256, 2) void conv_fwd_blds_splitk_kernel(const ConvArgs p, const unsigned x_shift, const unsigned x_records) {
    conv_fwd_blds_body<true>(p, x_shift, x_records);
}

// Second pass of split-K: sum the slabs in a fixed order, then the ordinary epilogue.
__device__ __forceinline__ void conv_splitk_epilogue_body(const ConvArgs& p, const unsigned block) {
    const long long i = (long long)block * 256 + threadIdx.x;
    if (i >= (long long)p.M * p.Cout) return;
    const int m = (int)(i / p.Cout), n = (int)(i - (long long)m * p.Cout);
    const float a = mrcnn_slab_sum<float>(0.f, p.slab, (long long)p.M * p.Cout, i, p.ksplit);
    const int c = n % p.cmod;
    float zv = a + (p.bias ? p.bias[c] : 0.f);
    const float sc = p.scale ? p.scale[c] : 1.f, sh = p.scale ? p.shift[c] : 0.f;
    const int ohw = p.OH * p.OW;
    long long addr, raddr;
    if (p.dense && p.res_mode != MRCNN_RES_UP2) {
        addr = i;
        raddr = i;
    } else {
        int ni = m / ohw, rem = m - ni * ohw;
        int oh = rem / p.OW, ow = rem - oh * p.OW;
        if (p.out_mode == MRCNN_OUT_DECONV2) {
            int ab = n / p.cmod;
            addr = (long long)ni * p.ons + (long long)(2 * oh + (ab >> 1)) * p.ohs + (long long)(2 * ow + (ab & 1)) * p.ows + c;
        } else {
            addr = (long long)ni * p.ons + (long long)oh * p.ohs + (long long)ow * p.ows + n;
        }
        raddr = addr;
        if (p.res_mode == MRCNN_RES_UP2)
            raddr = (((long long)ni * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1)) * p.Cout + n;
    }
    if (p.z) p.z[addr] = zv;
    float y = sc * zv + sh;
    if (p.res_mode != MRCNN_RES_NONE) y += p.res[raddr];
    if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
    else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
    p.out[addr] = y;
}

// Four channels per thread (Cout, cmod and the output strides multiples of 4, 16-byte aligned buffers, M*Cout < 2^31):
// no 64-bit division, float4 traffic.  Same arithmetic per element as conv_splitk_epilogue_kernel.
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvArgs p) { conv_splitk_epilogue_body(p, blockIdx.x); }

__device__ __forceinline__ void conv_splitk_epilogue_vec_body(const ConvArgs& p, const unsigned block) {
    const unsigned i4 = block * 256u + threadIdx.x;
    const unsigned c4 = (unsigned)p.Cout >> 2;
    if (i4 >= (unsigned)p.M * c4) return;
    const int m = (int)fast_div(i4, p.d_c4), n = (int)(i4 - (unsigned)m * c4) * 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 a = mrcnn_slab_sum<f32x4>(zero4, p.slab, (long long)p.M * p.Cout, (long long)i4 * 4, p.ksplit);
    int ab = 0, c = n;
    if (p.cmod != p.Cout) { ab = n / p.cmod; c = n - ab * p.cmod; }
    f32x4 bias = zero4, sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4;
    if (p.bias) bias = *(const f32x4*)(p.bias + c);
    if (p.scale) { sc = *(const f32x4*)(p.scale + c); sh = *(const f32x4*)(p.shift + c); }
    long long addr, raddr;
    if (p.dense && p.res_mode != MRCNN_RES_UP2) {
        addr = (long long)i4 * 4;
        raddr = addr;
    } else {
        const int ohw = p.OH * p.OW;
        const int ni = (int)fast_div((unsigned)m, p.d_ohw), rem = m - ni * ohw;
        const int oh = (int)fast_div((unsigned)rem, p.d_ow), ow = rem - oh * p.OW;
        if (p.out_mode == MRCNN_OUT_DECONV2)
            addr = (long long)ni * p.ons + (long long)(2 * oh + (ab >> 1)) * p.ohs + (long long)(2 * ow + (ab & 1)) * p.ows + c;
        else
            addr = (long long)ni * p.ons + (long long)oh * p.ohs + (long long)ow * p.ows + n;
        raddr = addr;
        if (p.res_mode == MRCNN_RES_UP2)
            raddr = (((long long)ni * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1)) * p.Cout + n;
    }
    f32x4 zv, y;
#pragma unroll
    for (int q = 0; q < 4; ++q) zv[q] = a[q] + bias[q];
    if (p.z) *(f32x4*)(p.z + addr) = zv;
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = sc[q] * zv[q] + sh[q];
    if (p.res_mode != MRCNN_RES_NONE) {
        const f32x4 r = *(const f32x4*)(p.res + raddr);
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] += r[q];
    }
    if (p.act == MRCNN_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = fmaxf(y[q], 0.f);
    } else if (p.act == MRCNN_ACT_SIGMOID) {
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = 1.f / (1.f + expf(-y[q]));
    }
    *(f32x4*)(p.out + addr) = y;
}

__global__ __launch_bounds__(256) void conv_splitk_epilogue_vec_kernel(const ConvArgs p) { conv_splitk_epilogue_vec_body(p, blockIdx.x); }

__global__ __launch_bounds__(256) void conv_splitk_epilogue_multi_kernel(const ConvMultiArgs mp) {
    int g = 0;
    while (g + 1 < mp.n && (int)blockIdx.x >= mp.efirst[g + 1]) ++g;
    if (mp.evec[g]) conv_splitk_epilogue_vec_body(mp.a[g], blockIdx.x - (unsigned)mp.efirst[g]);
    else conv_splitk_epilogue_body(mp.a[g], blockIdx.x - (unsigned)mp.efirst[g]);
}

static bool splitk_epilogue_vec_ok(const ConvArgs& a) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((a.Cout & 3) || (a.cmod & 3) || (long long)a.M * a.Cout >= (1ll << 31)) return false;
    if (!a.dense && ((a.ons | a.ohs | a.ows) & 3)) return false;
    return al(a.slab) && al(a.out) && al(a.z) && al(a.res) && al(a.bias) && al(a.scale) && al(a.shift);
}

// Split-K second pass fused with the epilogue backward of the layer below (mrcnn_conv2d_dgrad_ep on small feature
// maps): y = sum of the slabs in slice order (+ res), then exactly what epilogue_bwd_vec_kernel does with y -- one
// launch and one round trip of y less per layer.  C = 4 * 2^k >= 16; a thread owns one float4 channel group and walks
// rows; channel sums stay in registers until one LDS + one global atomic per channel per workgroup.
#define SPLITK_EPI_U 2
__global__ __launch_bounds__(256) void conv_splitk_epilogue_bwd_kernel(const ConvArgs p, const long long rows_per_block, const int lg) {
    __shared__ float sacc[3 * 4 * 256];   // [3][4L]
    const int C = p.Cout;
    const int L = 1 << lg, R = 256 >> lg;
    for (int c = threadIdx.x; c < 12 * L; c += 256) sacc[c] = 0.f;
    __syncthreads();
    const int rsub = threadIdx.x >> lg, lane = threadIdx.x & (L - 1);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > p.M) r1 = p.M;
    const long long slab_stride = (long long)p.M * C;
    const int c = (blockIdx.y * L + lane) * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
    if (p.fb_scale) sc = *(const f32x4*)(p.fb_scale + c);
    if (p.fb_dgamma) { mu = *(const f32x4*)(p.fb_mean + c); rs = *(const f32x4*)(p.fb_rstd + c); }
    f32x4 a_db = {0.f, 0.f, 0.f, 0.f}, a_dg = a_db, a_bias = a_db;
    // rows in batches of SPLITK_EPI_U: all loads of a batch before its first store
    for (long long rb = r0 + rsub; rb < r1; rb += (long long)R * SPLITK_EPI_U) {
        f32x4 gg[SPLITK_EPI_U], oo[SPLITK_EPI_U], zz[SPLITK_EPI_U];
#pragma unroll
        for (int u = 0; u < SPLITK_EPI_U; ++u) {
            long long r = rb + (long long)u * R;
            if (r >= r1) r = r1 - 1;                       // clamped: in range, result discarded below
            const long long e = r * C + c;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            f32x4 g = mrcnn_slab_sum<f32x4>(zero4, p.slab, slab_stride, e, p.ksplit);
            if (p.res_mode != MRCNN_RES_NONE) {
                const f32x4 v = *(const f32x4*)(p.res + e);
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] += v[q];
            }
            gg[u] = g;
            if (p.fb_act == MRCNN_ACT_RELU) oo[u] = *(const f32x4*)(p.fb_out + e);
            if (p.fb_dgamma) zz[u] = *(const f32x4*)(p.fb_z + e);
        }
#pragma unroll
        for (int u = 0; u < SPLITK_EPI_U; ++u) {
            const long long r = rb + (long long)u * R;
            if (r >= r1) break;
            const long long e = r * C + c;
            f32x4 g = gg[u];
            if (p.fb_act == MRCNN_ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = oo[u][q] > 0.f ? g[q] : 0.f;
            }
            f32x4 dz;
#pragma unroll
            for (int q = 0; q < 4; ++q) dz[q] = g[q] * sc[q];
            *(f32x4*)(p.out + e) = dz;
            if (p.fb_dy) *(f32x4*)(p.fb_dy + e) = g;
            if (p.fb_dgamma) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a_dg[q] += g[q] * (zz[u][q] - mu[q]) * rs[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { a_db[q] += g[q]; a_bias[q] += dz[q]; }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (p.fb_dbeta || p.fb_dgamma) atomicAdd(&sacc[lane * 4 + q], a_db[q]);
        if (p.fb_dgamma) atomicAdd(&sacc[4 * L + lane * 4 + q], a_dg[q]);
        if (p.fb_dbias) atomicAdd(&sacc[8 * L + lane * 4 + q], a_bias[q]);
    }
    __syncthreads();
    const int cb = blockIdx.y * 4 * L;
    for (int j = threadIdx.x; j < 4 * L; j += 256) {
        if (p.fb_dbeta) atomicAdd(&p.fb_dbeta[cb + j], sacc[j]);
        if (p.fb_dgamma) atomicAdd(&p.fb_dgamma[cb + j], sacc[4 * L + j]);
        if (p.fb_dbias) atomicAdd(&p.fb_dbias[cb + j], sacc[8 * L + j]);
    }
}


// ---------------------------------------------------------------------------------------------------
// Single-launch kernel for the small layers of the trunk (the feature maps of C3..C5 / P3..P6: a few thousand pixels).
// Those layers used to be cut into K slices run by separate workgroups, with a second launch summing the slabs: two
// launches, a float32 slab round trip through HBM, and per workgroup a dependent chain of 64-cycle fp32 MFMAs.  Here a
// workgroup owns one 32 x 32 output tile and its FOUR WAVES take one quarter of K each -- 1 024 wave-sized tasks for
// M = 1 024, N = 256, one per SIMD of the chip -- every wave streaming its own operands global -> LDS by DMA (two
// private 8 KiB stages, the next K-step in flight behind a counted vmcnt: no barrier inside the K loop, the waves never
// wait for one another; 64 KiB per workgroup, two workgroups per CU).  The four partial tiles meet in LDS, are summed in wave order (fixed order: bitwise reproducible) and
// every thread finishes four adjacent columns of one row: bias, frozen-BN affine, residual, activation, float4 stores --
// or, for a data gradient, the backward epilogue of the layer below (conv_splitk_epilogue_bwd_kernel's arithmetic).
// LDS rows are 128 bytes (32 floats) for both operands; A rows use the chunk swizzle c ^ ((r >> 1) & 7) on the DMA's
// source side and in the ds_read_b128 operand reads; B rows ([k][32 columns]) are read with ds_read_b32 (lane = column).
#define SK_STAGE_BYTES 8192
#define SK_NSTAGE 2
__global__ __launch_bounds__(256, 2) void conv_fwd_sk_kernel(const ConvArgs p, const unsigned x_shift, const unsigned x_records) {
    __shared__ __attribute__((aligned(16))) char lds[4 * SK_NSTAGE * SK_STAGE_BYTES];     // 64 KiB: 4 waves x 2 stages x (A 4 KiB + B 4 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.Cout >> 5;
    const int mtile = blockIdx.x / ntiles, ntile = blockIdx.x - mtile * ntiles;
    const int m0 = mtile * 32, n0 = ntile * 32;
    const int ohw = p.OH * p.OW;
    char* my = lds + wave * (SK_NSTAGE * SK_STAGE_BYTES);

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - x_shift), 0, x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + n0), 0, (unsigned)(((long long)p.Ktot * p.Cout - n0) * 4), 0x00020000);

    // A: 4 pieces of 8 rows x 128 B; lane -> row 8 j + (lane >> 3), physical chunk lane & 7
    unsigned a_voff[4];
    unsigned long long a_mask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = j * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / ohw, rem = mm - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        a_voff[j] = (unsigned)((((long long)n * p.H * p.W + (long long)ih0 * p.W + iw0) * p.Cin + cl * 4) * 4 + x_shift);
        unsigned long long mk = 0ull;
        if (ok)
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int th = t / p.KW, tw = t - th * p.KW;
                if ((unsigned)(ih0 + th) < (unsigned)p.H && (unsigned)(iw0 + tw) < (unsigned)p.W) mk |= 1ull << t;
            }
        a_mask[j] = mk;
    }
    // B: 4 pieces of 8 k-rows x 128 B (32 columns): lane -> k-row 8 j + (lane >> 3), chunk lane & 7 (no swizzle: b32 reads)
    const unsigned b_voff = (unsigned)(((lane >> 3) * p.Cout + (lane & 7) * 4) * 4);

    // this wave's K-steps (of 32): [ks0, ks1)
    const int nk = p.Ktot >> 5;
    const int ks0 = (nk * wave) >> 2, ks1 = (nk * (wave + 1)) >> 2;
    const int cpt = p.Cin >> 5;                                  // K-steps per filter tap
    int tap = ks0 / cpt, cc = ks0 - tap * cpt;
    int kh = tap / p.KW, kw = tap - kh * p.KW;
    auto stage = [&](char* buf, int ks) {
        const unsigned soff_a = (unsigned)(((kh * p.W + kw) * p.Cin + cc * 32) * 4);
        const unsigned long long bit = 1ull << tap;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned vo = (a_mask[j] & bit) ? a_voff[j] : CONV_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (conv_lds_ptr)(buf + j * 1024), 16, vo, soff_a, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (conv_lds_ptr)(buf + 4096 + j * 1024), 16, b_voff,
                                                     (unsigned)(((ks * 32 + j * 8) * p.Cout) * 4), 0, 0);
        if (++cc == cpt) { cc = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    int a_rd[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) a_rd[t4] = li * 128 + (((lh * 4 + t4) ^ ((li >> 1) & 7)) << 4);
    const int b_rd = 4096 + lh * 16 * 128 + li * 4;
    auto compute = [&](const char* buf) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const f32x4 av = *(const f32x4*)(buf + a_rd[t4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bv = *(const float*)(buf + b_rd + (t4 * 4 + e) * 128);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv, acc, 0, 0, 0);
            }
        }
    };
    const int n_my = ks1 - ks0;
    if (n_my > 0) stage(my, ks0);
    int slot = 0;
    for (int i = 0; i < n_my; ++i) {
        if (i + 1 < n_my) {
            stage(my + (slot ^ 1) * SK_STAGE_BYTES, ks0 + i + 1); // its previous content (step i - 1) has been consumed
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // 8 DMA instructions per stage: step i + 1 stays in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        compute(my + slot * SK_STAGE_BYTES);
        slot ^= 1;
    }

    // ---- the four partial tiles meet in LDS: [wave][row 32][col 32] ---------------------------------------------
    __syncthreads();                                             // every wave is done with its stages
    float* part = (float*)lds;
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave * 1024 + (4 * lh + (r & 3) + 8 * (r >> 2)) * 32 + li] = acc[r];
    __syncthreads();
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    f32x4 v = *(const f32x4*)&part[row * 32 + c4];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const f32x4 q = *(const f32x4*)&part[w * 1024 + row * 32 + c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += q[e];
    }
    const int m = m0 + row, n = n0 + c4;
    const long long addr = (long long)m * p.Cout + n;
    if (p.fb_act < 0) {
        if (m >= p.M) return;
        f32x4 zv, y;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
        const f32x4 bias = p.bias ? *(const f32x4*)(p.bias + n) : zero4;
        const f32x4 sc = p.scale ? *(const f32x4*)(p.scale + n) : one4, sh = p.scale ? *(const f32x4*)(p.shift + n) : zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) zv[e] = v[e] + bias[e];
        if (p.z) *(f32x4*)(p.z + addr) = zv;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = sc[e] * zv[e] + sh[e];
        if (p.res_mode != MRCNN_RES_NONE) {
            const f32x4 r4 = *(const f32x4*)(p.res + addr);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] += r4[e];
        }
        if (p.act == MRCNN_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
        } else if (p.act == MRCNN_ACT_SIGMOID) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = 1.f / (1.f + expf(-y[e]));
        }
        *(f32x4*)(p.out + addr) = y;
        return;
    }
    // ---- data gradient: y = v (+ res) is d(loss)/d(out_below); apply that layer's epilogue backward ---------------
    float* sums = part + 4 * 1024;                               // [3][32] column sums of this workgroup (behind the partial tiles)
    if (tid < 96) sums[tid] = 0.f;
    __syncthreads();
    if (m < p.M) {
        f32x4 g = v;
        if (p.res_mode != MRCNN_RES_NONE) {
            const f32x4 r4 = *(const f32x4*)(p.res + addr);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += r4[e];
        }
        if (p.fb_act == MRCNN_ACT_RELU) {
            const f32x4 o4 = *(const f32x4*)(p.fb_out + addr);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = o4[e] > 0.f ? g[e] : 0.f;
        }
        const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
        const f32x4 sc = p.fb_scale ? *(const f32x4*)(p.fb_scale + n) : one4;
        f32x4 dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = g[e] * sc[e];
        *(f32x4*)(p.out + addr) = dz;
        if (p.fb_dy) *(f32x4*)(p.fb_dy + addr) = g;
        if (p.fb_dgamma) {
            const f32x4 z4 = *(const f32x4*)(p.fb_z + addr), mu = *(const f32x4*)(p.fb_mean + n), rs = *(const f32x4*)(p.fb_rstd + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&sums[32 + c4 + e], g[e] * (z4[e] - mu[e]) * rs[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { atomicAdd(&sums[c4 + e], g[e]); atomicAdd(&sums[64 + c4 + e], dz[e]); }
    }
    __syncthreads();
    if (tid < 32) {
        if (p.fb_dbeta) atomicAdd(p.fb_dbeta + n0 + tid, sums[tid]);
        if (p.fb_dgamma) atomicAdd(p.fb_dgamma + n0 + tid, sums[32 + tid]);
        if (p.fb_dbias) atomicAdd(p.fb_dbias + n0 + tid, sums[64 + tid]);
    }
}

// ---- 16 x 16 tiles for the layers too small even for conv_fwd_sk_kernel (round 3) ---------------------------------------------------
// At batch-1 detect a res4 layer is 256 pixels x 256 channels = 64 tiles of 32 x 32: the single-launch kernel above leaves three
// quarters of the SIMDs idle there and the planner fell back to K slices over workgroups + a reduction launch (8.3 + 4.9 us per
// layer, 220 launches for the trunk).  With 16 x 16 tiles (v_mfma_f32_16x16x4_f32) the same layer is 256 workgroups of four waves,
// one launch, no slab: a wave takes a quarter of K, streams 16 rows x 32 k of A and 32 k x 16 columns of B per step (2 + 2 KiB,
// two private stages, counted vmcnt, no barrier in the K loop), the four partial tiles are summed in LDS in wave order.
// Operand mapping of the instruction: lane (i = l % 16, g = l / 16) supplies A[i][4 t + g] and B[4 t + g][j = l % 16] for k-block t,
// and holds D[4 g + r][l % 16], r = 0..3.  LDS: A rows of 128 bytes with the chunk swizzle c ^ ((row >> 1) & 7) (source side of
// the DMA and in the reads), B rows of 64 bytes ([k][16 columns]: the four lane groups read four consecutive rows, 256 bytes).
// Forward epilogue only (bias, frozen-BN affine, residual, activation, optional pre-BN output).
#define SK16_STAGE_BYTES 4096
__global__ __launch_bounds__(256, 4) void conv_fwd_sk16_kernel(const ConvArgs p, const unsigned x_shift, const unsigned x_records) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 2 * SK16_STAGE_BYTES];     // 32 KiB: 4 waves x 2 stages x (A 2 KiB + B 2 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.Cout >> 4;
    const int mtile = blockIdx.x / ntiles, ntile = blockIdx.x - mtile * ntiles;
    const int m0 = mtile * 16, n0 = ntile * 16;
    const int ohw = p.OH * p.OW;
    char* my = lds + wave * (2 * SK16_STAGE_BYTES);

    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - x_shift), 0, x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + n0), 0, (unsigned)(((long long)p.Ktot * p.Cout - n0) * 4), 0x00020000);

    // A: 2 pieces of 8 rows x 128 B; lane -> row 8 j + (lane >> 3), physical chunk lane & 7
    unsigned a_voff[2];
    unsigned long long a_mask[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j * 8 + (lane >> 3);
        const int cl = (lane & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / ohw, rem = mm - n * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad_t, iw0 = ow * p.stride - p.pad_l;
        a_voff[j] = (unsigned)((((long long)n * p.H * p.W + (long long)ih0 * p.W + iw0) * p.Cin + cl * 4) * 4 + x_shift);
        unsigned long long mk = 0ull;
        if (ok)
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int th = t / p.KW, tw = t - th * p.KW;
                if ((unsigned)(ih0 + th) < (unsigned)p.H && (unsigned)(iw0 + tw) < (unsigned)p.W) mk |= 1ull << t;
            }
        a_mask[j] = mk;
    }
    // B: 2 pieces of 16 k-rows x 64 B: lane -> k-row 16 j + (lane >> 2), 16-byte chunk lane & 3
    const unsigned b_voff = (unsigned)(((lane >> 2) * p.Cout + (lane & 3) * 4) * 4);

    const int nk = p.Ktot >> 5;                                  // K-steps of 32
    const int ks0 = (nk * wave) >> 2, ks1 = (nk * (wave + 1)) >> 2;
    const int cpt = p.Cin >> 5;
    int tap = ks0 / cpt, cc = ks0 - tap * cpt;
    int kh = tap / p.KW, kw = tap - kh * p.KW;
    auto stage = [&](char* buf, int ks) {
        const unsigned soff_a = (unsigned)(((kh * p.W + kw) * p.Cin + cc * 32) * 4);
        const unsigned long long bit = 1ull << tap;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = (a_mask[j] & bit) ? a_voff[j] : CONV_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (conv_lds_ptr)(buf + j * 1024), 16, vo, soff_a, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (conv_lds_ptr)(buf + 2048 + j * 1024), 16, b_voff,
                                                     (unsigned)(((ks * 32 + j * 16) * p.Cout) * 4), 0, 0);
        if (++cc == cpt) { cc = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }
    };

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lg = lane >> 4;
    const int a_row = li * 128 + lg * 4, a_sw = (li >> 1) & 7;
    const int b_rd = 2048 + lg * 64 + li * 4;
    auto compute = [&](const char* buf) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float av = *(const float*)(buf + a_row + ((t ^ a_sw) << 4));
            const float bv = *(const float*)(buf + b_rd + t * 256);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
    };
    const int n_my = ks1 - ks0;
    if (n_my > 0) stage(my, ks0);
    int slot = 0;
    for (int i = 0; i < n_my; ++i) {
        if (i + 1 < n_my) {
            stage(my + (slot ^ 1) * SK16_STAGE_BYTES, ks0 + i + 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // 4 DMA instructions per stage: step i + 1 stays in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        compute(my + slot * SK16_STAGE_BYTES);
        slot ^= 1;
    }

    // ---- the four partial tiles meet in LDS: [wave][row 16][col 16] ---------------------------------------------------
    __syncthreads();
    float* part = (float*)lds;
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave * 256 + (4 * lg + r) * 16 + li] = acc[r];
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    float v = part[row * 16 + col];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += part[w * 256 + row * 16 + col];
    const int m = m0 + row, n = n0 + col;
    if (m >= p.M) return;
    const long long addr = (long long)m * p.Cout + n;
    const float zv = v + (p.bias ? p.bias[n] : 0.f);
    if (p.z) p.z[addr] = zv;
    float y = (p.scale ? p.scale[n] : 1.f) * zv + (p.scale ? p.shift[n] : 0.f);
    if (p.res_mode != MRCNN_RES_NONE) y += p.res[addr];
    if (p.act == MRCNN_ACT_RELU) y = fmaxf(y, 0.f);
    else if (p.act == MRCNN_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
    p.out[addr] = y;
}

static long long sk_max_tiles() { static const long long v = getenv("MRCNN_SK_MAX_TILES") ? atoll(getenv("MRCNN_SK_MAX_TILES")) : 512; return v; }
// applicability window, measured (ResNet-101, layer loops and whole steps): 192 .. 512 tiles of 32 x 32 and >= 16 K-steps.
// Fewer tiles (batch-1 detect: 64) leave three quarters of the SIMDs without a wave -- the old split over 16 K slices
// spreads wider; more tiles than two per CU run in several rounds (M = 4 096: 39 vs 33 us); short K has nothing to split.
// In the window: res4 2a 17.3 -> 10.6 us, 2b 26.0 -> 22.2 us per layer; sparse step 29.1 -> 28.5 ms.
static long long sk_min_tiles() { static const long long v = getenv("MRCNN_SK_MIN_TILES") ? atoll(getenv("MRCNN_SK_MIN_TILES")) : 192; return v; }
static int sk_min_steps() { static const int v = getenv("MRCNN_SK_MIN_STEPS") ? atoi(getenv("MRCNN_SK_MIN_STEPS")) : 16; return v; }

// Shapes the single-launch kernel takes: a "fast" layer (Cin % 32 == 0), Cout % 32 == 0, dense NHWC output and residual,
// 16-byte aligned operands, at most 64 taps, buffer-addressable tensors -- and small enough that the large-tile kernels
// would not fill the chip (the caller decides that part).
static bool conv_sk_ok(const mrcnn_conv_desc* d, const ConvArgs& a) {
    static const bool on = !(getenv("MRCNN_SK_KERNEL") && getenv("MRCNN_SK_KERNEL")[0] == '0');
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!on || !a.fastA || !a.dense || d->Cout % 32 || d->out_mode != MRCNN_OUT_NHWC || d->res_mode == MRCNN_RES_UP2) return false;
    if (d->KH * d->KW > 64) return false;
    const long long xb = (long long)d->N * d->H * d->W * d->Cin * 4 + ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
    if (xb >= 0x7FFFFFF0LL || (long long)a.Ktot * d->Cout * 4 >= 0x7FFFFFF0LL) return false;
    return al(a.w) && al(a.out) && al(a.z) && al(a.res) && al(a.bias) && al(a.scale) && al(a.shift) && al(a.fb_out) && al(a.fb_z) &&
           al(a.fb_scale) && al(a.fb_mean) && al(a.fb_rstd) && al(a.fb_dy);
}

static void launch_splitk_reduction(const ConvArgs& a, hipStream_t s);

template <int BM, int BN, int WM, int WN>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
    const int mt = (a.M + BM - 1) / BM, nt = (a.Cout + BN - 1) / BN;
    if (a.fastA)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, true>), dim3((unsigned)(mt * nt * a.ksplit)), dim3(WM * WN * 64), 0, s, a);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, false>), dim3((unsigned)(mt * nt * a.ksplit)), dim3(WM * WN * 64), 0, s, a);
    launch_splitk_reduction(a, s);
    return mrcnn_launch_status();
}

// second pass of a split-K launch: slab sum + epilogue (or the backward epilogue of the layer below)
static void launch_splitk_reduction(const ConvArgs& a, hipStream_t s) {
    if (a.ksplit > 1 && a.fb_act >= 0) {         // slabs -> epilogue backward of the layer below
        const EpiGrid g = mrcnn_epilogue_grid(a.M, a.Cout);
        hipLaunchKernelGGL(conv_splitk_epilogue_bwd_kernel, dim3(g.row_blocks, g.chan_blocks), dim3(256), 0, s, a,
                           g.rows_per_block, g.lg);
    } else if (a.ksplit > 1) {
        const long long n = (long long)a.M * a.Cout;
        if (splitk_epilogue_vec_ok(a))
            hipLaunchKernelGGL(conv_splitk_epilogue_vec_kernel, dim3((unsigned)cdiv64(n / 4, 256)), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, a);
    }
}

// 128-row tiles only when they give every CU several workgroups (256 CUs x 5 resident): fewer, and the chip idles
#ifndef CONV_BIG_TILE_MIN_BLOCKS
#define CONV_BIG_TILE_MIN_BLOCKS 640
#endif
struct ConvPlan { int bm, bn, ksplit, ksteps; bool dma_split; };

// Tile + split-K choice.  Large problems: 128-row tiles, no split.  Small feature maps (C3..C5, P3..P6
// at 256^2 inputs) have only 4..128 output tiles, far fewer than the 256 CUs, so the K loop is cut into
// up to 16 slices that run as separate workgroups (slabs summed by conv_splitk_epilogue_kernel).
static ConvPlan plan_conv(const mrcnn_conv_desc* d, bool allow_dma_split = true) {
    ConvPlan pl;
    const long long M = (long long)d->N * d->OH * d->OW;
    const int Cout = d->Cout;
    const int nk = (d->KH * d->KW * d->Cin + 31) / 32;
    long long blocks;
    if (Cout <= 32) {
        pl.bn = 32;
        pl.bm = ((M + 127) / 128 >= 256) ? 128 : 64;
    } else if (Cout <= 64) {
        pl.bn = 64;
        pl.bm = ((M + 127) / 128 >= 256) ? 128 : 64;
    } else {
        pl.bn = 128;
        pl.bm = (((M + 127) / 128) * ((Cout + 127) / 128) >= CONV_BIG_TILE_MIN_BLOCKS) ? 128 : 64;
        if (pl.bm == 64 && ((M + 63) / 64) * ((Cout + 127) / 128) < 128) pl.bn = 64;   // more, smaller tiles
    }
    pl.dma_split = false;
    // mid-size layers (too few 128x128 tiles to fill the chip, enough K to cut): LDS-DMA kernel with split-K --
    // each slice at least 16 K-steps of 16; MRCNN_DMA_SPLIT_MIN_TILES=0 switches it off (A/B)
    static const long long dma_min_tiles = getenv("MRCNN_DMA_SPLIT_MIN_TILES") ? atoll(getenv("MRCNN_DMA_SPLIT_MIN_TILES")) : 96;
    const long long t128 = ((M + 127) / 128) * (Cout / 128);
    const int nk16 = d->KH * d->KW * d->Cin / 16;
    if (allow_dma_split && dma_min_tiles > 0 && pl.bm == 64 && Cout % 128 == 0 && d->Cin % 32 == 0 && d->KH * d->KW <= 64 &&
        t128 >= dma_min_tiles && t128 < CONV_BIG_TILE_MIN_BLOCKS && nk16 >= 32) {
        static const long long dma_target = getenv("MRCNN_DMA_SPLIT_TARGET") ? atoll(getenv("MRCNN_DMA_SPLIT_TARGET")) : 768;
        long long ks = (dma_target + t128 - 1) / t128;
        if (ks > nk16 / 16) ks = nk16 / 16;
        if (ks > 16) ks = 16;
        if (ks >= 2) {
            pl.bm = 128; pl.bn = 128; pl.dma_split = true;
            pl.ksteps = (int)((nk16 + ks - 1) / ks);
            pl.ksteps += pl.ksteps & 1;                         // the loop is unrolled over the two LDS buffers
            pl.ksplit = (nk16 + pl.ksteps - 1) / pl.ksteps;
            return pl;
        }
    }
    blocks = ((M + pl.bm - 1) / pl.bm) * ((Cout + pl.bn - 1) / pl.bn);
    pl.ksplit = 1;
    pl.ksteps = nk;
    static const long long split_below = getenv("MRCNN_SPLITK_BELOW") ? atoll(getenv("MRCNN_SPLITK_BELOW")) : 768;
    if (blocks < split_below && nk >= 8) {
        static const long long target = getenv("MRCNN_SPLITK_TARGET") ? atoll(getenv("MRCNN_SPLITK_TARGET")) : 512;
        long long want = (target + blocks - 1) / blocks;         // ~2 workgroups per CU: more slices cost more slab traffic than they hide latency
        long long maxs = nk / 4;                               // at least 4 K-steps per slice
        long long ks = want < maxs ? want : maxs;
        if (ks > 16) ks = 16;
        if (ks >= 2) {
            pl.ksteps = (int)((nk + ks - 1) / ks);
            pl.ksplit = (nk + pl.ksteps - 1) / pl.ksteps;
        }
    }
    return pl;
}

// Tail of a large LDS-DMA layer.  The unsplit kernel keeps 5 workgroups per CU resident; a grid of r.f rounds costs
// ceil(r.f) rounds (M = 200 704, Cout = 256: 3136 tiles on 1280 slots = 2.45 rounds -> 3, 76 % of the 128-row kernel's rate).
// When the last round would be at most 60 % full, the whole rounds go to the unsplit kernel and the remaining tiles to the
// split-K variant with K cut so that they fill one round of shorter workgroups; their slabs (rows relative to the first
// tail tile) are reduced by the usual second launch on the tail rows.
struct ConvTailPlan { int ok, mt_full, ksplit, ksteps; size_t slab_bytes; };

static ConvTailPlan plan_conv_tail(const mrcnn_conv_desc* d) {
    ConvTailPlan t = {0, 0, 0, 0, 0};
    // OFF by default (MRCNN_CONV_TAIL_SPLIT=1 turns it on, read per call).  Measured on M = 200 704, Cout = 256, K = 2304:
    // alone 1.88 -> 1.82 ms (79.9 -> 82.9 % of the matrix peak) for the forward, nothing for the data gradient with the fused
    // backward epilogue; INSIDE the ResNet-50 step 30.99 -> 31.35 ms -- there the weight-gradient stream already fills the
    // slots of the partial round, and the two extra launches only cost.
    const char* env = getenv("MRCNN_CONV_TAIL_SPLIT");
    const int enabled = env ? atoi(env) : 0;
    const long long M = (long long)d->N * d->OH * d->OW;
    if (!enabled || d->Cout % 128 || d->Cin % 16) return t;
    const long long mt = (M + 127) / 128, nt = d->Cout / 128, tiles = mt * nt;
    const long long slots = 5LL * mrcnn_num_cus();
    const long long full = tiles / slots * slots / nt * nt, tail = tiles - full;
    const int nk = d->KH * d->KW * d->Cin / 16;
    if (full <= 0 || tail <= 0 || tail * 10 > slots * 6 || nk < 32) return t;
    // K slices: the split variant keeps 4 workgroups per CU (123 VGPRs); tail * ks workgroups of 1 / ks length cost
    // ceil(tail * ks / slots4) / ks rounds, every slice adds one slab of the tail rows to write and read back (~4.5 TB/s,
    // against ~4.4 us per K-step of a workgroup at full occupancy).  Slices start on a channel-chunk boundary (tap 0).
    const long long slots4 = 4LL * mrcnn_num_cus();
    const int taps = d->KH * d->KW;
    const double slab_cost = (double)(M - full / nt * 128) * d->Cout * 8.0 / 4.5e12 / (nk * 4.4e-6);
    double best = 1.0;                                          // the unsplit tail: one more round of the unsplit kernel
    long long ks = 1, steps = nk;
    for (long long k = 2; k <= 8; ++k) {
        long long st = (nk + k - 1) / k;
        st = (st + taps - 1) / taps * taps;
        const long long kk = (nk + st - 1) / st;
        if (kk < 2) continue;
        const double cost = (double)((tail * kk + slots4 - 1) / slots4) * st / nk + slab_cost * kk;
        if (cost < best - 0.05) { best = cost; ks = kk; steps = st; }
    }
    if (ks < 2) return t;
    t.ok = 1; t.mt_full = (int)(full / nt); t.ksplit = (int)ks; t.ksteps = (int)steps;
    t.slab_bytes = (size_t)ks * (size_t)(M - (long long)t.mt_full * 128) * d->Cout * sizeof(float);
    return t;
}

extern "C" size_t mrcnn_conv2d_fwd_workspace(const mrcnn_conv_desc* d) {
    if (!d || d->N <= 0 || d->OH <= 0 || d->OW <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->Cin <= 0) return 0;
    ConvPlan pl = plan_conv(d);
    if (pl.ksplit <= 1) {
        if (pl.bm == 128 && pl.bn == 128) return plan_conv_tail(d).slab_bytes;     // 0 unless the tail split applies
        return 0;
    }
    return (size_t)pl.ksplit * d->N * d->OH * d->OW * d->Cout * sizeof(float);
}

// argument checks + everything of ConvArgs that does not depend on the launch plan
static int conv_fill_args(const mrcnn_conv_desc* d, const float* x, const float* w, const float* bias, const float* scale,
                          const float* shift, const float* res, float* out, float* z_out, ConvArgs& a) {
    if (!d || !x || !w || !out) return MRCNN_ERR_ARG;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 ||
        d->stride <= 0 || d->OH <= 0 || d->OW <= 0 || d->cmod <= 0)
        return MRCNN_ERR_ARG;
    if (d->res_mode != MRCNN_RES_NONE && !res) return MRCNN_ERR_ARG;
    if (scale && !shift) return MRCNN_ERR_ARG;
    if (d->out_mode == MRCNN_OUT_DECONV2 && (d->Cout != 4 * d->cmod || d->res_mode != MRCNN_RES_NONE))
        return MRCNN_ERR_ARG;
    if (d->res_mode == MRCNN_RES_UP2 && ((d->OH & 1) || (d->OW & 1))) return MRCNN_ERR_ARG;
    // the last input row/col touched must exist for at least one tap (host-side shape sanity)
    if ((long long)(d->OH - 1) * d->stride - d->pad_t >= d->H || (long long)(d->OW - 1) * d->stride - d->pad_l >= d->W)
        return MRCNN_ERR_ARG;
    long long M = (long long)d->N * d->OH * d->OW;
    if (M >= (1LL << 31) || (long long)d->KH * d->KW * d->Cin >= (1LL << 31)) return MRCNN_ERR_ARG;

    a.x = x; a.w = w; a.bias = bias; a.scale = scale; a.shift = shift; a.res = res; a.out = out; a.z = z_out;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.OH = d->OH; a.OW = d->OW;
    a.act = d->act; a.res_mode = d->res_mode; a.out_mode = d->out_mode; a.cmod = d->cmod;
    a.ons = d->out_n_stride; a.ohs = d->out_h_stride; a.ows = d->out_w_stride;
    a.M = (int)M; a.Ktot = d->KH * d->KW * d->Cin; a.nk = (a.Ktot + 31) / 32;
    a.d_ohw = make_fastdiv((unsigned)(d->OH * d->OW)); a.d_ow = make_fastdiv((unsigned)d->OW);
    a.d_c4 = make_fastdiv((unsigned)(d->Cout >= 4 ? d->Cout / 4 : 1));
    a.fastA = (d->Cin % 32 == 0) && (d->KH * d->KW <= 64) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    a.vecB = (d->Cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    a.dense = d->out_mode == MRCNN_OUT_NHWC && d->out_w_stride == d->Cout &&
              d->out_h_stride == (int64_t)d->OW * d->Cout && d->out_n_stride == (int64_t)d->OH * d->OW * d->Cout;
    a.ksplit = 1; a.ksteps = a.nk; a.slab = nullptr; a.mtile0 = 0; a.bat_rows = 0;
    a.fb_act = -1;
    a.fb_out = a.fb_z = a.fb_scale = a.fb_mean = a.fb_rstd = nullptr;
    a.fb_dgamma = a.fb_dbeta = a.fb_dbias = nullptr;
    a.fb_dy = nullptr;
    return MRCNN_OK;
}

static int conv_fwd_impl(const mrcnn_conv_desc* d, const float* x, const float* w,
                         const float* bias, const float* scale, const float* shift,
                         const float* res, float* out, float* z_out, void* workspace,
                         size_t workspace_bytes, const mrcnn_bwd_epilogue* ep, void* stream) {
    ConvArgs a;
    const int rc = conv_fill_args(d, x, w, bias, scale, shift, res, out, z_out, a);
    if (rc != MRCNN_OK) return rc;
    const long long M = a.M;
    hipStream_t s = (hipStream_t)stream;

    const long long xbytes_all = (long long)d->N * d->H * d->W * d->Cin * 4 + ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
    const bool buf_ok = a.fastA && a.vecB && xbytes_all < 0x7FFFFFF0LL && (long long)a.Ktot * d->Cout * 4 < 0x7FFFFFF0LL &&
                        !mrcnn_force_flat_glds();
    ConvPlan pl = plan_conv(d, buf_ok);
    size_t need = pl.ksplit > 1 ? (size_t)pl.ksplit * M * d->Cout * sizeof(float) : 0;
    if (pl.dma_split && (!workspace || workspace_bytes < need)) {  // sized for another plan: fall back to the 64-row tiles
        pl = plan_conv(d, false);
        need = pl.ksplit > 1 ? (size_t)pl.ksplit * M * d->Cout * sizeof(float) : 0;
    }
    if (need && (!workspace || workspace_bytes < need)) {        // no room: run unsplit
        pl.ksplit = 1;
        pl.ksteps = a.nk;
    }
    a.ksplit = pl.ksplit; a.ksteps = pl.ksteps; a.slab = (float*)workspace;
    const bool lds_dma = pl.bm == 128 && pl.bn == 128 && pl.ksplit == 1 && a.fastA && a.vecB && d->Cout % 128 == 0;
    if (ep) {                                   // fused backward epilogue: LDS-DMA kernel, dense output, plain store
        const long long xb = (long long)d->N * d->H * d->W * d->Cin * 4 + ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
        const bool pow2c = d->Cout >= 16 && (d->Cout & (d->Cout - 1)) == 0;
        auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const bool split_ok = pl.ksplit > 1 && pow2c && al(out) && al(res) && al(ep->out) && al(ep->z) && al(ep->scale) &&
                              al(ep->mean) && al(ep->rstd);
        const bool dma_ok = lds_dma && xb < 0x7FFFFFF0LL && !mrcnn_force_flat_glds() && !ep->dy;   // second output: split-K path only
        if (!(dma_ok || split_ok) || !a.dense || bias || scale || z_out || d->act != MRCNN_ACT_NONE || d->res_mode == MRCNN_RES_UP2)
            return MRCNN_ERR_UNSUPPORTED;
        if ((ep->act != MRCNN_ACT_NONE && ep->act != MRCNN_ACT_RELU) || (ep->act == MRCNN_ACT_RELU && !ep->out)) return MRCNN_ERR_ARG;
        if (ep->dgamma && (!ep->z || !ep->mean || !ep->rstd)) return MRCNN_ERR_ARG;
        a.fb_act = ep->act; a.fb_out = ep->out; a.fb_z = ep->z; a.fb_scale = ep->scale; a.fb_mean = ep->mean; a.fb_rstd = ep->rstd;
        a.fb_dgamma = ep->dgamma; a.fb_dbeta = ep->dbeta; a.fb_dbias = ep->dbias;
        a.fb_dy = ep->dy;
        if (ep->dy && (reinterpret_cast<uintptr_t>(ep->dy) & 15)) return MRCNN_ERR_ARG;
    }
    // small layer that the planner would cut into K slices: one launch instead, the four waves of a workgroup split K
    // ... and the layers too small even for that (batch-1 detect: res4 / res5, the upper pyramid levels): 16 x 16 tiles, forward only
    {
        static const bool sk16_on = !(getenv("MRCNN_SK16_KERNEL") && getenv("MRCNN_SK16_KERNEL")[0] == '0');
        static const long long sk16_min = getenv("MRCNN_SK16_MIN_TILES") ? atoll(getenv("MRCNN_SK16_MIN_TILES")) : 64;
        const long long t32 = (long long)((a.M + 31) / 32) * (d->Cout / 32), t16 = (long long)((a.M + 15) / 16) * (d->Cout / 16);
        // too few 32 x 32 tiles for conv_fwd_sk_kernel, or enough of them but a K too short for it (res4 2c at batch 1: K = 256)
        const bool small = t32 < sk_min_tiles() || (t32 <= sk_max_tiles() && a.nk < sk_min_steps() && !getenv("MRCNN_SK16_NO_SHORTK"));
        // inference only (mrcnn_tuning_set("sk16", 1) around engine.infer): in training it measured level, and the float32 replay
        // tests sit within one proposal flip of any change in a forward layer's rounding
        if (sk16_on && g_mrcnn_sk16 && pl.ksplit > 1 && a.fb_act < 0 && conv_sk_ok(d, a) && small && t16 >= sk16_min && t16 <= 2048 && a.Ktot >= 256) {
            const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 4;
            const long long shift = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
            a.ksplit = 1; a.ksteps = a.nk; a.slab = nullptr;
            hipLaunchKernelGGL(conv_fwd_sk16_kernel, dim3((unsigned)t16), dim3(256), 0, s, a, (unsigned)shift, (unsigned)(xbytes + shift));
            return mrcnn_launch_status();
        }
    }
    if (pl.ksplit > 1 && !pl.dma_split && conv_sk_ok(d, a) && (long long)((a.M + 31) / 32) * (d->Cout / 32) <= sk_max_tiles() &&
        (long long)((a.M + 31) / 32) * (d->Cout / 32) >= sk_min_tiles() && a.nk >= sk_min_steps()) {
        const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 4;
        const long long shift = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
        a.ksplit = 1; a.ksteps = a.nk; a.slab = nullptr;
        hipLaunchKernelGGL(conv_fwd_sk_kernel, dim3((unsigned)(((a.M + 31) / 32) * (d->Cout / 32))), dim3(256), 0, s, a, (unsigned)shift,
                           (unsigned)(xbytes + shift));
        return mrcnn_launch_status();
    }
    if (pl.dma_split && pl.ksplit > 1) {        // mid-size layer: LDS-DMA tiles, K cut into slices, slabs reduced by a second launch
        const int mt = (a.M + 127) / 128, nt = a.Cout / 128;
        const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 4;
        const long long shift = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
        hipLaunchKernelGGL(conv_fwd_blds_splitk_kernel, dim3((unsigned)(mt * nt * a.ksplit)), dim3(256), 0, s, a, (unsigned)shift,
                           (unsigned)(xbytes + shift));
        launch_splitk_reduction(a, s);
        return mrcnn_launch_status();
    }
    if (lds_dma) {
        const int mt = (a.M + 127) / 128, nt = a.Cout / 128;
        // buffer-addressed variant when the input (plus the padding shift) fits a 32-bit descriptor range
        const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 4;
        const long long shift = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
        const long long wbytes = (long long)a.Ktot * d->Cout * 4;
        if (xbytes + shift < 0x7FFFFFF0LL && wbytes < 0x7FFFFFF0LL && !mrcnn_force_flat_glds()) {
            const ConvTailPlan tp = plan_conv_tail(d);
            auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
            const bool pow2c = d->Cout >= 16 && (d->Cout & (d->Cout - 1)) == 0;
            const bool tail_ok = tp.ok && a.dense && d->res_mode != MRCNN_RES_UP2 && d->out_mode == MRCNN_OUT_NHWC && workspace &&
                                 workspace_bytes >= tp.slab_bytes && al(workspace) &&
                                 (a.fb_act < 0 || (pow2c && !ep->dy && al(out) && al(res) && al(ep->out) && al(ep->z) && al(ep->scale) &&
                                                   al(ep->mean) && al(ep->rstd)));
            if (!tail_ok) {
                hipLaunchKernelGGL(conv_fwd_blds_kernel, dim3((unsigned)(mt * nt)), dim3(256), 0, s, a, (unsigned)shift,
                                   (unsigned)(xbytes + shift));
                return mrcnn_launch_status();
            }
            // whole rounds: unsplit kernel with its fused epilogue; tail tiles: K slices + slab reduction on the tail rows
            hipLaunchKernelGGL(conv_fwd_blds_kernel, dim3((unsigned)(tp.mt_full * nt)), dim3(256), 0, s, a, (unsigned)shift,
                               (unsigned)(xbytes + shift));
            ConvArgs t = a;
            t.mtile0 = tp.mt_full; t.ksplit = tp.ksplit; t.ksteps = tp.ksteps; t.slab = (float*)workspace;
            hipLaunchKernelGGL(conv_fwd_blds_splitk_kernel, dim3((unsigned)((mt - tp.mt_full) * nt * tp.ksplit)), dim3(256), 0, s, t,
                               (unsigned)shift, (unsigned)(xbytes + shift));
            const long long m1 = (long long)tp.mt_full * 128, off = m1 * d->Cout;
            ConvArgs r = t;                       // the reduction sees the tail rows as a problem of its own (dense NHWC)
            r.M = (int)(a.M - m1); r.mtile0 = 0;
            r.out = a.out + off;
            if (a.z) r.z = a.z + off;
            if (a.res) r.res = a.res + off;
            if (a.fb_act >= 0) {
                if (a.fb_out) r.fb_out = a.fb_out + off;
                if (a.fb_z) r.fb_z = a.fb_z + off;
            }
            launch_splitk_reduction(r, s);
        } else
            hipLaunchKernelGGL(conv_fwd_glds_kernel, dim3((unsigned)(mt * nt)), dim3(256), 0, s, a);
        return mrcnn_launch_status();
    }
    if (pl.bn == 32) return pl.bm == 128 ? launch_conv<128, 32, 4, 1>(a, s) : launch_conv<64, 32, 2, 1>(a, s);
    if (pl.bn == 64) return pl.bm == 128 ? launch_conv<128, 64, 2, 2>(a, s) : launch_conv<64, 64, 2, 2>(a, s);
    return pl.bm == 128 ? launch_conv<128, 128, 2, 2>(a, s) : launch_conv<64, 128, 2, 2>(a, s);
}

// The 16 GEMMs of a Winograd F(2x2, 3x3) layer in one launch of the LDS-DMA kernel: V [nb][rows][K] . U [nb][K][Cout] ->
// Mt [nb][rows][Cout], rows a multiple of 128 (a tile never straddles two weight matrices), K % 16 == 0, Cout % 128 == 0.
// To the kernel it is a 1 x 1 convolution over nb * rows pixels whose weight matrix is chosen by the tile's row block.
extern "C" int mrcnn_gemm_batched_f32(const float* V, const float* U, float* Mt, int nb, int rows, int K, int Cout, void* stream) {
    if (!V || !U || !Mt || nb <= 0 || rows <= 0 || rows % 128 || K <= 0 || K % 16 || Cout <= 0 || Cout % 128) return MRCNN_ERR_ARG;
    const long long M = (long long)nb * rows;
    if (M >= (1LL << 31) || M * K * 4 >= 0x7FFFFFF0LL || (long long)K * Cout * 4 >= 0x7FFFFFF0LL) return MRCNN_ERR_UNSUPPORTED;
    mrcnn_conv_desc d = {};
    d.N = (int)M; d.H = 1; d.W = 1; d.Cin = K; d.Cout = Cout; d.KH = 1; d.KW = 1; d.stride = 1; d.pad_t = 0; d.pad_l = 0; d.OH = 1; d.OW = 1;
    d.act = MRCNN_ACT_NONE; d.res_mode = MRCNN_RES_NONE; d.out_mode = MRCNN_OUT_NHWC; d.cmod = Cout;
    d.out_n_stride = Cout; d.out_h_stride = Cout; d.out_w_stride = Cout;
    ConvArgs a;
    const int rc = conv_fill_args(&d, V, U, nullptr, nullptr, nullptr, nullptr, Mt, nullptr, a);
    if (rc != MRCNN_OK) return rc;
    if (!a.fastA || !a.vecB || !a.dense) return MRCNN_ERR_UNSUPPORTED;
    a.bat_rows = rows;
    const int mt = (int)(M / 128), nt = Cout / 128;
    hipLaunchKernelGGL(conv_fwd_blds_kernel, dim3((unsigned)(mt * nt)), dim3(256), 0, (hipStream_t)stream, a, 0u, (unsigned)(M * K * 4));
    return mrcnn_launch_status();
}

extern "C" int mrcnn_conv2d_fwd_ws(const mrcnn_conv_desc* d, const float* x, const float* w,
                                   const float* bias, const float* scale, const float* shift,
                                   const float* res, float* out, float* z_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return conv_fwd_impl(d, x, w, bias, scale, shift, res, out, z_out, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int mrcnn_conv2d_dgrad_ep(const mrcnn_conv_desc* d, const float* dz, const float* w_t, const float* res,
                                     float* dz_below, const mrcnn_bwd_epilogue* ep, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!ep) return MRCNN_ERR_ARG;
    return conv_fwd_impl(d, dz, w_t, nullptr, nullptr, nullptr, res, dz_below, nullptr, workspace, workspace_bytes, ep, stream);
}

extern "C" int mrcnn_conv2d_fwd(const mrcnn_conv_desc* d, const float* x, const float* w,
                                const float* bias, const float* scale, const float* shift,
                                const float* res, float* out, float* z_out, void* stream) {
    return mrcnn_conv2d_fwd_ws(d, x, w, bias, scale, shift, res, out, z_out, nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------------
// mrcnn_conv2d_fwd_multi: n <= 5 independent convolutions in one launch (+ one launch for their split-K reductions).
// The reference applies the shared RPN model to the pyramid levels in a Python loop (model.py:2040-2055) and the FPN
// smoothing convolutions one Keras layer after the other (model.py:2018-2026); on small feature maps those are a few
// workgroups each and every dependent launch costs ~5 us, so the independent ones travel together.  Same arithmetic per
// problem as mrcnn_conv2d_fwd_ws (same kernels, same K order within a slice; the slice count may differ).
struct ConvMultiPlan { int bn, ks[CONV_MULTI_MAX], ksteps[CONV_MULTI_MAX]; size_t slab_off[CONV_MULTI_MAX], bytes; bool fast; };

static int plan_conv_multi(const mrcnn_conv_problem* pr, int n, ConvMultiPlan& pl) {
    if (!pr || n < 1 || n > CONV_MULTI_MAX) return MRCNN_ERR_ARG;
    auto bn_of = [](int Cout) { return Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128); };
    pl.bn = bn_of(pr[0].d.Cout);
    long long tiles = 0;
    for (int g = 0; g < n; ++g) {
        const mrcnn_conv_desc& d = pr[g].d;
        if (d.N <= 0 || d.OH <= 0 || d.OW <= 0 || d.Cout <= 0 || d.KH <= 0 || d.KW <= 0 || d.Cin <= 0) return MRCNN_ERR_ARG;
        if (bn_of(d.Cout) != pl.bn) return MRCNN_ERR_UNSUPPORTED;
        const long long M = (long long)d.N * d.OH * d.OW;
        tiles += ((M + 63) / 64) * ((d.Cout + pl.bn - 1) / pl.bn);
    }
    if (tiles >= (1 << 20)) return MRCNN_ERR_UNSUPPORTED;
    const long long want = tiles < 768 ? (1024 + tiles - 1) / tiles : 1;   // as plan_conv, on the launch as a whole
    pl.bytes = 0;
    for (int g = 0; g < n; ++g) {
        const mrcnn_conv_desc& d = pr[g].d;
        const int nk = (d.KH * d.KW * d.Cin + 31) / 32;
        long long ks = want < nk / 4 ? want : nk / 4;
        if (ks > 16) ks = 16;
        pl.ks[g] = 1; pl.ksteps[g] = nk;
        if (nk >= 8 && ks >= 2) {
            pl.ksteps[g] = (int)((nk + ks - 1) / ks);
            pl.ks[g] = (nk + pl.ksteps[g] - 1) / pl.ksteps[g];
        }
        pl.slab_off[g] = pl.bytes;
        if (pl.ks[g] > 1) pl.bytes += (((size_t)pl.ks[g] * d.N * d.OH * d.OW * d.Cout * sizeof(float)) + 255) & ~(size_t)255;
    }
    return MRCNN_OK;
}

extern "C" size_t mrcnn_conv2d_fwd_multi_workspace(const mrcnn_conv_problem* problems, int n) {
    ConvMultiPlan pl;
    if (plan_conv_multi(problems, n, pl) != MRCNN_OK) return 0;
    return pl.bytes;
}

template <int BN, int WN_, int WM_>
static void launch_conv_multi(const ConvMultiArgs& mp, unsigned blocks, bool fast, hipStream_t s) {
    if (fast) hipLaunchKernelGGL((conv_fwd_multi_kernel<64, BN, WM_, WN_, true>), dim3(blocks), dim3(WM_ * WN_ * 64), 0, s, mp);
    else hipLaunchKernelGGL((conv_fwd_multi_kernel<64, BN, WM_, WN_, false>), dim3(blocks), dim3(WM_ * WN_ * 64), 0, s, mp);
}

extern "C" int mrcnn_conv2d_fwd_multi(const mrcnn_conv_problem* problems, int n, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    ConvMultiPlan pl;
    int rc = plan_conv_multi(problems, n, pl);
    if (rc != MRCNN_OK) return rc;
    if (pl.bytes && (!workspace || workspace_bytes < pl.bytes || (reinterpret_cast<uintptr_t>(workspace) & 15))) return MRCNN_ERR_ARG;
    ConvMultiArgs mp;
    mp.n = n;
    bool fast = true;
    long long blocks = 0, eblocks = 0;
    for (int g = 0; g < n; ++g) {
        const mrcnn_conv_problem& q = problems[g];
        rc = conv_fill_args(&q.d, q.x, q.w, q.bias, q.scale, q.shift, q.res, q.out, q.z_out, mp.a[g]);
        if (rc != MRCNN_OK) return rc;
        ConvArgs& a = mp.a[g];
        fast = fast && a.fastA;
        a.ksplit = pl.ks[g]; a.ksteps = pl.ksteps[g];
        a.slab = pl.ks[g] > 1 ? (float*)((char*)workspace + pl.slab_off[g]) : nullptr;
        mp.first[g] = (int)blocks;
        blocks += (long long)((a.M + 63) / 64) * ((a.Cout + pl.bn - 1) / pl.bn) * a.ksplit;
        mp.efirst[g] = (int)eblocks;
        mp.evec[g] = 0;
        if (a.ksplit > 1) {
            mp.evec[g] = splitk_epilogue_vec_ok(a) ? 1 : 0;
            const long long ne = (long long)a.M * a.Cout;
            eblocks += mp.evec[g] ? cdiv64(ne / 4, 256) : cdiv64(ne, 256);
        }
    }
    mp.first[n] = (int)blocks; mp.efirst[n] = (int)eblocks;
    for (int g = n; g < CONV_MULTI_MAX; ++g) { mp.first[g + 1] = (int)blocks; mp.efirst[g + 1] = (int)eblocks; mp.evec[g] = 0; }
    if (!fast)
        for (int g = 0; g < n; ++g) mp.a[g].fastA = 0;
    if (blocks >= (1ll << 31) || eblocks >= (1ll << 31)) return MRCNN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (pl.bn == 32) launch_conv_multi<32, 1, 2>(mp, (unsigned)blocks, fast, s);
    else if (pl.bn == 64) launch_conv_multi<64, 2, 2>(mp, (unsigned)blocks, fast, s);
    else launch_conv_multi<128, 2, 2>(mp, (unsigned)blocks, fast, s);
    if (eblocks > 0) hipLaunchKernelGGL(conv_splitk_epilogue_multi_kernel, dim3((unsigned)eblocks), dim3(256), 0, s, mp);
    return mrcnn_launch_status();
}
