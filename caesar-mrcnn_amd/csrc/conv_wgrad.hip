// Weight gradient of the implicit-GEMM convolution on the fp32 matrix cores:
//   dW[(kh,kw,ci), co] = sum over pixels m of  X[n, oh*s+kh-pt, ow*s+kw-pl, ci] * dY[m, co]
// (the reference obtains it from TF autodiff of KL.Conv2D; mrcnn/model.py:2487 fit_generator).
// GEMM view: rows i = (tap, ci) in [0, K), columns co, contraction over the M = N*OH*OW pixels.
// A workgroup owns a BI x BN tile of dW for one pixel range ("split"); BI divides Cin on the fast path
// so the tile sits inside one filter tap and its X rows are contiguous channel runs.  Both operands are
// staged pixel-major in LDS ([32 pixels][channels]) which is already the layout the 32x32x2 MFMA wants
// (lane = channel, k = pixel): every LDS read is a conflict-free ds_read_b32.  Splits write partial
// slabs that a second kernel sums in a fixed order, so dW is bitwise reproducible.
#include "common.h"
#include <type_traits>
#include <cstdlib>

struct WgradArgs {
    const float* x; const float* dy; float* out;   // out: dw (splits==1) or slabs
    int N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, OH, OW;
    int M, Ktot, fast, splits, chunk, acc;
    FastDiv d_ohw, d_ow;
    const PixelEntry* table;      // large layers: per-pixel offsets / tap masks (conv_wgrad_blds_kernel<.., true>)
};

template <int BI, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BI / WM / 32, TN = BN / WN / 32;
    constexpr int XST = BI + 4, YST = BN + 4;
    constexpr int XV = 8 * BI / NT, YV = 8 * BN / NT;     // float4 loads per thread per step
    __shared__ __attribute__((aligned(16))) float lds[32 * XST + 32 * YST];
    float* Xs = lds;
    float* Ys = lds + 32 * XST;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = (p.Cout + BN - 1) / BN;
    const int itiles = (p.Ktot + BI - 1) / BI;
    int bid = blockIdx.x;
    const int ntile = bid % ntiles; bid /= ntiles;
    const int itile = bid % itiles;
    const int split = bid / itiles;
    const int i0 = itile * BI, n0 = ntile * BN;
    const int m_begin = split * p.chunk;
    const int m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
    const int tap = i0 / p.Cin, ci0 = i0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int ohw = p.OH * p.OW;

    f32x4 rx[XV], ry[YV];
    auto load_tiles = [&](int mb) {
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int idx = tid + v * NT;
            const int row = idx / (BI / 4), c4 = idx % (BI / 4);
            const int m = mb + row;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                const int n = (int)fast_div((unsigned)m, p.d_ohw), rem = m - n * ohw;
                const int oh = (int)fast_div((unsigned)rem, p.d_ow), ow = rem - oh * p.OW;
                if (p.fast) {
                    const int ih = oh * p.stride - p.pad_t + kh, iw = ow * p.stride - p.pad_l + kw;
                    if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                        val = *(const f32x4*)(p.x + (((long long)n * p.H + ih) * p.W + iw) * p.Cin + ci0 + c4 * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = i0 + c4 * 4 + e;
                        if (i < p.Ktot) {
                            const int t2 = i / p.Cin, ci = i - t2 * p.Cin;
                            const int k2 = t2 / p.KW, l2 = t2 - k2 * p.KW;
                            const int ih = oh * p.stride - p.pad_t + k2, iw = ow * p.stride - p.pad_l + l2;
                            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                                val[e] = p.x[(((long long)n * p.H + ih) * p.W + iw) * p.Cin + ci];
                        }
                    }
                }
            }
            rx[v] = val;
        }
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            const int idx = tid + v * NT;
            const int row = idx / (BN / 4), c4 = idx % (BN / 4);
            const int m = mb + row, n = n0 + c4 * 4;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                const float* ptr = p.dy + (long long)m * p.Cout + n;
                if ((p.Cout & 3) == 0) {
                    if (n < p.Cout) val = *(const f32x4*)ptr;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Cout) val[e] = ptr[e];
                }
            }
            ry[v] = val;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int idx = tid + v * NT;
            *(f32x4*)&Xs[(idx / (BI / 4)) * XST + (idx % (BI / 4)) * 4] = rx[v];
        }
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            const int idx = tid + v * NT;
            *(f32x4*)&Ys[(idx / (BN / 4)) * YST + (idx % (BN / 4)) * 4] = ry[v];
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
    if (m_begin < m_end) {
        load_tiles(m_begin);
        store_tiles();
        __syncthreads();
        for (int mb = m_begin; mb < m_end; mb += 32) {
            const bool more = mb + 32 < m_end;
            if (more) load_tiles(mb + 32);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int prow = lh * 16 + t;
                float av[TM], bv[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) av[a] = Xs[prow * XST + wm * TM * 32 + a * 32 + li];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[b] = Ys[prow * YST + wn * TN * 32 + b * 32 + li];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    float* dst = p.out + (p.splits > 1 ? (long long)split * p.Ktot * p.Cout : 0LL);
    const bool accumulate = p.splits == 1 && p.acc;
    auto store_tile = [&](const f32x16& c, int ibase, int n) {
        if (n >= p.Cout) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ibase + (r & 3) + 8 * (r >> 2);
            if (i < p.Ktot) {
                float* q = dst + (long long)i * p.Cout + n;
                *q = accumulate ? *q + c[r] : c[r];
            }
        }
    };
    const int ib = i0 + wm * TM * 32 + 4 * lh, nb = n0 + wn * TN * 32 + li;
    if constexpr (TM >= 1 && TN >= 1) store_tile(acc[0][0], ib, nb);
    if constexpr (TM >= 1 && TN >= 2) store_tile(acc[0][1], ib, nb + 32);
    if constexpr (TM >= 2 && TN >= 1) store_tile(acc[1][0], ib + 32, nb);
    if constexpr (TM >= 2 && TN >= 2) store_tile(acc[1][1], ib + 32, nb + 32);
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant for full tiles (Cin % BI == 0, Cout % BN == 0): both operand tiles are [BP pixels][128
// channels] with 512-byte rows, exactly the lane-linear image a `global_load_lds_dwordx4` writes (1 KiB =
// 2 rows per wave-instruction) and exactly what the MFMA operand reads want (lane = channel: conflict-free
// ds_read_b32, no padding, no swizzle).  No staging VGPRs, no ds_write, one barrier per step: the DMA of
// step t+1 lands in the other LDS buffer while the MFMAs of step t run.  Out-of-image taps and rows past
// the pixel range read from a 64-byte zero page instead of being masked.
__device__ __attribute__((aligned(64))) float g_zero_page[16];

#ifndef WGRAD_BP
#define WGRAD_BP 16
#endif

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int BI, int BN, int BP>
__global__ __launch_bounds__(256, 2) void conv_wgrad_glds_kernel(const WgradArgs p) {
    constexpr int TM = BI / 2 / 32, TN = BN / 2 / 32;
    constexpr int XF = BP * BI, YF = BP * BN;
    constexpr int NXI = XF / 256, NYI = YF / 256;            // 1 KiB DMA pieces per tile
    constexpr int XLPR = BI / 4, YLPR = BN / 4;              // lanes per row
    constexpr int XRPI = 64 / XLPR, YRPI = 64 / YLPR;        // rows per piece
    __shared__ __attribute__((aligned(16))) float lds[2 * (XF + YF)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.Cout / BN, itiles = p.Ktot / BI;
    int bid = blockIdx.x;
    const int ntile = bid % ntiles; bid /= ntiles;
    const int itile = bid % itiles;
    const int split = bid / itiles;
    const int i0 = itile * BI, n0 = ntile * BN;
    const int m_begin = split * p.chunk;
    const int m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
    const int tap = i0 / p.Cin, ci0 = i0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int ohw = p.OH * p.OW;

    auto stage = [&](int buf, int mb) {
        float* xb = lds + buf * (XF + YF);
        float* yb = xb + XF;
#pragma unroll
        for (int jj = 0; jj < (NXI + 3) / 4; ++jj) {
            const int j = wave + jj * 4;
            if (j < NXI) {
                const int m = mb + j * XRPI + lane / XLPR, c4 = lane % XLPR;
                const float* src = g_zero_page;
                if (m < m_end) {
                    const int n = (int)fast_div((unsigned)m, p.d_ohw), rem = m - n * ohw;
                    const int oh = (int)fast_div((unsigned)rem, p.d_ow), ow = rem - oh * p.OW;
                    const int ih = oh * p.stride - p.pad_t + kh, iw = ow * p.stride - p.pad_l + kw;
                    if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                        src = p.x + (((long long)n * p.H + ih) * p.W + iw) * p.Cin + ci0 + c4 * 4;
                }
                glds16(src, xb + j * 256);
            }
        }
#pragma unroll
        for (int jj = 0; jj < (NYI + 3) / 4; ++jj) {
            const int j = wave + jj * 4;
            if (j < NYI) {
                const int m = mb + j * YRPI + lane / YLPR, c4 = lane % YLPR;
                const float* src = (m < m_end) ? p.dy + (long long)m * p.Cout + n0 + c4 * 4 : g_zero_page;
                glds16(src, yb + j * 256);
            }
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
    if (m_begin < m_end) {
        stage(0, m_begin);
        __syncthreads();
        int cur = 0;
        for (int mb = m_begin; mb < m_end; mb += BP) {
            if (mb + BP < m_end) stage(cur ^ 1, mb + BP);
            const float* xb = lds + cur * (XF + YF);
            const float* yb = xb + XF;
#pragma unroll
            for (int t = 0; t < BP / 2; ++t) {
                const int prow = lh * (BP / 2) + t;
                float av[TM], bv[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) av[a] = xb[prow * BI + wm * TM * 32 + a * 32 + li];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[b] = yb[prow * BN + wn * TN * 32 + b * 32 + li];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    float* dst = p.out + (p.splits > 1 ? (long long)split * p.Ktot * p.Cout : 0LL);
    const bool accumulate = p.splits == 1 && p.acc;
    auto store_tile = [&](const f32x16& c, int ibase, int n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ibase + (r & 3) + 8 * (r >> 2);
            float* q = dst + (long long)i * p.Cout + n;
            *q = accumulate ? *q + c[r] : c[r];
        }
    };
    const int ib = i0 + wm * TM * 32 + 4 * lh, nb = n0 + wn * TN * 32 + li;
    if constexpr (TM >= 1 && TN >= 1) store_tile(acc[0][0], ib, nb);
    if constexpr (TM >= 1 && TN >= 2) store_tile(acc[0][1], ib, nb + 32);
    if constexpr (TM >= 2 && TN >= 1) store_tile(acc[1][0], ib + 32, nb);
    if constexpr (TM >= 2 && TN >= 2) store_tile(acc[1][1], ib + 32, nb + 32);
}

// Same tile with buffer addressing (default; the flat variant above stays for tensors >= 2 GiB).  The flat
// variant spends ~3 vector instructions per MFMA on per-lane pixel decomposition (two divisions), bounds
// tests and 64-bit pointers, and vector instructions do not overlap the matrix pipe of their SIMD.  Here a
// 1 KiB piece = two pixel rows, so the pixel -> (n, oh, ow) decomposition is wave-uniform and runs on the
// SCALAR unit; a lane only selects one of the two row offsets and adds its channel offset (3 vector
// instructions per X piece, none per dY piece); padded taps and rows past the tensor take an out-of-range
// offset and the buffer range check writes zeros; the dY descriptor is rebuilt per piece (scalar) with the
// rows left in this split as its size, so the pixel tail needs no per-lane test either.  The loop is
// unrolled over the two LDS buffers: every LDS read is base register + immediate.
typedef __attribute__((address_space(3))) void* wgrad_lds_ptr;
#define WGRAD_OOB_OFFSET 0x80000000u

template <int BP, bool TABLE>
__device__ __forceinline__ void conv_wgrad_blds_body(const WgradArgs& p, const unsigned x_shift, const unsigned x_records,
                                                     const int block) {
    constexpr int BI = 128, BN = 128, TM = 2, TN = 2;
    constexpr int XF = BP * BI, YF = BP * BN;
    constexpr int NP = XF / 256;                              // 1 KiB pieces per tile (two pixel rows each)
    __shared__ __attribute__((aligned(16))) float lds[2 * (XF + YF)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.Cout / BN, itiles = p.Ktot / BI;
    int bid = block;
    const int ntile = bid % ntiles; bid /= ntiles;
    const int itile = bid % itiles;
    const int split = bid / itiles;
    const int i0 = itile * BI, n0 = ntile * BN;
    const int m_begin = split * p.chunk;
    const int m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
    const int tap = i0 / p.Cin, ci0 = i0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int ohw = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - x_shift), 0, x_records, 0x00020000);
    const unsigned lane_chan = (unsigned)(lane & 31) * 16u;
    const bool upper = lane >= 32;
    const unsigned y_voff = (unsigned)(((lane >> 5) * p.Cout + (lane & 31) * 4) * 4);

    // TABLE (layers with many pixels): the row offsets and tap masks come from a table built once per call and are
    // prefetched one step ahead -- nothing but a bit test is left per piece; otherwise the decomposition runs on the
    // scalar unit (no extra launch for the many small layers).
    const unsigned soff_tab = (unsigned)(((kh * p.W + kw) * p.Cin + ci0) * 4);
    const unsigned tapbit = 1u << (tap & 31);
    const bool tap_hi = tap >= 32;
    const PixelEntry* tab = TABLE ? p.table + (lane >> 5) : nullptr;
    PixelEntry ent[NP / 4];
    auto fetch = [&](int mb) {
        if constexpr (TABLE) {
#pragma unroll
            for (int jj = 0; jj < NP / 4; ++jj) ent[jj] = tab[mb + 2 * (wave + jj * 4)];
        }
    };
    auto stage = [&](float* xb, int mb) {
        float* yb = xb + XF;
#pragma unroll
        for (int jj = 0; jj < NP / 4; ++jj) {
            const int j = wave + jj * 4;
            if constexpr (TABLE) {
                const unsigned word = tap_hi ? ent[jj].mask_hi : ent[jj].mask_lo;
                const unsigned vo = ((word & tapbit) ? ent[jj].off : WGRAD_OOB_OFFSET) + lane_chan;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (wgrad_lds_ptr)(xb + j * 256), 16, vo, soff_tab, 0, 0);
                continue;
            }
            unsigned off[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                       // wave-uniform: scalar unit
                const int m = mb + j * 2 + h;
                const int n = (int)fast_div((unsigned)m, p.d_ohw), rem = m - n * ohw;
                const int oh = (int)fast_div((unsigned)rem, p.d_ow), ow = rem - oh * p.OW;
                const int ih = oh * p.stride - p.pad_t + kh, iw = ow * p.stride - p.pad_l + kw;
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                off[h] = ok ? (unsigned)((((n * p.H + ih) * p.W + iw) * p.Cin + ci0) * 4) + x_shift : WGRAD_OOB_OFFSET;
            }
            const unsigned vo = (upper ? off[1] : off[0]) + lane_chan;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (wgrad_lds_ptr)(xb + j * 256), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < NP / 4; ++jj) {
            const int j = wave + jj * 4;
            const int row0 = mb + j * 2;
            const int left = m_end - row0;                      // rows of this split from row0 on
            const unsigned rec = left > 0 ? (unsigned)(((left - 1) * p.Cout + BN) * 4) : 0u;
            const __amdgpu_buffer_rsrc_t rsrc_y =
                __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + (long long)row0 * p.Cout + n0), 0, rec, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (wgrad_lds_ptr)(yb + j * 256), 16, y_voff, 0, 0, 0);
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
    const float* x_rd = lds + lh * (BP / 2) * BI + wm * TM * 32 + li;
    const float* y_rd = lds + XF + lh * (BP / 2) * BN + wn * TN * 32 + li;
    auto compute = [&](auto curc) {
        constexpr int BO = decltype(curc)::value * (XF + YF);
#pragma unroll
        for (int t = 0; t < BP / 2; ++t) {
            float av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = x_rd[BO + t * BI + a * 32];
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = y_rd[BO + t * BN + b * 32];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    };

    if (m_begin < m_end) {
        // Table entries are fetched one half-step ahead of the stage() that consumes them, UNCONDITIONALLY (index
        // clamped into the table, which carries BP rows past M): a fetch under `if (more rows)` makes the entries a
        // conditional loop-carried value, the compiler copies the freshly loaded registers at the join, and the wait
        // those copies need (vmcnt(0): younger than the LDS-DMA just issued) stalled every step until the next
        // tile's DMA had landed -- no overlap of DMA and MFMA left (the table variant ran at 79 % MFMA-busy).
        auto fetch_at = [&](int mb) { fetch(mb < p.M ? mb : p.M); };
        fetch(m_begin);
        stage(lds, m_begin);
        fetch_at(m_begin + BP);
        __syncthreads();
        for (int mb = m_begin; mb < m_end; mb += 2 * BP) {
            const bool has1 = mb + BP < m_end;
            if (has1) stage(lds + (XF + YF), mb + BP);
            fetch_at(mb + 2 * BP);
            compute(std::integral_constant<int, 0>{});
            __syncthreads();
            if (has1 && mb + 2 * BP < m_end) stage(lds, mb + 2 * BP);
            fetch_at(mb + 3 * BP);
            if (has1) {
                compute(std::integral_constant<int, 1>{});
                __syncthreads();
            }
        }
    }

    float* dst = p.out + (p.splits > 1 ? (long long)split * p.Ktot * p.Cout : 0LL);
    const bool accumulate = p.splits == 1 && p.acc;
    auto store_tile = [&](const f32x16& c, int ibase, int n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ibase + (r & 3) + 8 * (r >> 2);
            float* q = dst + (long long)i * p.Cout + n;
            *q = accumulate ? *q + c[r] : c[r];
        }
    };
    const int ib = i0 + wm * TM * 32 + 4 * lh, nb = n0 + wn * TN * 32 + li;
    store_tile(acc[0][0], ib, nb);
    store_tile(acc[0][1], ib, nb + 32);
    store_tile(acc[1][0], ib + 32, nb);
    store_tile(acc[1][1], ib + 32, nb + 32);
}

template <int BP, bool TABLE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_blds_kernel(const WgradArgs p, const unsigned x_shift, const unsigned x_records) {
    conv_wgrad_blds_body<BP, TABLE>(p, x_shift, x_records, (int)blockIdx.x);
}

// Up to WGRAD_MULTI_MAX weight gradients in one launch (mrcnn_conv2d_wgrad_multi): the three convolutions of a
// bottleneck block.  Workgroups first[g] .. first[g+1]-1 belong to problem g; the slab reductions share a launch too.
#define WGRAD_MULTI_MAX 16      // 4 covers a bottleneck block; 16 = the transform-domain weight gradients of a Winograd F(2x2) layer (F(4x4): 3 x 12; 18 per launch measured no faster)
struct WgradMultiArgs {
    WgradArgs a[WGRAD_MULTI_MAX];
    unsigned x_shift[WGRAD_MULTI_MAX], x_records[WGRAD_MULTI_MAX];
    int first[WGRAD_MULTI_MAX + 1];
    float* dw[WGRAD_MULTI_MAX];
    int rfirst[WGRAD_MULTI_MAX + 1];      // reduction launch: blocks of 256 float4
    int n;
};

static_assert(sizeof(WgradMultiArgs) <= 4096, "kernel argument block");

template <int BP>
__global__ __launch_bounds__(256, 2) void conv_wgrad_blds_multi_kernel(const WgradMultiArgs mp) {
    int g = 0;
    while (g + 1 < mp.n && (int)blockIdx.x >= mp.first[g + 1]) ++g;
    conv_wgrad_blds_body<BP, false>(mp.a[g], mp.x_shift[g], mp.x_records[g], (int)blockIdx.x - mp.first[g]);
}

__global__ void wgrad_reduce_multi_kernel(const WgradMultiArgs mp) {
    int g = 0;
    while (g + 1 < mp.n && (int)blockIdx.x >= mp.rfirst[g + 1]) ++g;
    const WgradArgs& p = mp.a[g];
    if (p.splits <= 1) return;
    const long long n4 = (long long)p.Ktot * p.Cout / 4;
    const long long i = (long long)(blockIdx.x - mp.rfirst[g]) * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (p.acc) s = *(const f32x4*)(mp.dw[g] + 4 * i);
    *(f32x4*)(mp.dw[g] + 4 * i) = mrcnn_slab_sum<f32x4>(s, p.out, 4 * n4, 4 * i, p.splits);

}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, float* dw, long long n, int splits, int acc) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dw[i] = mrcnn_slab_sum<float>(acc ? dw[i] : 0.f, slabs, n, i, splits);
}

// n % 4 == 0, 16-byte aligned buffers: four elements per thread
__global__ void wgrad_reduce_vec_kernel(const float* __restrict__ slabs, float* dw, long long n4, int splits, int acc) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (acc) s = *(const f32x4*)(dw + 4 * i);
    *(f32x4*)(dw + 4 * i) = mrcnn_slab_sum<f32x4>(s, slabs, 4 * n4, 4 * i, splits);
}

struct WgradPlan { int bi, bn, splits, chunk, fast; };

static WgradPlan plan_wgrad(const mrcnn_conv_desc* d) {
    WgradPlan pl;
    const long long M = (long long)d->N * d->OH * d->OW;
    const int Ktot = d->KH * d->KW * d->Cin;
    pl.fast = 1;
    if (d->Cin % 128 == 0) pl.bi = 128;
    else if (d->Cin % 64 == 0) pl.bi = 64;
    else { pl.bi = 64; pl.fast = 0; }
    pl.bn = d->Cout >= 128 ? 128 : (d->Cout >= 64 ? 64 : 32);
    const long long tiles = (long long)((Ktot + pl.bi - 1) / pl.bi) * ((d->Cout + pl.bn - 1) / pl.bn);
    long long splits = 1024 / (tiles > 0 ? tiles : 1);
    static const long long min_px = getenv("MRCNN_WGRAD_MIN_PIXELS") ? atoll(getenv("MRCNN_WGRAD_MIN_PIXELS")) : 128;
    const long long max_splits = (M + min_px - 1) / min_px;       // at least min_px pixels per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    long long chunk = ((M + splits - 1) / splits + 31) / 32 * 32;
    splits = (M + chunk - 1) / chunk;
    pl.splits = (int)splits;
    pl.chunk = (int)chunk;
    return pl;
}

// layers with at least this many output pixels address X through the pixel table (one extra tiny launch)
#define WGRAD_TABLE_MIN_PIXELS 65536
static size_t wgrad_table_bytes(const mrcnn_conv_desc* d) {
    const long long M = (long long)d->N * d->OH * d->OW;
    const bool lds_dma = d->Cin % 128 == 0 && d->Cout % 128 == 0 && d->KH * d->KW <= 64;
    return (lds_dma && M >= WGRAD_TABLE_MIN_PIXELS) ? (size_t)(M + WGRAD_BP) * sizeof(PixelEntry) + 256 : 0;
}

extern "C" size_t mrcnn_conv2d_wgrad_workspace(const mrcnn_conv_desc* d) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
    WgradPlan pl = plan_wgrad(d);
    const size_t slabs = pl.splits > 1 ? (size_t)pl.splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float) : 0;
    return slabs + wgrad_table_bytes(d);
}

template <int BI, int BN, int WM, int WN>
static void launch_wgrad(const WgradArgs& a, hipStream_t s) {
    const int blocks = ((a.Ktot + BI - 1) / BI) * ((a.Cout + BN - 1) / BN) * a.splits;
    hipLaunchKernelGGL((conv_wgrad_kernel<BI, BN, WM, WN>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0, s, a);
}

extern "C" int mrcnn_conv2d_wgrad(const mrcnn_conv_desc* d, const float* x, const float* dy, float* dw,
                                  float* workspace, size_t workspace_bytes, int beta_acc, void* stream) {
    if (!d || !x || !dy || !dw) return MRCNN_ERR_ARG;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 ||
        d->stride <= 0 || d->OH <= 0 || d->OW <= 0)
        return MRCNN_ERR_ARG;
    const long long M = (long long)d->N * d->OH * d->OW;
    if (M >= (1LL << 31)) return MRCNN_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return MRCNN_ERR_ARG;
    WgradPlan pl = plan_wgrad(d);
    const size_t slab_bytes = pl.splits > 1 ? (size_t)pl.splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float) : 0;
    if (pl.splits > 1 && (!workspace || workspace_bytes < slab_bytes)) return MRCNN_ERR_WORKSPACE;
    WgradArgs a;
    a.x = x; a.dy = dy; a.out = pl.splits > 1 ? workspace : dw;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.OH = d->OH; a.OW = d->OW;
    a.M = (int)M; a.Ktot = d->KH * d->KW * d->Cin; a.fast = pl.fast; a.splits = pl.splits; a.chunk = pl.chunk;
    a.acc = beta_acc;
    a.d_ohw = make_fastdiv((unsigned)(d->OH * d->OW)); a.d_ow = make_fastdiv((unsigned)d->OW);
    hipStream_t s = (hipStream_t)stream;
    if (pl.fast && pl.bi == 128 && pl.bn == 128 && d->Cout % 128 == 0 && a.Ktot % 128 == 0) {
        const int blocks = (a.Ktot / 128) * (a.Cout / 128) * a.splits;
        const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 4;
        const long long shift = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
        const long long ybytes = M * d->Cout * 4;
        // + 16 pixel rows: offsets of rows past the tensor must not wrap before the range check sees them
        if (xbytes + shift + 16LL * d->H * d->W * d->Cin * 4 < 0x7FFFFFF0LL && ybytes < 0x7FFFFFF0LL && !mrcnn_force_flat_glds()) {
            const size_t tb = wgrad_table_bytes(d);
            a.table = nullptr;
            if (tb && workspace && workspace_bytes >= slab_bytes + tb) {
                char* q = (char*)workspace + slab_bytes;
                a.table = (const PixelEntry*)((reinterpret_cast<uintptr_t>(q) + 255) & ~(uintptr_t)255);
                const int rows = (int)M + WGRAD_BP;
                hipLaunchKernelGGL(pixel_table_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (PixelEntry*)a.table, d->N,
                                   d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->OH, d->OW, (int)M, rows,
                                   (unsigned)shift, 4);
                hipLaunchKernelGGL((conv_wgrad_blds_kernel<WGRAD_BP, true>), dim3((unsigned)blocks), dim3(256),
                                   (size_t)g_mrcnn_wgrad_lds_pad, s, a, (unsigned)shift, (unsigned)(xbytes + shift));
            } else {
                hipLaunchKernelGGL((conv_wgrad_blds_kernel<WGRAD_BP, false>), dim3((unsigned)blocks), dim3(256),
                                   (size_t)g_mrcnn_wgrad_lds_pad, s, a, (unsigned)shift, (unsigned)(xbytes + shift));
            }
        }
        else
            hipLaunchKernelGGL((conv_wgrad_glds_kernel<128, 128, WGRAD_BP>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    } else if (pl.bi == 128) {
        if (pl.bn == 128) launch_wgrad<128, 128, 2, 2>(a, s);
        else if (pl.bn == 64) launch_wgrad<128, 64, 2, 2>(a, s);
        else launch_wgrad<128, 32, 4, 1>(a, s);
    } else {
        if (pl.bn == 128) launch_wgrad<64, 128, 2, 2>(a, s);
        else if (pl.bn == 64) launch_wgrad<64, 64, 2, 2>(a, s);
        else launch_wgrad<64, 32, 2, 1>(a, s);
    }
    if (pl.splits > 1) {
        const long long n = (long long)a.Ktot * a.Cout;
        if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0)
            hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3((unsigned)cdiv64(n / 4, 256)), dim3(256), 0, s,
                               (const float*)workspace, dw, n / 4, pl.splits, beta_acc);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, workspace, dw, n,
                               pl.splits, beta_acc);
    }
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// mrcnn_conv2d_wgrad_multi: the weight gradients of up to 4 layers in one launch (+ one for their slab reductions).
// On the small feature maps of the backbone a weight gradient is 16-36 output tiles with a few hundred pixels to
// contract: ~30 us of mostly latency per launch plus its reduction, three per bottleneck block, and together they
// had become the longest chain of the backward pass (knock-out run: 2 ms of the step).  LDS-DMA kernel only.
struct WgradMultiPlan { int splits[WGRAD_MULTI_MAX], chunk[WGRAD_MULTI_MAX]; size_t slab_off[WGRAD_MULTI_MAX], bytes; };

static bool plan_wgrad_multi(const mrcnn_wgrad_problem* pr, int n, WgradMultiPlan& pl) {
    if (!pr || n < 1 || n > WGRAD_MULTI_MAX) return false;
    long long tiles = 0;
    for (int g = 0; g < n; ++g) {
        const mrcnn_conv_desc& d = pr[g].d;
        if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.KH <= 0 || d.KW <= 0 || d.stride <= 0 ||
            d.OH <= 0 || d.OW <= 0)
            return false;
        const long long M = (long long)d.N * d.OH * d.OW;
        const long long Ktot = (long long)d.KH * d.KW * d.Cin;
        const bool gemm = d.KH == 1 && d.KW == 1 && d.H == 1 && d.W == 1 && d.stride == 1;     // pure GEMM rows: no pixel decomposition to pay for
        if (d.Cin % 128 || d.Cout % 128 || d.KH * d.KW > 64 || (M >= WGRAD_TABLE_MIN_PIXELS && !gemm)) return false;
        const long long xbytes = (long long)d.N * d.H * d.W * d.Cin * 4;
        const long long shift = ((long long)d.pad_t * d.W + d.pad_l) * d.Cin * 4;
        if (xbytes + shift + 16LL * d.H * d.W * d.Cin * 4 >= 0x7FFFFFF0LL || M * d.Cout * 4 >= 0x7FFFFFF0LL) return false;
        tiles += (Ktot / 128) * (d.Cout / 128);
    }
    // workgroups the launch aims at: 1024 = four per CU (what a launch gets beside another stream's kernels); MRCNN_WGRAD_MULTI_TARGET
    // for A/B (1280 = all five slots per CU: the 36 transform-domain GEMMs of a Winograd layer go 48 tiles x 21 -> x 26 splits)
    static const long long target = getenv("MRCNN_WGRAD_MULTI_TARGET") ? atoll(getenv("MRCNN_WGRAD_MULTI_TARGET")) : 1024;
    const long long want = target / (tiles > 0 ? tiles : 1);
    pl.bytes = 0;
    for (int g = 0; g < n; ++g) {
        const mrcnn_conv_desc& d = pr[g].d;
        const long long M = (long long)d.N * d.OH * d.OW;
        long long splits = want;
        const long long max_splits = (M + 127) / 128;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        if (splits > 64) splits = 64;
        const long long chunk = ((M + splits - 1) / splits + 31) / 32 * 32;
        splits = (M + chunk - 1) / chunk;
        pl.splits[g] = (int)splits; pl.chunk[g] = (int)chunk;
        pl.slab_off[g] = pl.bytes;
        if (splits > 1) pl.bytes += (((size_t)splits * d.KH * d.KW * d.Cin * d.Cout * sizeof(float)) + 255) & ~(size_t)255;
    }
    return true;
}

extern "C" size_t mrcnn_conv2d_wgrad_multi_workspace(const mrcnn_wgrad_problem* problems, int n) {
    WgradMultiPlan pl;
    if (!plan_wgrad_multi(problems, n, pl)) return 0;
    return pl.bytes + 256;
}

extern "C" int mrcnn_conv2d_wgrad_multi(const mrcnn_wgrad_problem* problems, int n, float* workspace, size_t workspace_bytes,
                                        void* stream) {
    WgradMultiPlan pl;
    if (!plan_wgrad_multi(problems, n, pl) || mrcnn_force_flat_glds()) return MRCNN_ERR_UNSUPPORTED;
    if (pl.bytes && (!workspace || workspace_bytes < pl.bytes || (reinterpret_cast<uintptr_t>(workspace) & 15))) return MRCNN_ERR_WORKSPACE;
    WgradMultiArgs mp;
    mp.n = n;
    long long blocks = 0, rblocks = 0;
    for (int g = 0; g < n; ++g) {
        const mrcnn_wgrad_problem& q = problems[g];
        const mrcnn_conv_desc* d = &q.d;
        if (!q.x || !q.dy || !q.dw) return MRCNN_ERR_ARG;
        if ((reinterpret_cast<uintptr_t>(q.x) | reinterpret_cast<uintptr_t>(q.dy) | reinterpret_cast<uintptr_t>(q.dw)) & 15)
            return MRCNN_ERR_UNSUPPORTED;
        const long long M = (long long)d->N * d->OH * d->OW;
        WgradArgs& a = mp.a[g];
        a.x = q.x; a.dy = q.dy;
        a.out = pl.splits[g] > 1 ? (float*)((char*)workspace + pl.slab_off[g]) : q.dw;
        a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
        a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.OH = d->OH; a.OW = d->OW;
        a.M = (int)M; a.Ktot = d->KH * d->KW * d->Cin; a.fast = 1; a.splits = pl.splits[g]; a.chunk = pl.chunk[g];
        a.acc = q.accumulate;
        a.d_ohw = make_fastdiv((unsigned)(d->OH * d->OW)); a.d_ow = make_fastdiv((unsigned)d->OW);
        a.table = nullptr;
        const long long xbytes = (long long)d->N * d->H * d->W * d->Cin * 4;
        const long long shift = ((long long)d->pad_t * d->W + d->pad_l) * d->Cin * 4;
        mp.x_shift[g] = (unsigned)shift; mp.x_records[g] = (unsigned)(xbytes + shift);
        mp.dw[g] = q.dw;
        mp.first[g] = (int)blocks;
        blocks += (long long)(a.Ktot / 128) * (a.Cout / 128) * a.splits;
        mp.rfirst[g] = (int)rblocks;
        if (a.splits > 1) rblocks += cdiv64((long long)a.Ktot * a.Cout / 4, 256);
    }
    for (int g = n; g <= WGRAD_MULTI_MAX; ++g) { mp.first[g] = (int)blocks; mp.rfirst[g] = (int)rblocks; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((conv_wgrad_blds_multi_kernel<WGRAD_BP>), dim3((unsigned)blocks), dim3(256), 0, s, mp);
    if (rblocks > 0) hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, mp);
    return mrcnn_launch_status();
}

// w_t[((KH-1-kh)*KW + (KW-1-kw))*Cout + co][ci] = w[((kh*KW + kw)*Cin + ci)][co]
__global__ void flip_transpose_kernel(const float* __restrict__ w, float* wt, int KH, int KW, int Cin, int Cout) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int kh = tap / KW, kw = tap % KW;
    const int tap_t = (KH - 1 - kh) * KW + (KW - 1 - kw);
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < Cin && co < Cout) ? w[((long long)tap * Cin + ci) * Cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int co = co0 + r, ci = ci0 + tx;
        if (co < Cout && ci < Cin) wt[((long long)tap_t * Cout + co) * Cin + ci] = tile[tx][r];
    }
}

extern "C" int mrcnn_weight_flip_transpose(const float* w, float* w_t, int KH, int KW, int Cin, int Cout, void* stream) {
    if (!w || !w_t || KH <= 0 || KW <= 0 || Cin <= 0 || Cout <= 0) return MRCNN_ERR_ARG;
    dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, KH * KW);
    hipLaunchKernelGGL(flip_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, w_t, KH, KW, Cin, Cout);
    return mrcnn_launch_status();
}

// All layers of a flat parameter buffer in one launch: table[l] = {offset (floats, same in w and w_t), KH, KW,
// Cin, Cout, first_tile}; a tile is 32 ci x 32 co of one tap.  The weights change once per optimiser step, so
// the data-gradient operands of every layer are refreshed by one HBM-bound pass instead of one launch per layer.
struct FlipEntry { long long off; int KH, KW, Cin, Cout, first_tile, pad; };

__global__ void flip_transpose_batched_kernel(const float* __restrict__ w, float* wt, const FlipEntry* __restrict__ table, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;                         // last entry with first_tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FlipEntry e = table[lo];
    int t = blockIdx.x - e.first_tile;
    const int cot = (e.Cout + 31) / 32, cit = (e.Cin + 31) / 32;
    const int co0 = (t % cot) * 32; t /= cot;
    const int ci0 = (t % cit) * 32;
    const int tap = t / cit;
    const int kh = tap / e.KW, kw = tap % e.KW;
    const int tap_t = (e.KH - 1 - kh) * e.KW + (e.KW - 1 - kw);
    const float* src = w + e.off;
    float* dst = wt + e.off;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < e.Cin && co < e.Cout) ? src[((long long)tap * e.Cin + ci) * e.Cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < e.Cout && ci < e.Cin) dst[((long long)tap_t * e.Cout + co) * e.Cin + ci] = tile[tx][r];
    }
}

extern "C" int mrcnn_weight_flip_transpose_batched(const float* params, float* params_t, const void* table, int n_layers,
                                                   int total_tiles, void* stream) {
    if (!params || !params_t || !table || n_layers <= 0 || total_tiles <= 0) return MRCNN_ERR_ARG;
    hipLaunchKernelGGL(flip_transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, params,
                       params_t, (const FlipEntry*)table, n_layers);
    return mrcnn_launch_status();
}
