// Winograd F(2x2, 3x3) for the stride-1 "same" 3x3 convolutions of the mask head (float32): Y = A^T [(G g G^T) . (B^T d B)] A per
// 4 x 4 input tile / 2 x 2 output tile, summed over input channels -- 16 multiplications per 4 outputs and channel pair instead
// of 36, i.e. 16 batched GEMMs [tiles x Cin] . [Cin x Cout] in the transform domain (2.25 x fewer MFMA flops than the direct
// form).  Three passes per layer:
//   winograd_input_kernel    x [N, H, W, C]            -> V [16][T][C],   T = N (H/2)(W/2) tiles   (reads x once through L2, writes 4 x its size)
//   16 GEMMs                 V[xi] [T, C] . U[xi] [C, Cout] -> Mt [16][T][Cout]                     (conv_fwd_blds_kernel, batched over xi)
//   winograd_output_kernel   Mt -> out [N, H, W, Cout] with the convolution's epilogue (bias, frozen-BN affine, activation, z)
// plus winograd_weight_kernel U = G g G^T once per weight update.  The transforms are exact additions / halvings in float32
// (B^T, A^T have entries 0, +-1; G has 0, 1, +-1/2), so the result differs from the direct kernel by summation order only.
// H and W even (mask head: 14 x 14).
#include "common.h"
#include <type_traits>

// B^T d B for one 4 x 4 tile held as d[r][c] (float4 = 4 channels per thread)
__device__ __forceinline__ void wino_bt_d_b(const f32x4 d[4][4], f32x4 v[4][4]) {
    f32x4 t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {                // rows: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        t[0][c] = d[0][c] - d[2][c];
        t[1][c] = d[1][c] + d[2][c];
        t[2][c] = d[2][c] - d[1][c];
        t[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                // columns: the same with B
        v[r][0] = t[r][0] - t[r][2];
        v[r][1] = t[r][1] + t[r][2];
        v[r][2] = t[r][2] - t[r][1];
        v[r][3] = t[r][1] - t[r][3];
    }
}

// one thread: one tile x 4 channels.  Tiles are numbered (n, th, tw) row-major: tile t covers outputs (2 th .. 2 th + 1, 2 tw .. + 1)
__global__ __launch_bounds__(256) void winograd_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W,
                                                             int C, long long T, long long Tp) {
    const int c4n = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * c4n) return;
    const long long t = i / c4n;
    const int c = (int)(i - t * c4n) * 4;
    const int tw_n = W >> 1, th_n = H >> 1;
    const int tw = (int)(t % tw_n);
    const long long q = t / tw_n;
    const int th = (int)(q % th_n);
    const int n = (int)(q / th_n);
    const int ih0 = 2 * th - 1, iw0 = 2 * tw - 1;
    f32x4 d[4][4], v[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ih = ih0 + r, iw = iw0 + s;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                val = *(const f32x4*)(x + (((long long)n * H + ih) * W + iw) * C + c);
            d[r][s] = val;
        }
    wino_bt_d_b(d, v);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) *(f32x4*)(V + ((long long)(r * 4 + s) * Tp + t) * C + c) = v[r][s];
}

// U[xi][ci][co] = (G g G^T)[xi] from the HWIO kernel g[3][3][ci][co]; one thread per (ci, co)
__global__ __launch_bounds__(256) void winograd_weight_kernel(const float* __restrict__ g, float* __restrict__ U, int Cin, int Cout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)Cin * Cout;
    if (i >= n) return;
    float w[3][3], t[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) w[a][b] = g[(long long)(a * 3 + b) * n + i];
#pragma unroll
    for (int b = 0; b < 3; ++b) {                // G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
        t[0][b] = w[0][b];
        t[1][b] = 0.5f * (w[0][b] + w[1][b] + w[2][b]);
        t[2][b] = 0.5f * (w[0][b] - w[1][b] + w[2][b]);
        t[3][b] = w[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        U[(long long)(a * 4 + 0) * n + i] = t[a][0];
        U[(long long)(a * 4 + 1) * n + i] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        U[(long long)(a * 4 + 2) * n + i] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        U[(long long)(a * 4 + 3) * n + i] = t[a][2];
    }
}

// A^T m A (2 x 2 outputs of one tile) + epilogue: z = y + bias (stored when asked), out = act(scale z + shift)
__global__ __launch_bounds__(256) void winograd_output_kernel(const float* __restrict__ Mt, float* __restrict__ out, float* __restrict__ z,
                                                              const float* __restrict__ bias, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int N, int H, int W, int C, long long T,
                                                              long long Tp, int act) {
    const int c4n = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * c4n) return;
    const long long t = i / c4n;
    const int c = (int)(i - t * c4n) * 4;
    const int tw_n = W >> 1, th_n = H >> 1;
    const int tw = (int)(t % tw_n);
    const long long q = t / tw_n;
    const int th = (int)(q % th_n);
    const int n = (int)(q / th_n);
    f32x4 m[4][4], s[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) m[r][k] = *(const f32x4*)(Mt + ((long long)(r * 4 + k) * Tp + t) * C + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {                // A^T = [1 1 1 0; 0 1 -1 -1]
        s[0][k] = m[0][k] + m[1][k] + m[2][k];
        s[1][k] = m[1][k] - m[2][k] - m[3][k];
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    const f32x4 bi = bias ? *(const f32x4*)(bias + c) : zero4;
    const f32x4 sc = scale ? *(const f32x4*)(scale + c) : one4, sh = scale ? *(const f32x4*)(shift + c) : zero4;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        f32x4 y[2];
        y[0] = s[a][0] + s[a][1] + s[a][2];
        y[1] = s[a][1] - s[a][2] - s[a][3];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long long addr = (((long long)n * H + 2 * th + a) * W + 2 * tw + b) * C + c;
            f32x4 zv, o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zv[e] = y[b][e] + bi[e];
                float v = sc[e] * zv[e] + sh[e];
                if (act == MRCNN_ACT_RELU) v = fmaxf(v, 0.f);
                o[e] = v;
            }
            if (z) *(f32x4*)(z + addr) = zv;
            *(f32x4*)(out + addr) = o;
        }
    }
}

// The output transform of a DATA gradient fused with the epilogue backward of the layer below (what mrcnn_conv2d_dgrad_ep does
// for the direct kernel): y = A^T m A is d(loss)/d(activated output of the layer below);
//   g = y * act'(out_below);  dz = g * scale_below -> stored;  dbeta += sum g, dgamma += sum g (z_below - mean) rstd, dbias += sum dz.
// A workgroup walks a range of tiles: lane = 4-channel group (C / 4 of them, a power of two <= 256), 256 / (C / 4) tiles in flight;
// channel sums stay in registers, meet in LDS and leave as one atomic per channel and workgroup.
__global__ __launch_bounds__(256) void winograd_output_bwd_kernel(const float* __restrict__ Mt, float* __restrict__ dz_out,
                                                                  const float* __restrict__ below_out, const float* __restrict__ below_z,
                                                                  const float* __restrict__ scale, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* dgamma, float* dbeta, float* dbias,
                                                                  int N, int H, int W, int C, long long T, long long Tp, int act,
                                                                  long long tiles_per_block, int lg) {
    __shared__ float sacc[3 * 4 * 256];
    const int L = 1 << lg, R = 256 >> lg;                       // L = C / 4 lanes, R tiles in flight
    for (int c = threadIdx.x; c < 12 * L; c += 256) sacc[c] = 0.f;
    __syncthreads();
    const int rsub = threadIdx.x >> lg, lane = threadIdx.x & (L - 1);
    const int c = lane * 4;
    const long long t0 = (long long)blockIdx.x * tiles_per_block;
    long long t1 = t0 + tiles_per_block;
    if (t1 > T) t1 = T;
    const int tw_n = W >> 1, th_n = H >> 1;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *(const f32x4*)(scale + c);
    if (dgamma) { mu = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c); }
    f32x4 a_db = {0.f, 0.f, 0.f, 0.f}, a_dg = a_db, a_bias = a_db;
    for (long long t = t0 + rsub; t < t1; t += R) {
        const int tw = (int)(t % tw_n);
        const long long q = t / tw_n;
        const int th = (int)(q % th_n);
        const int n = (int)(q / th_n);
        f32x4 m[4][4], s[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) m[r][k] = *(const f32x4*)(Mt + ((long long)(r * 4 + k) * Tp + t) * C + c);
        f32x4 oo[2][2], zz[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long long addr = (((long long)n * H + 2 * th + a) * W + 2 * tw + b) * C + c;
                if (act == MRCNN_ACT_RELU) oo[a][b] = *(const f32x4*)(below_out + addr);
                if (dgamma) zz[a][b] = *(const f32x4*)(below_z + addr);
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[0][k] = m[0][k] + m[1][k] + m[2][k];
            s[1][k] = m[1][k] - m[2][k] - m[3][k];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f32x4 y[2];
            y[0] = s[a][0] + s[a][1] + s[a][2];
            y[1] = s[a][1] - s[a][2] - s[a][3];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long long addr = (((long long)n * H + 2 * th + a) * W + 2 * tw + b) * C + c;
                f32x4 g = y[b], dz;
                if (act == MRCNN_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = oo[a][b][e] > 0.f ? g[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = g[e] * sc[e];
                *(f32x4*)(dz_out + addr) = dz;
                if (dgamma) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a_dg[e] += g[e] * (zz[a][b][e] - mu[e]) * rs[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { a_db[e] += g[e]; a_bias[e] += dz[e]; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (dbeta || dgamma) atomicAdd(&sacc[lane * 4 + k], a_db[k]);
        if (dgamma) atomicAdd(&sacc[4 * L + lane * 4 + k], a_dg[k]);
        if (dbias) atomicAdd(&sacc[8 * L + lane * 4 + k], a_bias[k]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 4 * L; j += 256) {
        if (dbeta) atomicAdd(&dbeta[j], sacc[j]);
        if (dgamma) atomicAdd(&dgamma[j], sacc[4 * L + j]);
        if (dbias) atomicAdd(&dbias[j], sacc[8 * L + j]);
    }
}

// Weight gradient through the same domain: dU[xi] = V[xi]^T . dM[xi] (16 GEMMs contracting over the tiles: 1 x 1 weight
// gradients for the float32 weight-gradient kernel), with V = B^T d B the forward's input transform (kept from the forward
// pass) and dM = A dy A^T the ADJOINT of the output transform, A = [1 0; 1 1; 1 -1; 0 -1]; then dW = G^T dU G.
__global__ __launch_bounds__(256) void winograd_dy_kernel(const float* __restrict__ dy, float* __restrict__ dM, int N, int H, int W, int C,
                                                          long long T, long long Tp) {
    const int c4n = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * c4n) return;
    const long long t = i / c4n;
    const int c = (int)(i - t * c4n) * 4;
    const int tw_n = W >> 1, th_n = H >> 1;
    const int tw = (int)(t % tw_n);
    const long long q = t / tw_n;
    const int th = (int)(q % th_n);
    const int n = (int)(q / th_n);
    f32x4 d[2][2], u[4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) d[a][b] = *(const f32x4*)(dy + (((long long)n * H + 2 * th + a) * W + 2 * tw + b) * C + c);
#pragma unroll
    for (int b = 0; b < 2; ++b) {                // rows: A d
        u[0][b] = d[0][b];
        u[1][b] = d[0][b] + d[1][b];
        u[2][b] = d[0][b] - d[1][b];
        u[3][b] = -d[1][b];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                // columns: (A d) A^T
        const f32x4 m0 = u[r][0], m1 = u[r][0] + u[r][1], m2 = u[r][0] - u[r][1], m3 = -u[r][1];
        *(f32x4*)(dM + ((long long)(r * 4 + 0) * Tp + t) * C + c) = m0;
        *(f32x4*)(dM + ((long long)(r * 4 + 1) * Tp + t) * C + c) = m1;
        *(f32x4*)(dM + ((long long)(r * 4 + 2) * Tp + t) * C + c) = m2;
        *(f32x4*)(dM + ((long long)(r * 4 + 3) * Tp + t) * C + c) = m3;
    }
}

// dW [3][3][ci][co] (= or +=) G^T dU G, dU [16][ci][co]; one thread per (ci, co)
__global__ __launch_bounds__(256) void winograd_dw_kernel(const float* __restrict__ dU, float* dW, int Cin, int Cout, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)Cin * Cout;
    if (i >= n) return;
    float u[4][4], t[3][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) u[a][b] = dU[(long long)(a * 4 + b) * n + i];
#pragma unroll
    for (int b = 0; b < 4; ++b) {                // G^T = [1 1/2 1/2 0; 0 1/2 -1/2 0; 0 1/2 1/2 1]
        t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
        t[1][b] = 0.5f * (u[1][b] - u[2][b]);
        t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float w0 = t[a][0] + 0.5f * (t[a][1] + t[a][2]), w1 = 0.5f * (t[a][1] - t[a][2]), w2 = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
        float* o = dW + (long long)(a * 3) * n + i;
        o[0] = accumulate ? o[0] + w0 : w0;
        o[n] = accumulate ? o[n] + w1 : w1;
        o[2 * n] = accumulate ? o[2 * n] + w2 : w2;
    }
}

// ---- the 16 transform-domain GEMMs, persistent form ----------------------------------------------------------------------------
// V [nb][rows][K] . U [nb][K][N] -> Mt [nb][rows][N] with the tile, LDS image and MFMA loop of conv_fwd_blds_kernel (128 x 128
// tile, K-step 16, both operands by buffer-addressed LDS-DMA, A chunks XOR-swizzled on the source side), but workgroups that
// walk tiles blockIdx.x, + gridDim.x, ...: K is only 256 here (16 K-steps per tile), so the one-tile-per-workgroup kernel spends
// ~26 % of a launch in phases where every resident workgroup of a CU is storing its tile or waiting for its first stage at the
// same time (tools/gemm_k_probe.py: 0.15 ms fixed + 1.5 us per unit of K over 4.9 rounds).  Here the first stage of the NEXT
// tile is on its way before the stores of the current tile are issued, and the workgroups of a CU drift apart.
typedef __attribute__((address_space(3))) void* wino_lds_ptr;
#define WINO_OOB 0xFFFFFFF0u

template <int TN>      // 2: 128 x 128 tile (5 workgroups per CU); 4: 128 x 256 tile -- N = 256 whole: every V row block is fetched once
__global__ __launch_bounds__(256, 2) void winograd_gemm_kernel(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ Mt,
                                                               int rows, int K, int N, int total_tiles, unsigned v_records) {
    constexpr int BM = 128, BN = 64 * TN, BK = 16, TM = 2;
    constexpr int AF = BM * BK, BF = BK * BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * (AF + BF)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = N / BN;
    const int nk = K / BK;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, v_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, 0x7FFFFFF0u, 0x00020000);

    // per tile: lane offsets of the two A pieces (16 rows x 64 bytes each) and the B pieces (2 k-rows x 512 bytes each)
    constexpr int NBP = BF / 256 / 4;                     // 1 KiB B pieces per wave and stage (2 or 4)
    unsigned a_voff[2], b_voff[4];                        // fixed bound: a template-dependent one captured by the lambdas below makes
                                                          // hipcc drop the kernel's host stub (build.py checks for that)
    int m0 = 0, n0 = 0;
    auto setup = [&](int tile) {
        const int mtile = tile / ntiles, ntile = tile - mtile * ntiles;
        m0 = mtile * BM; n0 = ntile * BN;
        const unsigned wbase = (unsigned)(((long long)(m0 / rows) * K * N + n0) * 4);      // the tile's weight matrix (U [nb][K][N], < 2 GiB)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int r = (wave + jj * 4) * 16 + (lane >> 2);
            const int cl = (lane & 3) ^ ((r >> 2) & 3);
            a_voff[jj] = (unsigned)(((long long)(m0 + r) * K + cl * 4) * 4);
        }
        // B [16 k][BN]: a 1 KiB piece is 256 consecutive floats of the tile: piece pc -> k row (pc * 256) / BN, columns (pc * 256) % BN ..
#pragma unroll
        for (int jj = 0; jj < NBP; ++jj) {
            const int f = (wave + jj * 4) * 256 + lane * 4;       // float index inside the [16][BN] tile
            b_voff[jj] = wbase + (unsigned)(((f / BN) * N + (f % BN)) * 4);
        }
    };
    int k0 = 0;
    auto stage = [&](float* ab) {
        float* bb = ab + AF;
        const unsigned soff_a = (unsigned)(k0 * 4), soff_b = (unsigned)(k0 * N * 4);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (wino_lds_ptr)(ab + (wave + jj * 4) * 256), 16, a_voff[jj], soff_a, 0, 0);
#pragma unroll
        for (int jj = 0; jj < NBP; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (wino_lds_ptr)(bb + (wave + jj * 4) * 256), 16, b_voff[jj], soff_b, 0, 0);
        k0 += BK;
    };
    f32x16 acc[TM][TN];
    const int li = lane & 31, lh = lane >> 5;
    const int row0 = wm * 64 + li;
    const float* a_rd0 = lds + row0 * BK + (((2 * lh + 0) ^ ((row0 >> 2) & 3)) << 2);
    const float* a_rd1 = lds + row0 * BK + (((2 * lh + 1) ^ ((row0 >> 2) & 3)) << 2);
    const float* b_rd = lds + AF + lh * 8 * BN + wn * (32 * TN) + li;
    auto compute = [&](auto curc) {
        constexpr int BO = decltype(curc)::value * (AF + BF);
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
                    for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][e], bv[b], acc[a][b], 0, 0, 0);
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    int tile = blockIdx.x;
    setup(tile);
    stage(lds);
    for (;;) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        __syncthreads();
        for (int ks = 0; ks < nk; ks += 2) {
            if (ks + 1 < nk) stage(lds + (AF + BF));
            compute(I0{});
            __syncthreads();
            if (ks + 1 < nk) {
                if (ks + 2 < nk) stage(lds);
                compute(I1{});
                __syncthreads();
            }
        }
        const int mw0 = m0 + wm * 64 + 4 * lh, nw0 = n0 + wn * (32 * TN) + li;   // the tile the accumulators belong to
        const int next = tile + (int)gridDim.x;
        const bool more = next < total_tiles;
        if (more) { k0 = 0; setup(next); stage(lds); }                         // its first stage travels under this tile's stores
        const int Mtot = (int)(v_records / (unsigned)(K * 4));
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw0 + a * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < Mtot) Mt[(long long)m * N + nw0 + b * 32] = acc[a][b][r];
                }
        if (!more) break;
        tile = next;
    }
}

static inline long long wino_tiles(int N, int H, int W) { return (long long)N * (H >> 1) * (W >> 1); }
static inline long long wino_rows(long long T) { return (T + 127) / 128 * 128; }      // rows per transform-domain matrix: whole 128-row tiles

static int wino_shape_ok(int N, int H, int W, int C) { return N > 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) && C > 0 && !(C & 3); }

/* floats of V (input transform) or Mt (transform-domain product) for a layer: 16 x rows x C */
extern "C" size_t mrcnn_winograd_buffer_floats(int N, int H, int W, int C) {
    if (!wino_shape_ok(N, H, W, C)) return 0;
    return (size_t)16 * (size_t)wino_rows(wino_tiles(N, H, W)) * (size_t)C;
}

extern "C" int mrcnn_winograd_input(const float* x, float* V, int N, int H, int W, int C, void* stream) {
    if (!x || !V || !wino_shape_ok(N, H, W, C)) return MRCNN_ERR_ARG;
    const long long T = wino_tiles(N, H, W);
    hipLaunchKernelGGL(winograd_input_kernel, dim3((unsigned)cdiv64(T * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream, x, V, N, H, W, C, T,
                       wino_rows(T));
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_weights(const float* g, float* U, int Cin, int Cout, void* stream) {
    if (!g || !U || Cin <= 0 || Cout <= 0) return MRCNN_ERR_ARG;
    hipLaunchKernelGGL(winograd_weight_kernel, dim3((unsigned)cdiv64((long long)Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream, g, U, Cin, Cout);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_output(const float* Mt, float* out, float* z, const float* bias, const float* scale, const float* shift,
                                     int N, int H, int W, int C, int act, void* stream) {
    if (!Mt || !out || !wino_shape_ok(N, H, W, C) || (scale && !shift)) return MRCNN_ERR_ARG;
    if (act != MRCNN_ACT_NONE && act != MRCNN_ACT_RELU) return MRCNN_ERR_UNSUPPORTED;
    const long long T = wino_tiles(N, H, W);
    hipLaunchKernelGGL(winograd_output_kernel, dim3((unsigned)cdiv64(T * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream, Mt, out, z, bias,
                       scale, shift, N, H, W, C, T, wino_rows(T), act);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_output_bwd(const float* Mt, float* dz_below, const float* below_out, const float* below_z, const float* scale,
                                         const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H, int W,
                                         int C, int act, void* stream) {
    if (!Mt || !dz_below || !wino_shape_ok(N, H, W, C)) return MRCNN_ERR_ARG;
    if ((act != MRCNN_ACT_NONE && act != MRCNN_ACT_RELU) || (act == MRCNN_ACT_RELU && !below_out)) return MRCNN_ERR_ARG;
    if (dgamma && (!below_z || !mean || !rstd)) return MRCNN_ERR_ARG;
    const int c4n = C >> 2;
    if (c4n > 256 || (c4n & (c4n - 1))) return MRCNN_ERR_UNSUPPORTED;
    int lg = 0;
    while ((1 << lg) < c4n) ++lg;
    const long long T = wino_tiles(N, H, W);
    const long long R = 256 >> lg;
    long long per = (T + 2047) / 2048;                          // ~2048 workgroups; every one a whole number of passes
    per = (per + R - 1) / R * R;
    hipLaunchKernelGGL(winograd_output_bwd_kernel, dim3((unsigned)cdiv64(T, per)), dim3(256), 0, (hipStream_t)stream, Mt, dz_below, below_out,
                       below_z, scale, mean, rstd, dgamma, dbeta, dbias, N, H, W, C, T, wino_rows(T), act, per, lg);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_dy(const float* dy, float* dM, int N, int H, int W, int C, void* stream) {
    if (!dy || !dM || !wino_shape_ok(N, H, W, C)) return MRCNN_ERR_ARG;
    const long long T = wino_tiles(N, H, W);
    hipLaunchKernelGGL(winograd_dy_kernel, dim3((unsigned)cdiv64(T * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream, dy, dM, N, H, W, C, T,
                       wino_rows(T));
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_dw(const float* dU, float* dw_hwio, int Cin, int Cout, int accumulate, void* stream) {
    if (!dU || !dw_hwio || Cin <= 0 || Cout <= 0) return MRCNN_ERR_ARG;
    hipLaunchKernelGGL(winograd_dw_kernel, dim3((unsigned)cdiv64((long long)Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream, dU, dw_hwio,
                       Cin, Cout, accumulate);
    return mrcnn_launch_status();
}

/* the persistent form of mrcnn_gemm_batched_f32 (same arguments and result; K % 16 == 0, N % 128 == 0, rows % 128 == 0) */
extern "C" int mrcnn_winograd_gemm(const float* V, const float* U, float* Mt, int nb, int rows, int K, int N, void* stream) {
    if (!V || !U || !Mt || nb <= 0 || rows <= 0 || rows % 128 || K <= 0 || K % 16 || N <= 0 || N % 128) return MRCNN_ERR_ARG;
    const long long M = (long long)nb * rows;
    if (M * K * 4 >= 0x7FFFFFF0LL || (long long)nb * K * N * 4 >= 0x7FFFFFF0LL || M * N >= (1LL << 40)) return MRCNN_ERR_UNSUPPORTED;
    // OFF by default: alone the wide tile is 4 % faster (1.69 -> 1.62 ms, 0.79 -> 0.83 of the matrix peak: every V row block
    // is fetched once), but inside the step it is slower (44.0 -> 44.9 ms): at 166 VGPRs / 48 KiB it leaves less room for the
    // other stream's kernels on a CU
    static const int wide = getenv("MRCNN_WINOGRAD_GEMM_WIDE") ? atoi(getenv("MRCNN_WINOGRAD_GEMM_WIDE")) : 0;
    if (wide && N % 256 == 0) {                                 // 128 x 256 tiles: 48 KiB LDS, 3 workgroups per CU
        const long long tiles = (M / 128) * (N / 256);
        const long long slots = 3LL * mrcnn_num_cus();
        const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
        hipLaunchKernelGGL(winograd_gemm_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, V, U, Mt, rows, K, N, (int)tiles,
                           (unsigned)(M * K * 4));
        return mrcnn_launch_status();
    }
    const long long tiles = (M / 128) * (N / 128);
    const long long slots = 5LL * mrcnn_num_cus();
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL(winograd_gemm_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, V, U, Mt, rows, K, N, (int)tiles,
                       (unsigned)(M * K * 4));
    return mrcnn_launch_status();
}
