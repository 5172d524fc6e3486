// Winograd F(2x2, 3x3) / F(4x4, 3x3) for the stride-1 "same" 3x3 convolutions of the mask head (float32):
// Y = A^T [(G g G^T) . (B^T d B)] A per 4 x 4 input tile / 2 x 2 output tile, summed over input channels -- 16 multiplications per 4
// outputs and channel pair instead of 36, i.e. 16 batched GEMMs [tiles x Cin] . [Cin x Cout] in the transform domain (2.25 x fewer
// MFMA flops than the direct form); with 6 x 6 / 4 x 4 tiles 36 GEMMs over a quarter of the tiles (4 x fewer).  Three passes per layer:
//   winograd_input_kernel    x [N, H, W, C]            -> V [16][T][C],   T = N (H/2)(W/2) tiles   (reads x once through L2, writes 4 x its size)
//   16 GEMMs                 V[xi] [T, C] . U[xi] [C, Cout] -> Mt [16][T][Cout]                     (conv_fwd_blds_kernel, batched over xi)
//   winograd_output_kernel   Mt -> out [N, H, W, Cout] with the convolution's epilogue (bias, frozen-BN affine, activation, z)
// plus winograd_weight_kernel U = G g G^T once per weight update.  For the 2 x 2 tile the transforms are exact additions / halvings
// in float32 (B^T, A^T have entries 0, +-1; G has 0, 1, +-1/2), so the result differs from the direct kernel by summation order
// only, and H and W must be even (mask head: 14 x 14); the 4 x 4 tile takes any extent (its last tiles hang over the edge).
#include "common.h"
#include <type_traits>
#include <string.h>

// ---- the 1-D transforms, m = OT outputs per tile and dimension (IT = OT + 2 inputs) -------------------------------------------
// OT = 2: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], A^T = [1 1 1 0; 0 1 -1 -1].
// OT = 4 (F(4x4, 3x3), interpolation points 0, +-1, +-2, inf): 36 multiplications per 16 outputs (4 x fewer than direct, 1.78 x
// fewer than OT = 2 -- 1.36 x on 14 x 14 maps, which it covers with 4 x 4 tiles of which the last row / column is half used);
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// its constants up to 8 cost about a decimal digit: ~7e-6 of the output range against ~5e-7 for OT = 2 (float32, K = 256).
template <int CV> struct WinoVec;
template <> struct WinoVec<4> { typedef f32x4 type; };
template <> struct WinoVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

template <int OT, typename V> __device__ __forceinline__ void wino_bt(const V (&d)[OT + 2], V (&t)[OT + 2]) {
    if constexpr (OT == 2) {
        t[0] = d[0] - d[2];
        t[1] = d[1] + d[2];
        t[2] = d[2] - d[1];
        t[3] = d[1] - d[3];
    } else {
        const V a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
        t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
        t[1] = a + b;
        t[2] = a - b;
        t[3] = c + e;
        t[4] = c - e;
        t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
    }
}
template <int OT, typename V> __device__ __forceinline__ void wino_at(const V (&m)[OT + 2], V (&y)[OT]) {
    if constexpr (OT == 2) {
        y[0] = m[0] + m[1] + m[2];
        y[1] = m[1] - m[2] - m[3];
    } else {
        const V p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
        y[0] = m[0] + p + r;
        y[1] = q + 2.f * s;
        y[2] = p + 4.f * r;
        y[3] = q + 8.f * s + m[5];
    }
}
// the adjoint of wino_at: A y
template <int OT, typename V> __device__ __forceinline__ void wino_a(const V (&y)[OT], V (&m)[OT + 2]) {
    if constexpr (OT == 2) {
        m[0] = y[0];
        m[1] = y[0] + y[1];
        m[2] = y[0] - y[1];
        m[3] = -y[1];
    } else {
        const V e = y[0] + y[2], o = y[1] + y[3], e4 = y[0] + 4.f * y[2], o4 = 2.f * y[1] + 8.f * y[3];
        m[0] = y[0];
        m[1] = e + o;
        m[2] = e - o;
        m[3] = e4 + o4;
        m[4] = e4 - o4;
        m[5] = y[3];
    }
}
template <int OT> __device__ __forceinline__ void wino_g(const float (&g)[3], float (&u)[OT + 2]) {
    if constexpr (OT == 2) {
        u[0] = g[0];
        u[1] = 0.5f * (g[0] + g[1] + g[2]);
        u[2] = 0.5f * (g[0] - g[1] + g[2]);
        u[3] = g[2];
    } else {
        const float e = g[0] + g[2];
        u[0] = 0.25f * g[0];
        u[1] = (-1.f / 6.f) * (e + g[1]);
        u[2] = (-1.f / 6.f) * (e - g[1]);
        const float f = (1.f / 24.f) * g[0] + (1.f / 6.f) * g[2], h = (1.f / 12.f) * g[1];
        u[3] = f + h;
        u[4] = f - h;
        u[5] = g[2];
    }
}
// the adjoint of wino_g: G^T u
template <int OT> __device__ __forceinline__ void wino_gt(const float (&u)[OT + 2], float (&w)[3]) {
    if constexpr (OT == 2) {
        w[0] = u[0] + 0.5f * (u[1] + u[2]);
        w[1] = 0.5f * (u[1] - u[2]);
        w[2] = 0.5f * (u[1] + u[2]) + u[3];
    } else {
        const float p = u[1] + u[2], q = u[3] + u[4];
        w[0] = 0.25f * u[0] - (1.f / 6.f) * p + (1.f / 24.f) * q;
        w[1] = (1.f / 6.f) * (u[2] - u[1]) + (1.f / 12.f) * (u[3] - u[4]);
        w[2] = (1.f / 6.f) * (q - p) + u[5];
    }
}

// A GROUP of tiles: th_n x tw_n tiles per map of OTH x OTW outputs each, the first one at output (oh0, ow0).  A uniform tiling is
// one group at (0, 0); a 14 x 14 map without overhang is four: 3 x 3 tiles of 4 x 4, 3 x 1 of 4 x 2 (columns 12..13), 1 x 3 of
// 2 x 4 (rows 12..13) and one 2 x 2 -- 484 multiplications per map and channel pair instead of 576.
// tile t of [0, N * th_n * tw_n) -> (n, th, tw), row-major
struct WinoGrp { int th_n, tw_n, oh0, ow0; };
struct WinoTile { int n, th, tw; };
__device__ __forceinline__ WinoTile wino_tile(long long t, int th_n, int tw_n) {
    WinoTile r;
    r.tw = (int)(t % tw_n);
    const long long q = t / tw_n;
    r.th = (int)(q % th_n);
    r.n = (int)(q / th_n);
    return r;
}

// B^T d B: one thread = one tile x CV channels.  V [ITH * ITW][Tp][C]
template <int OTH, int OTW, int CV>
__global__ __launch_bounds__(256) void winograd_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W,
                                                             int C, long long T, long long Tp, const WinoGrp g) {
    typedef typename WinoVec<CV>::type vec;
    constexpr int ITH = OTH + 2, ITW = OTW + 2;
    const int cn = C / CV;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * cn) return;
    const long long t = i / cn;
    const int c = (int)(i - t * cn) * CV;
    const WinoTile q = wino_tile(t, g.th_n, g.tw_n);
    const int ih0 = g.oh0 + OTH * q.th - 1, iw0 = g.ow0 + OTW * q.tw - 1;
    vec tt[ITH][ITW];
#pragma unroll
    for (int s = 0; s < ITW; ++s) {              // columns of the tile: B^T applied down the rows
        vec col[ITH], o[ITH];
#pragma unroll
        for (int r = 0; r < ITH; ++r) {
            const int ih = ih0 + r, iw = iw0 + s;
            vec val = {};
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) val = *(const vec*)(x + (((long long)q.n * H + ih) * W + iw) * C + c);
            col[r] = val;
        }
        wino_bt<OTH>(col, o);
#pragma unroll
        for (int r = 0; r < ITH; ++r) tt[r][s] = o[r];
    }
#pragma unroll
    for (int r = 0; r < ITH; ++r) {              // then along each row
        vec v[ITW];
        wino_bt<OTW>(tt[r], v);
#pragma unroll
        for (int s = 0; s < ITW; ++s) *(vec*)(V + ((long long)(r * ITW + s) * Tp + t) * C + c) = v[s];
    }
}

// U[xi][ci][co] = (G_h g G_w^T)[xi] from the HWIO kernel g[3][3][ci][co]; one thread per (ci, co)
template <int OTH, int OTW>
__global__ __launch_bounds__(256) void winograd_weight_kernel(const float* __restrict__ g, float* __restrict__ U, int Cin, int Cout) {
    constexpr int ITH = OTH + 2, ITW = OTW + 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)Cin * Cout;
    if (i >= n) return;
    float t[ITH][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float col[3], o[ITH];
#pragma unroll
        for (int a = 0; a < 3; ++a) col[a] = g[(long long)(a * 3 + b) * n + i];
        wino_g<OTH>(col, o);
#pragma unroll
        for (int a = 0; a < ITH; ++a) t[a][b] = o[a];
    }
#pragma unroll
    for (int a = 0; a < ITH; ++a) {
        float u[ITW];
        wino_g<OTW>(t[a], u);
#pragma unroll
        for (int b = 0; b < ITW; ++b) U[(long long)(a * ITW + b) * n + i] = u[b];
    }
}

// A^T m A (OTH x OTW outputs of one tile) + epilogue: z = y + bias (stored when asked), out = act(scale z + shift)
// wino_output_tile: tile t x CV channels from c on; shared by the stand-alone kernel and by the GEMM's fused epilogue
template <int OTH, int OTW, int CV>
__device__ __forceinline__ void wino_output_tile(const float* __restrict__ Mt, float* __restrict__ out, float* __restrict__ z,
                                                 const typename WinoVec<CV>::type bi, const typename WinoVec<CV>::type sc,
                                                 const typename WinoVec<CV>::type sh, int H, int W, int C, long long Tp, int act,
                                                 const WinoGrp& g, long long t, int c) {
    typedef typename WinoVec<CV>::type vec;
    constexpr int ITH = OTH + 2, ITW = OTW + 2;
    const WinoTile q = wino_tile(t, g.th_n, g.tw_n);
    vec s[OTH][ITW];
#pragma unroll
    for (int k = 0; k < ITW; ++k) {              // A^T down the rows of every column k
        vec col[ITH], o[OTH];
#pragma unroll
        for (int r = 0; r < ITH; ++r) col[r] = *(const vec*)(Mt + ((long long)(r * ITW + k) * Tp + t) * C + c);
        wino_at<OTH>(col, o);
#pragma unroll
        for (int a = 0; a < OTH; ++a) s[a][k] = o[a];
    }
#pragma unroll
    for (int a = 0; a < OTH; ++a) {
        vec y[OTW];
        wino_at<OTW>(s[a], y);
        const int oh = g.oh0 + OTH * q.th + a;
        if (oh >= H) continue;
#pragma unroll
        for (int b = 0; b < OTW; ++b) {
            const int ow = g.ow0 + OTW * q.tw + b;
            if (ow >= W) continue;
            const long long addr = (((long long)q.n * H + oh) * W + ow) * C + c;
            vec zv, o;
#pragma unroll
            for (int e = 0; e < CV; ++e) {
                zv[e] = y[b][e] + bi[e];
                float v = sc[e] * zv[e] + sh[e];
                if (act == MRCNN_ACT_RELU) v = fmaxf(v, 0.f);
                o[e] = v;
            }
            if (z) *(vec*)(z + addr) = zv;
            *(vec*)(out + addr) = o;
        }
    }
}

template <int OTH, int OTW, int CV>
__global__ __launch_bounds__(256) void winograd_output_kernel(const float* __restrict__ Mt, float* __restrict__ out, float* __restrict__ z,
                                                              const float* __restrict__ bias, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int N, int H, int W, int C, long long T,
                                                              long long Tp, int act, const WinoGrp g) {
    typedef typename WinoVec<CV>::type vec;
    const int cn = C / CV;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * cn) return;
    const long long t = i / cn;
    const int c = (int)(i - t * cn) * CV;
    vec bi = {}, sc, sh = {};
#pragma unroll
    for (int e = 0; e < CV; ++e) sc[e] = 1.f;
    if (bias) bi = *(const vec*)(bias + c);
    if (scale) { sc = *(const vec*)(scale + c); sh = *(const vec*)(shift + c); }
    wino_output_tile<OTH, OTW, CV>(Mt, out, z, bi, sc, sh, H, W, C, Tp, act, g, t, c);
}

// The output transform of a DATA gradient fused with the epilogue backward of the layer below (what mrcnn_conv2d_dgrad_ep does
// for the direct kernel): y = A^T m A is d(loss)/d(activated output of the layer below);
//   g = y * act'(out_below);  dz = g * scale_below -> stored;  dbeta += sum g, dgamma += sum g (z_below - mean) rstd, dbias += sum dz.
// A workgroup walks a range of tiles: lane = CV-channel group (C / CV of them, a power of two <= 256), 256 / (C / CV) tiles in
// flight; channel sums stay in registers, meet in LDS and leave as one atomic per channel and workgroup.
// one tile x CV channels of the data-gradient output transform + epilogue backward; the channel sums go to the caller's registers
template <int OTH, int OTW, int CV>
__device__ __forceinline__ void wino_output_bwd_tile(const float* __restrict__ Mt, float* __restrict__ dz_out, const float* __restrict__ below_out,
                                                     const float* __restrict__ below_z, const typename WinoVec<CV>::type sc,
                                                     const typename WinoVec<CV>::type mu, const typename WinoVec<CV>::type rs,
                                                     const typename WinoVec<CV>::type fsh, bool zmask, bool want_dg, int H, int W, int C,
                                                     long long Tp, int act, const WinoGrp& g, long long t, int c,
                                                     typename WinoVec<CV>::type& a_db, typename WinoVec<CV>::type& a_dg,
                                                     typename WinoVec<CV>::type& a_bias) {
    typedef typename WinoVec<CV>::type vec;
    constexpr int ITH = OTH + 2, ITW = OTW + 2;
    const WinoTile q = wino_tile(t, g.th_n, g.tw_n);
    vec s[OTH][ITW];
#pragma unroll
    for (int k = 0; k < ITW; ++k) {
        vec col[ITH], o[OTH];
#pragma unroll
        for (int r = 0; r < ITH; ++r) col[r] = *(const vec*)(Mt + ((long long)(r * ITW + k) * Tp + t) * C + c);
        wino_at<OTH>(col, o);
#pragma unroll
        for (int a = 0; a < OTH; ++a) s[a][k] = o[a];
    }
#pragma unroll
    for (int a = 0; a < OTH; ++a) {
        vec y[OTW];
        wino_at<OTW>(s[a], y);
        const int oh = g.oh0 + OTH * q.th + a;
        if (oh >= H) continue;
#pragma unroll
        for (int b = 0; b < OTW; ++b) {
            const int ow = g.ow0 + OTW * q.tw + b;
            if (ow >= W) continue;
            const long long addr = (((long long)q.n * H + oh) * W + ow) * C + c;
            vec gg = y[b], dz, zz = {};
            if (want_dg || zmask) zz = *(const vec*)(below_z + addr);
            if (act == MRCNN_ACT_RELU) {
                if (zmask) {
#pragma unroll
                    for (int e = 0; e < CV; ++e) {
                        const float v = sc[e] * zz[e] + fsh[e];
                        gg[e] = v > 0.f ? gg[e] : 0.f;
                    }
                } else {
                    const vec oo = *(const vec*)(below_out + addr);
#pragma unroll
                    for (int e = 0; e < CV; ++e) gg[e] = oo[e] > 0.f ? gg[e] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < CV; ++e) dz[e] = gg[e] * sc[e];
            *(vec*)(dz_out + addr) = dz;
            if (want_dg) {
#pragma unroll
                for (int e = 0; e < CV; ++e) a_dg[e] += gg[e] * (zz[e] - mu[e]) * rs[e];
            }
#pragma unroll
            for (int e = 0; e < CV; ++e) { a_db[e] += gg[e]; a_bias[e] += dz[e]; }
        }
    }
}

template <int OTH, int OTW, int CV>
__global__ __launch_bounds__(256) void winograd_output_bwd_kernel(const float* __restrict__ Mt, float* __restrict__ dz_out,
                                                                  const float* __restrict__ below_out, const float* __restrict__ below_z,
                                                                  const float* __restrict__ scale, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* dgamma, float* dbeta, float* dbias,
                                                                  int N, int H, int W, int C, long long T, long long Tp, int act,
                                                                  long long tiles_per_block, int lg, const WinoGrp g,
                                                                  const float* __restrict__ fwd_shift) {
    typedef typename WinoVec<CV>::type vec;
    constexpr int ITH = OTH + 2, ITW = OTW + 2;
    __shared__ float sacc[3 * 4 * 256];
    const int L = 1 << lg, R = 256 >> lg;                       // L = C / CV lanes, R tiles in flight
    for (int c = threadIdx.x; c < 3 * CV * L; c += 256) sacc[c] = 0.f;
    __syncthreads();
    const int rsub = threadIdx.x >> lg, lane = threadIdx.x & (L - 1);
    const int c = lane * CV;
    const long long t0 = (long long)blockIdx.x * tiles_per_block;
    long long t1 = t0 + tiles_per_block;
    if (t1 > T) t1 = T;
    vec sc, mu = {}, rs = {}, fsh = {};
#pragma unroll
    for (int e = 0; e < CV; ++e) sc[e] = 1.f;
    if (scale) sc = *(const vec*)(scale + c);
    if (dgamma) { mu = *(const vec*)(mean + c); rs = *(const vec*)(rstd + c); }
    // fwd_shift: the ReLU mask is taken from the stored pre-BN value -- out > 0 <=> scale * z + shift > 0, the very expression
    // (operation for operation, no contraction) the forward epilogue fed to max(., 0) -- so the activated output of the layer
    // below is not read at all: one of the pass's four tensor streams (0.41 of 2.25 GB per mask-head layer) gone, same bits
    const bool zmask = fwd_shift != nullptr;
    if (zmask) fsh = *(const vec*)(fwd_shift + c);
    vec a_db = {}, a_dg = {}, a_bias = {};
    for (long long t = t0 + rsub; t < t1; t += R)
        wino_output_bwd_tile<OTH, OTW, CV>(Mt, dz_out, below_out, below_z, sc, mu, rs, fsh, zmask, dgamma != nullptr, H, W, C, Tp, act, g, t, c,
                                           a_db, a_dg, a_bias);
#pragma unroll
    for (int k = 0; k < CV; ++k) {
        if (dbeta || dgamma) atomicAdd(&sacc[lane * CV + k], a_db[k]);
        if (dgamma) atomicAdd(&sacc[CV * L + lane * CV + k], a_dg[k]);
        if (dbias) atomicAdd(&sacc[2 * CV * L + lane * CV + k], a_bias[k]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < CV * L; j += 256) {
        if (dbeta) atomicAdd(&dbeta[j], sacc[j]);
        if (dgamma) atomicAdd(&dgamma[j], sacc[CV * L + j]);
        if (dbias) atomicAdd(&dbias[j], sacc[2 * CV * L + j]);
    }
}

// Weight gradient through the same domain: dU[xi] = V[xi]^T . dM[xi] (ITH * ITW GEMMs contracting over the tiles: 1 x 1 weight
// gradients for the float32 weight-gradient kernel), with V = B^T d B the forward's input transform (kept from the forward
// pass) and dM = A dy A^T the ADJOINT of the output transform (outputs past the map's edge count as zero); then dW = G^T dU G.
template <int OTH, int OTW, int CV>
__global__ __launch_bounds__(256) void winograd_dy_kernel(const float* __restrict__ dy, float* __restrict__ dM, int N, int H, int W, int C,
                                                          long long T, long long Tp, const WinoGrp g) {
    typedef typename WinoVec<CV>::type vec;
    constexpr int ITH = OTH + 2, ITW = OTW + 2;
    const int cn = C / CV;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * cn) return;
    const long long t = i / cn;
    const int c = (int)(i - t * cn) * CV;
    const WinoTile q = wino_tile(t, g.th_n, g.tw_n);
    vec u[ITH][OTW];
#pragma unroll
    for (int b = 0; b < OTW; ++b) {              // A down the rows of every output column b
        vec col[OTH], o[ITH];
#pragma unroll
        for (int a = 0; a < OTH; ++a) {
            const int oh = g.oh0 + OTH * q.th + a, ow = g.ow0 + OTW * q.tw + b;
            vec val = {};
            if (oh < H && ow < W) val = *(const vec*)(dy + (((long long)q.n * H + oh) * W + ow) * C + c);
            col[a] = val;
        }
        wino_a<OTH>(col, o);
#pragma unroll
        for (int r = 0; r < ITH; ++r) u[r][b] = o[r];
    }
#pragma unroll
    for (int r = 0; r < ITH; ++r) {              // (A d) A^T
        vec m[ITW];
        wino_a<OTW>(u[r], m);
#pragma unroll
        for (int k = 0; k < ITW; ++k) *(vec*)(dM + ((long long)(r * ITW + k) * Tp + t) * C + c) = m[k];
    }
}

// dW [3][3][ci][co] (= or +=) G_h^T dU G_w, dU [ITH * ITW][ci][co]; one thread per (ci, co)
template <int OTH, int OTW>
__global__ __launch_bounds__(256) void winograd_dw_kernel(const float* __restrict__ dU, float* dW, int Cin, int Cout, int accumulate) {
    constexpr int ITH = OTH + 2, ITW = OTW + 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)Cin * Cout;
    if (i >= n) return;
    float t[3][ITW];
#pragma unroll
    for (int b = 0; b < ITW; ++b) {
        float col[ITH], o[3];
#pragma unroll
        for (int a = 0; a < ITH; ++a) col[a] = dU[(long long)(a * ITW + b) * n + i];
        wino_gt<OTH>(col, o);
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a][b] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float w[3];
        wino_gt<OTW>(t[a], w);
        float* o = dW + (long long)(a * 3) * n + i;
#pragma unroll
        for (int b = 0; b < 3; ++b) o[b * n] = accumulate ? o[b * n] + w[b] : w[b];
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

// DECONV: the product is a 2 x 2 stride-2 transposed convolution (mrcnn_deconv2x2_gemm): row m = input pixel (img, h, w), column
// n = (tap a b, channel co) -> out[img][2 h + a][2 w + b][co] = act(acc + bias[co]) -- the pixel-shuffle store of the direct kernel
struct GemmDeconvEp {
    const float* bias;
    int Cd, H, W, act;
    FastDiv d_hw, d_w;
    int nb_order;        // > 0: tiles are walked ROW-BLOCK-major (row block b = tile / nb_order outermost, the nb_order matrices
                         // xi = tile % nb_order inside): all xi of a row block finish close together (fused output transform)
};

// FUSE (round 3): the OUTPUT transform of a 4 x 4 tile group inside the GEMM launch.  Tiles are walked row-block-major
// (GemmDeconvEp::nb_order), so the 36 products of a 128-tile row block finish close together; every workgroup counts its
// finished tile into the row block's counter and the one whose count completes the block -- nobody waits for anybody --
// transforms those 128 tiles right there: A^T m A + the forward epilogue (FUSE 1) or the epilogue backward of the layer below
// (FUSE 2), the code of the stand-alone kernels (wino_output_tile / wino_output_bwd_tile).  The separate launch disappears and
// its memory pass runs beside the MFMAs of the two other workgroups of the CU, which a separate kernel cannot (the GEMM's
// workgroups hold 3 x 166 of a SIMD's 512 registers: no transform wave becomes resident beside them).  Mt still goes through
// memory (36 x 128 x 256 floats per row block do not fit a CU), freshly written and read back from L2 / Infinity Cache.
// Cross-workgroup hand-off as MI355X_MICROARCH.md prescribes: every storing wave waits vmcnt(0), workgroup barrier, ONE lane
// releases at agent scope, waits, adds to the counter (agent scope); the last arriver acquires at agent scope, waits,
// workgroup barrier, then plain loads.  The counter returns to zero by the last arriver's store.
struct WinoFuseArgs {
    int H, W, C, act, per_block;
    long long T, Tp;
    WinoGrp g;
    float* out; float* z;
    const float* bias; const float* scale; const float* shift;
    const float* below_out; const float* below_z; const float* mean; const float* rstd;
    float* dgamma; float* dbeta; float* dbias;
    int* counters;
};

template <int TN, bool DECONV, int FUSE = 0>      // TN 2: 128 x 128 tile (5 workgroups per CU); 4: 128 x 256 tile -- N = 256 whole: every V row block is fetched once
__global__ __launch_bounds__(256, FUSE ? 3 : 2) void winograd_gemm_kernel(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ Mt,
                                                               int rows, int K, int N, int total_tiles, unsigned v_records,
                                                               const GemmDeconvEp ep, const WinoFuseArgs fz) {   // FUSE: three workgroups per CU as the plain wide tile (168 VGPRs)
    constexpr int BM = 128, BN = 64 * TN, BK = 16, TM = 2;
    constexpr int AF = BM * BK, BF = BK * BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * (AF + BF) + 4];       // + one word: the fused transform's "this workgroup was last" flag
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
        int mtile = tile / ntiles;
        const int ntile = tile - mtile * ntiles;
        if (ep.nb_order > 0) {                                                 // (b, xi) -> row tile xi * (rows / BM) + b
            const int b = mtile / ep.nb_order, xi = mtile - b * ep.nb_order;
            mtile = xi * (rows / BM) + b;
        }
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
                for (int r = 0; r < 16; ++r) {
                    // FUSE: the zeroing must stay HERE -- rotated to the loop's end it would keep 128 dead-but-zero registers alive
                    // across the fused transform (seen: 131 spilled VGPRs); a volatile definition is not moved
                    if constexpr (FUSE != 0) asm volatile("v_mov_b32 %0, 0" : "=v"(acc[a][b][r]));
                    else acc[a][b][r] = 0.f;
                }
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
        const int m0_cur = m0;
        const int next = tile + (int)gridDim.x;
        const bool more = next < total_tiles;
        if (more) { k0 = 0; setup(next); stage(lds); }                         // its first stage travels under this tile's stores
        const int Mtot = (int)(v_records / (unsigned)(K * 4));
        if constexpr (!DECONV) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mw0 + a * 32 + (r & 3) + 8 * (r >> 2);
                        if (m < Mtot) Mt[(long long)m * N + nw0 + b * 32] = acc[a][b][r];
                    }
        } else {
            float bi[TN];
            int coff[TN];                                                      // (a * 2 W + b) * Cd + co of this lane's column in group b
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int n = nw0 + b * 32;
                const int tap = (n >= ep.Cd) + (n >= 2 * ep.Cd) + (n >= 3 * ep.Cd);
                const int co = n - tap * ep.Cd;
                bi[b] = ep.bias ? ep.bias[co] : 0.f;
                coff[b] = ((tap >> 1) * 2 * ep.W + (tap & 1)) * ep.Cd + co;
            }
            const int hw = ep.H * ep.W;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw0 + a * 32 + (r & 3) + 8 * (r >> 2);
                    if (m >= Mtot) continue;
                    const int img = (int)fast_div((unsigned)m, ep.d_hw), rem = m - img * hw;
                    const int h = (int)fast_div((unsigned)rem, ep.d_w), w = rem - h * ep.W;
                    const long long base = (((long long)img * 2 * ep.H + 2 * h) * 2 * ep.W + 2 * w) * ep.Cd;
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        float v = acc[a][b][r] + bi[b];
                        if (ep.act == MRCNN_ACT_RELU) v = fmaxf(v, 0.f);
                        Mt[base + coff[b]] = v;
                    }
                }
        }
        if constexpr (FUSE != 0) {
            int* flag = (int*)(lds + 2 * (AF + BF));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's stores of the tile (and the prefetched stage) are out
            __syncthreads();
            if (tid == 0) {
                if (ep.act != 77) {                                            // (77: timing experiment without the release -- results may be stale)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const int rb = (m0_cur % rows) / BM;
                const int old = __hip_atomic_fetch_add(fz.counters + rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int mine = -1;
                if (old == fz.per_block - 1) {                                 // every tile of the row block has been counted in
                    mine = rb;
                    __hip_atomic_store(fz.counters + rb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *flag = mine;
            }
            __syncthreads();
            const int rb = *flag;
            if (rb >= 0) {
                typedef typename WinoVec<2>::type vec2;
                const int c = (tid & 127) * 2;                                  // C == 256: 128 channel pairs x 2 tiles in flight
                const long long tb = (long long)rb * BM + (tid >> 7);
                if constexpr (FUSE == 1) {
                    vec2 bi = {}, sc = {1.f, 1.f}, sh = {};
                    if (fz.bias) bi = *(const vec2*)(fz.bias + c);
                    if (fz.scale) { sc = *(const vec2*)(fz.scale + c); sh = *(const vec2*)(fz.shift + c); }
                    const int itn = ep.Cd > 0 ? ep.Cd - 1 : BM / 2;             // (timing experiments: MRCNN_WINOGRAD_FUSE_ITERS caps the loop)
                    for (int it = 0; it < itn; ++it) {
                        const long long t = tb + 2 * it;
                        if (t < fz.T) wino_output_tile<4, 4, 2>(Mt, fz.out, fz.z, bi, sc, sh, fz.H, fz.W, fz.C, fz.Tp, fz.act, fz.g, t, c);
                    }
                } else {
                    vec2 sc = {1.f, 1.f}, mu = {}, rs = {}, fsh = {};
                    if (fz.scale) sc = *(const vec2*)(fz.scale + c);
                    if (fz.dgamma) { mu = *(const vec2*)(fz.mean + c); rs = *(const vec2*)(fz.rstd + c); }
                    const bool zmask = fz.shift != nullptr;
                    if (zmask) fsh = *(const vec2*)(fz.shift + c);
                    vec2 a_db = {}, a_dg = {}, a_bias = {};
                    for (int it = 0; it < BM / 2; ++it) {
                        const long long t = tb + 2 * it;
                        if (t < fz.T)
                            wino_output_bwd_tile<4, 4, 2>(Mt, fz.out, fz.below_out, fz.below_z, sc, mu, rs, fsh, zmask, fz.dgamma != nullptr, fz.H,
                                                          fz.W, fz.C, fz.Tp, fz.act, fz.g, t, c, a_db, a_dg, a_bias);
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {                               // two threads per channel, 144 row blocks: a few hundred thousand atomics per layer
                        if (fz.dbeta) atomicAdd(fz.dbeta + c + e, a_db[e]);
                        if (fz.dgamma) atomicAdd(fz.dgamma + c + e, a_dg[e]);
                        if (fz.dbias) atomicAdd(fz.dbias + c + e, a_bias[e]);
                    }
                }
            }
            __syncthreads();                                                   // the flag word is rewritten at the next tile
        }
        if (!more) break;
        tile = next;
    }
}

// Timing experiments only (results are then wrong): MRCNN_WINO_KNOCKOUT = letters of the transform kernels NOT to launch --
// i (input), o (output), b (output_bwd), y (dy), w (weights / dw).  What the step would cost if those passes were free is the
// upper bound of any fusion of them into the GEMM.
static bool wino_knocked(char which) {
    static const char* k = getenv("MRCNN_WINO_KNOCKOUT");
    return k && strchr(k, which) != nullptr;
}

// ---- host side: tile groups ------------------------------------------------------------------------------------------------------
static inline long long wino_rows(long long T) { return (T + 127) / 128 * 128; }      // rows per transform-domain matrix: whole 128-row tiles
static inline int wino_tdim(int extent, int tile) { return (extent + tile - 1) / tile; }

static int wino_grp_ok(const mrcnn_wino_group* g, int N, int H, int W, int C) {
    if (!g || (g->oth != 2 && g->oth != 4) || (g->otw != 2 && g->otw != 4)) return 0;
    if (g->th_n <= 0 || g->tw_n <= 0 || g->oh0 < 0 || g->ow0 < 0 || g->oh0 >= H || g->ow0 >= W) return 0;
    return N > 0 && H > 0 && W > 0 && C > 0 && !(C & 3);
}
static inline long long wino_grp_tiles(const mrcnn_wino_group* g, int N) { return (long long)N * g->th_n * g->tw_n; }
static inline int wino_grp_nb(const mrcnn_wino_group* g) { return (g->oth + 2) * (g->otw + 2); }
static inline int wino_grp_cv(const mrcnn_wino_group* g) { return g->oth == 2 && g->otw == 2 ? 4 : 2; }   // channels per thread

// the uniform tilings of the `tile` entries as one group: tile 2 wants even extents, tile 4 lets the last tiles hang over the edge
static int wino_uniform(int N, int H, int W, int C, int tile, mrcnn_wino_group* g) {
    if (tile != 2 && tile != 4) return 0;
    if (tile == 2 && ((H & 1) || (W & 1))) return 0;
    if (!(N > 0 && H > 0 && W > 0 && C > 0 && !(C & 3))) return 0;
    g->oth = g->otw = tile; g->th_n = wino_tdim(H, tile); g->tw_n = wino_tdim(W, tile); g->oh0 = g->ow0 = 0;
    return 1;
}

// launch the instantiation of a transform kernel for a group: <2, 2, 4> (one thread = a tile x 4 channels), the others x 2 channels
// (36 + 36 values of 4 channels would not fit the register file without spilling)
#define WINO_LAUNCH_G(kernel, g, grid4, grid2, ...)                                                                          \
    do {                                                                                                                     \
        if ((g)->oth == 2 && (g)->otw == 2) hipLaunchKernelGGL((kernel<2, 2, 4>), grid4, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else if ((g)->oth == 4 && (g)->otw == 4) hipLaunchKernelGGL((kernel<4, 4, 2>), grid2, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else if ((g)->oth == 4) hipLaunchKernelGGL((kernel<4, 2, 2>), grid2, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<2, 4, 2>), grid2, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                  \
    } while (0)
#define WINO_LAUNCH_W(kernel, oth, otw, grid, ...)                                                                           \
    do {                                                                                                                     \
        if ((oth) == 2 && (otw) == 2) hipLaunchKernelGGL((kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else if ((oth) == 4 && (otw) == 4) hipLaunchKernelGGL((kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else if ((oth) == 4) hipLaunchKernelGGL((kernel<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);     \
        else hipLaunchKernelGGL((kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                     \
    } while (0)

extern "C" size_t mrcnn_winograd_group_floats(const mrcnn_wino_group* g, int N, int C) {
    if (!g || N <= 0 || C <= 0 || (g->oth != 2 && g->oth != 4) || (g->otw != 2 && g->otw != 4) || g->th_n <= 0 || g->tw_n <= 0) return 0;
    return (size_t)wino_grp_nb(g) * (size_t)wino_rows(wino_grp_tiles(g, N)) * (size_t)C;
}

extern "C" int mrcnn_winograd_input_g(const float* x, float* V, int N, int H, int W, int C, const mrcnn_wino_group* g, void* stream) {
    if (!x || !V || !wino_grp_ok(g, N, H, W, C)) return MRCNN_ERR_ARG;
    if (wino_knocked('i')) return MRCNN_OK;
    const long long T = wino_grp_tiles(g, N);
    const WinoGrp k = {g->th_n, g->tw_n, g->oh0, g->ow0};
    const dim3 g4((unsigned)cdiv64(T * C / 4, 256)), g2((unsigned)cdiv64(T * C / 2, 256));
    WINO_LAUNCH_G(winograd_input_kernel, g, g4, g2, x, V, N, H, W, C, T, wino_rows(T), k);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_weights_g(const float* w, float* U, int Cin, int Cout, int oth, int otw, void* stream) {
    if (!w || !U || Cin <= 0 || Cout <= 0 || (oth != 2 && oth != 4) || (otw != 2 && otw != 4)) return MRCNN_ERR_ARG;
    const dim3 grid((unsigned)cdiv64((long long)Cin * Cout, 256));
    WINO_LAUNCH_W(winograd_weight_kernel, oth, otw, grid, w, U, Cin, Cout);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_output_g(const float* Mt, float* out, float* z, const float* bias, const float* scale, const float* shift,
                                       int N, int H, int W, int C, int act, const mrcnn_wino_group* g, void* stream) {
    if (!Mt || !out || !wino_grp_ok(g, N, H, W, C) || (scale && !shift)) return MRCNN_ERR_ARG;
    if (act != MRCNN_ACT_NONE && act != MRCNN_ACT_RELU) return MRCNN_ERR_UNSUPPORTED;
    if (wino_knocked('o')) return MRCNN_OK;
    const long long T = wino_grp_tiles(g, N);
    const WinoGrp k = {g->th_n, g->tw_n, g->oh0, g->ow0};
    const dim3 g4((unsigned)cdiv64(T * C / 4, 256)), g2((unsigned)cdiv64(T * C / 2, 256));
    WINO_LAUNCH_G(winograd_output_kernel, g, g4, g2, Mt, out, z, bias, scale, shift, N, H, W, C, T, wino_rows(T), act, k);
    return mrcnn_launch_status();
}

static int wino_output_bwd_g(const float* Mt, float* dz_below, const float* below_out, const float* below_z, const float* scale,
                             const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H,
                             int W, int C, int act, const mrcnn_wino_group* g, void* stream, const float* fwd_shift);

extern "C" int mrcnn_winograd_output_bwd_g(const float* Mt, float* dz_below, const float* below_out, const float* below_z, const float* scale,
                                           const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H,
                                           int W, int C, int act, const mrcnn_wino_group* g, void* stream) {
    return wino_output_bwd_g(Mt, dz_below, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, N, H, W, C, act, g, stream, nullptr);
}

/* the same for a ReLU layer below whose forward epilogue was out = max(scale * z + shift, 0): the mask comes from z (bit for bit
 * the forward's decision), `out` is never read */
extern "C" int mrcnn_winograd_output_bwd_zmask_g(const float* Mt, float* dz_below, const float* below_z, const float* scale,
                                                 const float* shift, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                                 float* dbias, int N, int H, int W, int C, const mrcnn_wino_group* g, void* stream) {
    if (!below_z || !scale || !shift) return MRCNN_ERR_ARG;
    return wino_output_bwd_g(Mt, dz_below, nullptr, below_z, scale, mean, rstd, dgamma, dbeta, dbias, N, H, W, C, MRCNN_ACT_RELU, g, stream, shift);
}

static int wino_output_bwd_g(const float* Mt, float* dz_below, const float* below_out, const float* below_z, const float* scale,
                             const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H,
                             int W, int C, int act, const mrcnn_wino_group* g, void* stream, const float* fwd_shift) {
    if (!Mt || !dz_below || !wino_grp_ok(g, N, H, W, C)) return MRCNN_ERR_ARG;
    if ((act != MRCNN_ACT_NONE && act != MRCNN_ACT_RELU) || (act == MRCNN_ACT_RELU && !below_out && !fwd_shift)) return MRCNN_ERR_ARG;
    if (dgamma && (!below_z || !mean || !rstd)) return MRCNN_ERR_ARG;
    if (wino_knocked('b')) return MRCNN_OK;
    const int cv = wino_grp_cv(g);
    const int cn = C / cv;
    if (cn > 256 || (cn & (cn - 1))) return MRCNN_ERR_UNSUPPORTED;
    int lg = 0;
    while ((1 << lg) < cn) ++lg;
    const long long T = wino_grp_tiles(g, N);
    const long long R = 256 >> lg;
    long long per = (T + 2047) / 2048;                          // ~2048 workgroups; every one a whole number of passes
    per = (per + R - 1) / R * R;
    const dim3 grid((unsigned)cdiv64(T, per));
    const WinoGrp k = {g->th_n, g->tw_n, g->oh0, g->ow0};
    WINO_LAUNCH_G(winograd_output_bwd_kernel, g, grid, grid, Mt, dz_below, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, N, H, W,
                  C, T, wino_rows(T), act, per, lg, k, fwd_shift);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_dy_g(const float* dy, float* dM, int N, int H, int W, int C, const mrcnn_wino_group* g, void* stream) {
    if (!dy || !dM || !wino_grp_ok(g, N, H, W, C)) return MRCNN_ERR_ARG;
    if (wino_knocked('y')) return MRCNN_OK;
    const long long T = wino_grp_tiles(g, N);
    const WinoGrp k = {g->th_n, g->tw_n, g->oh0, g->ow0};
    const dim3 g4((unsigned)cdiv64(T * C / 4, 256)), g2((unsigned)cdiv64(T * C / 2, 256));
    WINO_LAUNCH_G(winograd_dy_kernel, g, g4, g2, dy, dM, N, H, W, C, T, wino_rows(T), k);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_winograd_dw_g(const float* dU, float* dw_hwio, int Cin, int Cout, int accumulate, int oth, int otw, void* stream) {
    if (!dU || !dw_hwio || Cin <= 0 || Cout <= 0 || (oth != 2 && oth != 4) || (otw != 2 && otw != 4)) return MRCNN_ERR_ARG;
    const dim3 grid((unsigned)cdiv64((long long)Cin * Cout, 256));
    WINO_LAUNCH_W(winograd_dw_kernel, oth, otw, grid, dU, dw_hwio, Cin, Cout, accumulate);
    return mrcnn_launch_status();
}

// ---- the uniform tilings (tile = 2 or 4 outputs per tile and dimension) as one group ----------------------------------------------
/* floats of V (input transform) or Mt (transform-domain product) for a layer: (tile + 2)^2 x rows x C */
extern "C" size_t mrcnn_winograd_buffer_floats(int N, int H, int W, int C, int tile) {
    mrcnn_wino_group g;
    if (!wino_uniform(N, H, W, C, tile, &g)) return 0;
    return mrcnn_winograd_group_floats(&g, N, C);
}

extern "C" int mrcnn_winograd_input(const float* x, float* V, int N, int H, int W, int C, int tile, void* stream) {
    mrcnn_wino_group g;
    if (!wino_uniform(N, H, W, C, tile, &g)) return MRCNN_ERR_ARG;
    return mrcnn_winograd_input_g(x, V, N, H, W, C, &g, stream);
}

extern "C" int mrcnn_winograd_weights(const float* w, float* U, int Cin, int Cout, int tile, void* stream) {
    return mrcnn_winograd_weights_g(w, U, Cin, Cout, tile, tile, stream);
}

extern "C" int mrcnn_winograd_output(const float* Mt, float* out, float* z, const float* bias, const float* scale, const float* shift,
                                     int N, int H, int W, int C, int act, int tile, void* stream) {
    mrcnn_wino_group g;
    if (!wino_uniform(N, H, W, C, tile, &g)) return MRCNN_ERR_ARG;
    return mrcnn_winograd_output_g(Mt, out, z, bias, scale, shift, N, H, W, C, act, &g, stream);
}

extern "C" int mrcnn_winograd_output_bwd(const float* Mt, float* dz_below, const float* below_out, const float* below_z, const float* scale,
                                         const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias, int N, int H, int W,
                                         int C, int act, int tile, void* stream) {
    mrcnn_wino_group g;
    if (!wino_uniform(N, H, W, C, tile, &g)) return MRCNN_ERR_ARG;
    return mrcnn_winograd_output_bwd_g(Mt, dz_below, below_out, below_z, scale, mean, rstd, dgamma, dbeta, dbias, N, H, W, C, act, &g, stream);
}

extern "C" int mrcnn_winograd_dy(const float* dy, float* dM, int N, int H, int W, int C, int tile, void* stream) {
    mrcnn_wino_group g;
    if (!wino_uniform(N, H, W, C, tile, &g)) return MRCNN_ERR_ARG;
    return mrcnn_winograd_dy_g(dy, dM, N, H, W, C, &g, stream);
}

extern "C" int mrcnn_winograd_dw(const float* dU, float* dw_hwio, int Cin, int Cout, int accumulate, int tile, void* stream) {
    return mrcnn_winograd_dw_g(dU, dw_hwio, Cin, Cout, accumulate, tile, tile, stream);
}

/* the persistent form of mrcnn_gemm_batched_f32 (same arguments and result; K % 16 == 0, N % 128 == 0, rows % 128 == 0) */
extern "C" int mrcnn_winograd_gemm(const float* V, const float* U, float* Mt, int nb, int rows, int K, int N, void* stream) {
    if (!V || !U || !Mt || nb <= 0 || rows <= 0 || rows % 128 || K <= 0 || K % 16 || N <= 0 || N % 128) return MRCNN_ERR_ARG;
    const long long M = (long long)nb * rows;
    if (M * K * 4 >= 0x7FFFFFF0LL || (long long)nb * K * N * 4 >= 0x7FFFFFF0LL || M * N >= (1LL << 40)) return MRCNN_ERR_UNSUPPORTED;
    // The wide tile (N = 256 whole: every V row block is fetched once; 166 VGPRs, 48 KiB LDS, 3 workgroups per CU) is 4-5 % faster
    // alone (F(4x4) layer: 1.25 -> 1.19 ms, 0.79 -> 0.83 of the matrix peak).  Inside the step it used to lose (F(2x2): 44.0 ->
    // 44.9 ms, less room for the other stream's kernels on a CU); with the F(4x4) layers it is level or slightly ahead on the
    // dense step (40.85 / 40.60 -> 40.61 / 40.51 ms) and level on the positive-quota step, so it is taken for products of at least
    // MRCNN_WINOGRAD_GEMM_WIDE_MIN (4096) wide tiles -- the dense layers and their halves; 0 = always, -1 = never.
    const char* wenv = getenv("MRCNN_WINOGRAD_GEMM_WIDE_MIN");                  // read per call (tests compare the two tiles)
    const long long wide_min = wenv ? atoll(wenv) : 4096;
    const bool wide = wide_min >= 0 && N % 256 == 0 && (M / 128) * (N / 256) >= wide_min;
    GemmDeconvEp none = {};
    { static const char* oe = getenv("MRCNN_WINOGRAD_GEMM_ORDER"); if (oe && oe[0] == 'b') none.nb_order = nb; }   // A/B (once per process): row-block-major tile order
    if (wide) {                                                 // 128 x 256 tiles: 48 KiB LDS, 3 workgroups per CU
        const long long tiles = (M / 128) * (N / 256);
        const long long slots = 3LL * mrcnn_num_cus();
        const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
        hipLaunchKernelGGL((winograd_gemm_kernel<4, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, V, U, Mt, rows, K, N, (int)tiles,
                           (unsigned)(M * K * 4), none, WinoFuseArgs{});
        return mrcnn_launch_status();
    }
    const long long tiles = (M / 128) * (N / 128);
    const long long slots = 5LL * mrcnn_num_cus();
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL((winograd_gemm_kernel<2, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, V, U, Mt, rows, K, N, (int)tiles,
                       (unsigned)(M * K * 4), none, WinoFuseArgs{});
    return mrcnn_launch_status();
}

/* The 36 transform-domain GEMMs of a 4 x 4 tile group with the group's OUTPUT transform fused into the launch (see
 * winograd_gemm_kernel, FUSE).  mode 1: forward (A^T m A + bias, frozen-BN affine, activation, optional z); mode 2: data gradient
 * (epilogue backward of the layer below: ReLU mask from below_out -- or, shift given, from below_z --, BN scale, channel sums).
 * Needs the wide tile: Cout == 256 (one column tile: a row block is complete after nb tiles), K % 16 == 0, a 4 x 4 group;
 * MRCNN_ERR_UNSUPPORTED otherwise (the caller runs mrcnn_winograd_gemm + mrcnn_winograd_output*_g).  counters: one int per
 * 128-row block of a matrix (rows / 128), zero on entry, zero again on exit. */
extern "C" int mrcnn_winograd_gemm_fused(const float* V, const float* U, float* Mt, int nb, int rows, int K, int N, const mrcnn_wino_fuse* f,
                                         void* stream) {
    if (!V || !U || !Mt || !f || nb <= 0 || rows <= 0 || rows % 128 || K <= 0 || K % 16 || N <= 0) return MRCNN_ERR_ARG;
    if ((f->mode != 1 && f->mode != 2) || !f->out || !f->counters || !wino_grp_ok(&f->g, f->N, f->H, f->W, N)) return MRCNN_ERR_ARG;
    if (N != 256 || f->g.oth != 4 || f->g.otw != 4 || nb != 36) return MRCNN_ERR_UNSUPPORTED;
    if (f->act != MRCNN_ACT_NONE && f->act != MRCNN_ACT_RELU) return MRCNN_ERR_UNSUPPORTED;
    const long long T = wino_grp_tiles(&f->g, f->N);
    if (wino_rows(T) != rows) return MRCNN_ERR_ARG;
    if (f->mode == 1 && f->scale && !f->shift) return MRCNN_ERR_ARG;
    if (f->mode == 2 && ((f->act == MRCNN_ACT_RELU && !f->below_out && !(f->shift && f->below_z && f->scale)) ||
                         (f->dgamma && (!f->below_z || !f->mean || !f->rstd))))
        return MRCNN_ERR_ARG;
    const long long M = (long long)nb * rows;
    if (M * K * 4 >= 0x7FFFFFF0LL || (long long)nb * K * N * 4 >= 0x7FFFFFF0LL || M * N >= (1LL << 40)) return MRCNN_ERR_UNSUPPORTED;
    WinoFuseArgs z = {};
    z.H = f->H; z.W = f->W; z.C = N; z.act = f->act; z.per_block = nb; z.T = T; z.Tp = rows;
    z.g = {f->g.th_n, f->g.tw_n, f->g.oh0, f->g.ow0};
    z.out = f->out; z.z = f->z; z.bias = f->bias; z.scale = f->scale; z.shift = f->shift;
    z.below_out = f->below_out; z.below_z = f->below_z; z.mean = f->mean; z.rstd = f->rstd;
    z.dgamma = f->dgamma; z.dbeta = f->dbeta; z.dbias = f->dbias; z.counters = f->counters;
    GemmDeconvEp ep = {};
    ep.nb_order = nb;
    // timing experiments only (read once per process; the fused path itself is opt-in): no release fence / a capped transform loop
    static const bool no_release = getenv("MRCNN_WINOGRAD_FUSE_NORELEASE") != nullptr;
    static const int iters = getenv("MRCNN_WINOGRAD_FUSE_ITERS") ? atoi(getenv("MRCNN_WINOGRAD_FUSE_ITERS")) + 1 : 0;
    if (no_release) ep.act = 77;
    if (iters > 0) ep.Cd = iters;
    const long long tiles = M / 128;
    const long long slots = 3LL * mrcnn_num_cus();
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    if (f->mode == 1)
        hipLaunchKernelGGL((winograd_gemm_kernel<4, false, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, V, U, Mt, rows, K, N, (int)tiles,
                           (unsigned)(M * K * 4), ep, z);
    else
        hipLaunchKernelGGL((winograd_gemm_kernel<4, false, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, V, U, Mt, rows, K, N, (int)tiles,
                           (unsigned)(M * K * 4), ep, z);
    return mrcnn_launch_status();
}

/* Conv2DTranspose(2 x 2, stride 2) as ONE product [N H W x Cin] . [Cin x 4 Cd] on the persistent GEMM above with the pixel-shuffle
 * store in its epilogue (bias, ReLU): K is one filter tap deep (256 in the mask head), the case the persistent form exists for.
 * w_gemm [Cin][(a, b, co)]; out [N][2 H][2 W][Cd].  Cin % 16 == 0, Cd % 32 == 0, 4 Cd % 128 == 0. */
extern "C" int mrcnn_deconv2x2_gemm(const float* x, const float* w_gemm, const float* bias, float* out, int N, int H, int W, int Cin, int Cd,
                                    int act, void* stream) {
    if (!x || !w_gemm || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cd <= 0) return MRCNN_ERR_ARG;
    if (act != MRCNN_ACT_NONE && act != MRCNN_ACT_RELU) return MRCNN_ERR_UNSUPPORTED;
    const long long M = (long long)N * H * W;
    const int Nn = 4 * Cd;
    if (Cin % 16 || Cd % 32 || Nn % 128 || M * Cin * 4 >= 0x7FFFFFF0LL || (long long)Cin * Nn * 4 >= 0x7FFFFFF0LL || M >= (1LL << 31) - 128)
        return MRCNN_ERR_UNSUPPORTED;
    GemmDeconvEp ep = {};
    ep.bias = bias; ep.Cd = Cd; ep.H = H; ep.W = W; ep.act = act;
    ep.d_hw = make_fastdiv((unsigned)(H * W)); ep.d_w = make_fastdiv((unsigned)W);
    const long long tiles = ((M + 127) / 128) * (Nn / 128);
    const long long slots = 5LL * mrcnn_num_cus();
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    const int rows = (int)((M + 127) / 128 * 128);              // one matrix: every tile uses weight matrix 0
    hipLaunchKernelGGL((winograd_gemm_kernel<2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w_gemm, out, rows, Cin, Nn, (int)tiles,
                       (unsigned)(M * Cin * 4), ep, WinoFuseArgs{});
    return mrcnn_launch_status();
}
