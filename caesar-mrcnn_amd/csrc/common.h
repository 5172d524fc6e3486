// Shared device/host helpers for the gfx950 kernels of the Mask R-CNN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include "mrcnn_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// Process-wide tuning values (mrcnn_tuning_set): host-side, read at launch time.
//   wgrad_lds_pad: extra dynamic LDS bytes requested by the large LDS-DMA weight-gradient kernel.  Its 32 KiB of static LDS
//   let five workgroups fill a CU's 160 KiB; padded to 40 KiB only four fit, which leaves one workgroup slot per CU to
//   the small latency-bound kernels of another stream (the engine sets it around the weight gradients it runs beside the
//   backbone's backward pass, see engine.py "deferred mask-head weight gradients").
extern int g_mrcnn_wgrad_lds_pad;
//   h16_phase: 1 (default) lets the 16-bit forward pick the persistent 256 x 256 kernel by itself; 0 keeps it off.  One
//   workgroup of it fills a CU's LDS, so beside another stream's big kernels (the mask head's weight gradients run next to
//   its data gradients) the statically assigned tiles start late on the CUs the other kernel held.
extern int g_mrcnn_h16_phase;
extern int g_mrcnn_h16_slab;
extern int g_mrcnn_sk16;
//   proposal_skip_zero: tests only (fault injection): 1 suppresses the reset of the multi-workgroup top-k's counters.
extern int g_mrcnn_proposal_skip_zero;

static inline int mrcnn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MRCNN_OK : MRCNN_ERR_LAUNCH;
}

// Compute units of the current device (persistent kernels launch one workgroup per CU); queried once per device.
static inline int mrcnn_num_cus() {
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Workgroups are dealt to the 8 XCDs round robin (workgroup b runs on XCD b % 8), and every XCD has its own L2.  This maps
// the hardware id to a LOGICAL id such that consecutive logical ids share an XCD: XCD x owns the logical range
// [x * per + min(x, rem), ...) with per = blocks / 8, rem = blocks % 8.  A bijection on [0, blocks).  Kernels whose
// neighbouring workgroups read the same operands (the pixel splits of a weight gradient: all (tap, channel tile) workgroups
// of one split) index their work with it, so that the operands are fetched into ONE L2 instead of eight.
__device__ __forceinline__ unsigned mrcnn_xcd_contiguous(unsigned b, unsigned blocks) {
    const unsigned per = blocks >> 3, rem = blocks & 7u, x = b & 7u, slot = b >> 3;
    return x * per + (x < rem ? x : rem) + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// TF non_max_suppression overlap test [3P]: canonical corners, 0 for empty boxes, IoU > thr.
__device__ __forceinline__ bool iou_gt(const float* a, const float* bx, float thr) {
    const float ymin_i = fminf(a[0], a[2]), xmin_i = fminf(a[1], a[3]);
    const float ymax_i = fmaxf(a[0], a[2]), xmax_i = fmaxf(a[1], a[3]);
    const float ymin_j = fminf(bx[0], bx[2]), xmin_j = fminf(bx[1], bx[3]);
    const float ymax_j = fmaxf(bx[0], bx[2]), xmax_j = fmaxf(bx[1], bx[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return false;
    const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
    const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
    const float iou = inter / (area_i + area_j - inter);
    return iou > thr;
}


// 64-bit value of lane `src` for a wave-uniform src: two v_readlane_b32 (a few cycles, result in SGPRs) instead of
// two LDS-crossbar shuffles
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int src) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
    unsigned lo = (unsigned)__shfl((int)(unsigned)(v & 0xFFFFFFFFull), src, 64);
    unsigned hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}


// apply_box_deltas_graph + clip_boxes_graph (mrcnn/model.py:287-326), float32 op by op.
__device__ __forceinline__ void decode_clip_box(const float* box, float d0, float d1, float d2, float d3,
                                                float wy1, float wx1, float wy2, float wx2, float* o) {
    float height = box[2] - box[0], width = box[3] - box[1];
    float cy = box[0] + 0.5f * height, cx = box[1] + 0.5f * width;
    cy += d0 * height;
    cx += d1 * width;
    height *= expf(d2);
    width *= expf(d3);
    float y1 = cy - 0.5f * height, x1 = cx - 0.5f * width;
    float y2 = y1 + height, x2 = x1 + width;
    o[0] = fmaxf(fminf(y1, wy2), wy1);
    o[1] = fmaxf(fminf(x1, wx2), wx1);
    o[2] = fmaxf(fminf(y2, wy2), wy1);
    o[3] = fmaxf(fminf(x2, wx2), wx1);
}

// Exact unsigned division by a runtime-constant divisor (n < 2^31): q = (umulhi(n, mul) + n) >> shift.
struct FastDiv {
    unsigned mul, shift, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.shift = l;
    f.d = d;
    return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.shift; }

// Per-output-pixel addressing table of the weight-gradient kernels: byte offset of input pixel
// (n, oh*stride - pad_t, ow*stride - pad_l) from the (shifted) tensor base and the validity mask of the
// KH*KW filter taps (<= 64).  Rows past M carry an out-of-range offset and an empty mask.  Built once per call,
// it takes the two divisions per pixel out of the kernels' loops.
#define PIXEL_TABLE_OOB 0x80000000u
struct PixelEntry { unsigned off, mask_lo, mask_hi, pad; };

static __global__ void pixel_table_kernel(PixelEntry* table, int N, int H, int W, int Cin, int KH, int KW, int stride, int pad_t,
                                          int pad_l, int OH, int OW, int M, int rows, unsigned x_shift, int esize) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= rows) return;
    PixelEntry e;
    e.off = PIXEL_TABLE_OOB; e.mask_lo = 0u; e.mask_hi = 0u; e.pad = 0u;
    if (m < M) {
        const int ohw = OH * OW;
        const int n = m / ohw, rem = m - n * ohw;
        const int oh = rem / OW, ow = rem - oh * OW;
        const int ih0 = oh * stride - pad_t, iw0 = ow * stride - pad_l;
        e.off = (unsigned)((((long long)n * H + ih0) * W + iw0) * Cin * esize + x_shift);
        unsigned long long mk = 0ull;
        for (int t = 0; t < KH * KW; ++t) {
            const int th = t / KW, tw = t - th * KW;
            if ((unsigned)(ih0 + th) < (unsigned)H && (unsigned)(iw0 + tw) < (unsigned)W) mk |= 1ull << t;
        }
        e.mask_lo = (unsigned)mk; e.mask_hi = (unsigned)(mk >> 32);
    }
    table[m] = e;
}

// MRCNN_CONV_FLAT_GLDS=1 (read once): keep the flat-addressed LDS-DMA kernels, which are otherwise only used for
// tensors too large for a 32-bit buffer descriptor -- lets the tests and A/B timings reach them on small shapes.
// Sum of `ks` split slabs at element offset e, in slab order (bitwise the same as the plain loop), with the loads
// issued eight at a time: written as `for k: s += slab[k]` the compiler waits for every load before the next
// (one memory round trip per slab -- measured 6-10 us per reduction launch on the latency-bound backbone layers).
template <typename V>
__device__ __forceinline__ V mrcnn_slab_sum(V acc, const float* __restrict__ slab, long long stride, long long e, int ks) {
    for (int k0 = 0; k0 < ks; k0 += 8) {               // uniform trip count
        V v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = k0 + j;
            if (k > ks - 1) k = ks - 1;                // clamped: a valid (cached) address, value not used
            v[j] = *(const V*)(slab + (long long)k * stride + e);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < ks) acc += v[j];
    }
    return acc;
}

// Launch shape of the vector backward-epilogue kernels (C = 4 * 2^k): workgroups of 2^lg float4 channel groups x
// rows_per_block rows.  MRCNN_EPI_MIN_ELEMS / MRCNN_EPI_LANES override the tile (read once; A/B timings).
struct EpiGrid { unsigned row_blocks, chan_blocks; long long rows_per_block; int lg; };
static inline EpiGrid mrcnn_epilogue_grid(long long M, int C) {
    static const long long min_elems = getenv("MRCNN_EPI_MIN_ELEMS") ? atoll(getenv("MRCNN_EPI_MIN_ELEMS")) : 4096;
    static const int max_lanes = getenv("MRCNN_EPI_LANES") ? atoi(getenv("MRCNN_EPI_LANES")) : 16;
    EpiGrid g;
    const int c4n = C >> 2;
    int lanes = c4n < 256 ? c4n : 256;
    if (lanes > max_lanes && max_lanes >= 4) lanes = max_lanes;
    g.lg = 0;
    while ((2 << g.lg) <= lanes) ++g.lg;                   // floor(log2(lanes)); lanes is a power of two here
    const long long R = 256 >> g.lg;
    long long rows = (M + 2047) / 2048;
    const long long min_rows = (min_elems + (4ll << g.lg) - 1) / (4ll << g.lg);
    if (rows < min_rows) rows = min_rows;
    if (rows < R) rows = R;
    g.rows_per_block = rows;
    g.row_blocks = (unsigned)((M + rows - 1) / rows);
    g.chan_blocks = (unsigned)(c4n >> g.lg);
    return g;
}

static inline bool mrcnn_force_flat_glds() {
    static const bool v = getenv("MRCNN_CONV_FLAT_GLDS") != nullptr;
    return v;
}
