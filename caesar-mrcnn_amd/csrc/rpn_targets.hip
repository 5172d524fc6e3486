// RPN training targets on the device: the per-image target builder of the CPU input pipeline
// (mrcnn/model.py:1536-1644, build_rpn_targets) as three small kernels.  Integer / float64 work, HBM- and
// latency-bound (A x G IoUs in double, A = 16 368 anchors at 256^2, G <= MAX_GT_INSTANCES):
//
//   rpn_colmax_kernel   per anchor: IoU with every real GT box (float64, same operation order as
//                       utils.compute_overlaps), column maxima via 64-bit atomicMax on the bit pattern
//   rpn_match_kernel    per anchor: row max / first argmax, crowd test, "every GT keeps its best anchors"
//                       (bitwise equality with the column maximum), >= 0.7 -> +1, < 0.3 -> -1
//   rpn_sample_kernel   one workgroup per image: cap positives at n_train/2 and negatives at
//                       n_train - positives by dropping the candidates with the SMALLEST injected random
//                       keys (radix select; ties -> lower anchor index first), then the box deltas of the
//                       positives in ascending anchor order, divided by RPN_BBOX_STD_DEV
//
// The reference draws the dropped subset with np.random.choice; the keys replace that draw exactly as
// rand_keys replace tf.random_shuffle in the detection-target kernel (same distribution, reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mrcnn_hip.h"

#define RT_THREADS 1024

struct RpnTArgs {
    const double* anchors;      // [A,4] pixels (y1,x1,y2,x2), float64 like utils.generate_pyramid_anchors
    const int* gt_cls;          // [B,G]  >0 real, <0 crowd, 0 padding
    const int* gt_boxes;        // [B,G,4] int32 pixels
    const float* keys;          // [B,A] uniform [0,1)
    int* match;                 // [B,A]
    float* bbox;                // [B,n_train,4]
    unsigned long long* colmax; // [B,G] workspace (bit patterns of non-negative doubles)
    int* arg;                   // [B,A] workspace: GT row of the first row-maximum
    int B, A, G, n_train;
    double sd[4];
};

__device__ __forceinline__ double rt_iou(const double a0, const double a1, const double a2, const double a3,
                                         const double aarea, const int* __restrict__ g, const double garea) {
    const double y1 = fmax((double)g[0], a0), y2 = fmin((double)g[2], a2);
    const double x1 = fmax((double)g[1], a1), x2 = fmin((double)g[3], a3);
    const double inter = fmax(x2 - x1, 0.0) * fmax(y2 - y1, 0.0);
    return inter / ((garea + aarea) - inter);
}

// GT boxes of one image staged in LDS: box ints + int32 area (numpy computes it in int32) + class
#define RT_MAXG 512
struct GtTile { int box[RT_MAXG][4]; int area[RT_MAXG]; int cls[RT_MAXG]; };

__device__ __forceinline__ void load_gt(GtTile& t, const RpnTArgs& p, int b) {
    for (int g = threadIdx.x; g < p.G; g += blockDim.x) {
        const int* src = p.gt_boxes + ((size_t)b * p.G + g) * 4;
        int y1 = src[0], x1 = src[1], y2 = src[2], x2 = src[3];
        t.box[g][0] = y1; t.box[g][1] = x1; t.box[g][2] = y2; t.box[g][3] = x2;
        t.area[g] = (y2 - y1) * (x2 - x1);
        t.cls[g] = p.gt_cls[(size_t)b * p.G + g];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) rpn_colmax_kernel(RpnTArgs p) {
    __shared__ GtTile t;
    const int b = blockIdx.y;
    load_gt(t, p, b);
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.A) return;
    const double a0 = p.anchors[a * 4 + 0], a1 = p.anchors[a * 4 + 1], a2 = p.anchors[a * 4 + 2], a3 = p.anchors[a * 4 + 3];
    const double aarea = (a2 - a0) * (a3 - a1);
    for (int g = 0; g < p.G; ++g) {
        if (t.cls[g] <= 0) continue;
        const double v = rt_iou(a0, a1, a2, a3, aarea, t.box[g], (double)t.area[g]);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        unsigned long long* dst = p.colmax + (size_t)b * p.G + g;
        if (bits > *(volatile unsigned long long*)dst) atomicMax(dst, bits);
    }
}

__global__ void __launch_bounds__(256) rpn_match_kernel(RpnTArgs p) {
    __shared__ GtTile t;
    __shared__ unsigned long long cm[RT_MAXG];
    const int b = blockIdx.y;
    for (int g = threadIdx.x; g < p.G; g += blockDim.x) cm[g] = p.colmax[(size_t)b * p.G + g];
    load_gt(t, p, b);
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.A) return;
    const double a0 = p.anchors[a * 4 + 0], a1 = p.anchors[a * 4 + 1], a2 = p.anchors[a * 4 + 2], a3 = p.anchors[a * 4 + 3];
    const double aarea = (a2 - a0) * (a3 - a1);
    double best = 0.0, crowd = 0.0;
    int arg = -1;
    bool has_crowd = false, gt_best = false;
    for (int g = 0; g < p.G; ++g) {
        const int c = t.cls[g];
        if (c == 0) continue;
        const double v = rt_iou(a0, a1, a2, a3, aarea, t.box[g], (double)t.area[g]);
        if (c < 0) { has_crowd = true; crowd = fmax(crowd, v); continue; }
        if (arg < 0 || v > best) { best = v; arg = g; }            // np.argmax: first maximum
        gt_best |= ((unsigned long long)__double_as_longlong(v) == cm[g]);
    }
    const bool no_crowd = !has_crowd || crowd < 0.001;
    int m = (best < 0.3 && no_crowd) ? -1 : 0;
    if (gt_best) m = 1;
    if (best >= 0.7) m = 1;
    p.match[(size_t)b * p.A + a] = m;
    p.arg[(size_t)b * p.A + a] = arg;
}

// ---- workgroup helpers -------------------------------------------------------------------------------
__device__ __forceinline__ int block_sum(int v, int* s_red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < RT_THREADS / 64; ++w) tot += s_red[w];
    return tot;
}

// exclusive prefix of v over the workgroup in thread order
__device__ __forceinline__ int block_exscan(int v, int* s_red) {
    int inc = v;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
    __syncthreads();
    if (lane == 63) s_red[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_red[w];
    return base + inc - v;
}

// Set match to 0 for the `drop` members of {a : match[a] == cls} with the smallest (key, a).
__device__ void drop_smallest(int* __restrict__ match, const float* __restrict__ keys, int A, int cls, int drop,
                              int* s_hist, int* s_red, int* s_bcast) {
    if (drop <= 0) return;
    const int chunk = (A + RT_THREADS - 1) / RT_THREADS;
    const int lo = threadIdx.x * chunk, hi = min(A, lo + chunk);
    unsigned prefix = 0, pmask = 0;
    int remaining = drop;                       // members with key < threshold are all dropped; `remaining` ties left
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < 256; i += RT_THREADS) s_hist[i] = 0;
        __syncthreads();
        for (int a = lo; a < hi; ++a) {
            if (match[a] != cls) continue;
            const unsigned k = __float_as_uint(keys[a]);
            if ((k & pmask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int cum = 0, d = 0;
            for (; d < 256; ++d) { if (cum + s_hist[d] >= remaining) break; cum += s_hist[d]; }
            s_bcast[0] = d; s_bcast[1] = remaining - cum;
        }
        __syncthreads();
        prefix |= ((unsigned)s_bcast[0]) << shift;
        pmask |= 255u << shift;
        remaining = s_bcast[1];
        __syncthreads();
    }
    // keys < prefix dropped; among keys == prefix the first `remaining` in anchor order
    int ties = 0;
    for (int a = lo; a < hi; ++a)
        if (match[a] == cls && __float_as_uint(keys[a]) == prefix) ++ties;
    int rank = block_exscan(ties, s_red);
    for (int a = lo; a < hi; ++a) {
        if (match[a] != cls) continue;
        const unsigned k = __float_as_uint(keys[a]);
        if (k < prefix) match[a] = 0;
        else if (k == prefix) { if (rank < remaining) match[a] = 0; ++rank; }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(RT_THREADS) rpn_sample_kernel(RpnTArgs p) {
    __shared__ int s_hist[256];
    __shared__ int s_red[RT_THREADS / 64];
    __shared__ int s_bcast[2];
    const int b = blockIdx.x;
    int* match = p.match + (size_t)b * p.A;
    const float* keys = p.keys + (size_t)b * p.A;
    const int* arg = p.arg + (size_t)b * p.A;
    float* bbox = p.bbox + (size_t)b * p.n_train * 4;
    const int chunk = (p.A + RT_THREADS - 1) / RT_THREADS;
    const int lo = threadIdx.x * chunk, hi = min(p.A, lo + chunk);

    int np_ = 0, nn = 0;
    for (int a = lo; a < hi; ++a) { np_ += match[a] == 1; nn += match[a] == -1; }
    int P = block_sum(np_, s_red);
    const int N = block_sum(nn, s_red);
    const int half = p.n_train / 2;
    if (P > half) { drop_smallest(match, keys, p.A, 1, P - half, s_hist, s_red, s_bcast); P = half; }
    const int want_neg = p.n_train - P;
    if (N > want_neg) drop_smallest(match, keys, p.A, -1, N - want_neg, s_hist, s_red, s_bcast);

    // box deltas of the positives, ascending anchor index (model.py:1616-1642)
    int mine = 0;
    for (int a = lo; a < hi; ++a) mine += match[a] == 1;
    int row = block_exscan(mine, s_red);
    for (int a = lo; a < hi; ++a) {
        if (match[a] != 1) continue;
        if (row < p.n_train) {
            const int* g = p.gt_boxes + ((size_t)b * p.G + arg[a]) * 4;
            const double gh = (double)(g[2] - g[0]), gw = (double)(g[3] - g[1]);
            const double gcy = (double)g[0] + 0.5 * gh, gcx = (double)g[1] + 0.5 * gw;
            const double a0 = p.anchors[a * 4 + 0], a1 = p.anchors[a * 4 + 1], a2 = p.anchors[a * 4 + 2], a3 = p.anchors[a * 4 + 3];
            const double ah = a2 - a0, aw = a3 - a1;
            const double acy = a0 + 0.5 * ah, acx = a1 + 0.5 * aw;
            bbox[row * 4 + 0] = (float)(((gcy - acy) / ah) / p.sd[0]);
            bbox[row * 4 + 1] = (float)(((gcx - acx) / aw) / p.sd[1]);
            bbox[row * 4 + 2] = (float)(log(gh / ah) / p.sd[2]);
            bbox[row * 4 + 3] = (float)(log(gw / aw) / p.sd[3]);
        }
        ++row;
    }
    // rows past the positives stay zero
    const int total = block_sum(mine, s_red);
    for (int i = min(total, p.n_train) * 4 + threadIdx.x; i < p.n_train * 4; i += RT_THREADS) bbox[i] = 0.f;
}

extern "C" size_t mrcnn_rpn_targets_workspace(const mrcnn_rpntarget_desc* d) {
    if (!d) return 0;
    return (size_t)d->B * d->G * sizeof(unsigned long long) + (size_t)d->B * d->A * sizeof(int);
}

extern "C" int mrcnn_rpn_targets(const mrcnn_rpntarget_desc* d, const double* anchors, const int32_t* gt_class_ids,
                                 const int32_t* gt_boxes, const float* rand_keys, int32_t* rpn_match, float* rpn_bbox,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !anchors || !gt_class_ids || !gt_boxes || !rand_keys || !rpn_match || !rpn_bbox || !workspace) return -1;
    if (d->B <= 0 || d->A <= 0 || d->G <= 0 || d->G > RT_MAXG || d->n_train <= 0) return -2;
    if (workspace_bytes < mrcnn_rpn_targets_workspace(d)) return -3;
    hipStream_t s = (hipStream_t)stream;
    RpnTArgs p;
    p.anchors = anchors; p.gt_cls = gt_class_ids; p.gt_boxes = gt_boxes; p.keys = rand_keys;
    p.match = rpn_match; p.bbox = rpn_bbox;
    p.colmax = (unsigned long long*)workspace;
    p.arg = (int*)((char*)workspace + (size_t)d->B * d->G * sizeof(unsigned long long));
    p.B = d->B; p.A = d->A; p.G = d->G; p.n_train = d->n_train;
    for (int i = 0; i < 4; ++i) p.sd[i] = d->bbox_std_dev[i];
    if (hipMemsetAsync(p.colmax, 0, (size_t)d->B * d->G * sizeof(unsigned long long), s) != hipSuccess) return -4;
    dim3 grid((d->A + 255) / 256, d->B);
    hipLaunchKernelGGL(rpn_colmax_kernel, grid, dim3(256), 0, s, p);
    hipLaunchKernelGGL(rpn_match_kernel, grid, dim3(256), 0, s, p);
    hipLaunchKernelGGL(rpn_sample_kernel, dim3(d->B), dim3(RT_THREADS), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
