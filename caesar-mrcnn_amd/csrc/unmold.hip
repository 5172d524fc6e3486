// detect() post-processing on the device: what MaskRCNN.unmold_detections does per detection on the host
// (mrcnn/model.py:2607-2619 -> utils.unmold_mask, mrcnn/utils.py:629-645 -> utils.resize, :957-978) for ALL detections of
// an image in two launches.  The reference resizes each MH x MW class mask to its integer box with
// skimage.transform.resize(order=1, mode='constant', cval=0, clip=True), thresholds at 0.5 and pastes it into a full-size
// boolean plane; n detections -> [H, W, n].  Here a thread owns one (pixel, detection) cell of that array (or one byte of
// eight detections in the bit-packed form) and evaluates the same float64 expression in the same order as the host
// restatement caesar-mrcnn_amd/utils.py:resize (half-pixel-centre bilinear warp, zero outside the mask, result clipped to
// the mask's [min, max]), so the boolean output is identical, not merely close.  HBM-bound: H*W*n bytes written.
#include "common.h"

// One workgroup per detection: gathers the class channel of its mask into a compact [MH*MW] row (the paste kernel's four
// corner reads then touch 3 KiB per detection instead of a C-strided 12.5 KiB) and records the row's min / max -- the
// bounds skimage's clip=True clamps the interpolated values to.  A NaN anywhere makes both NaN (numpy's min / max
// propagate it and the clipped image is then NaN everywhere: every comparison false).
__global__ __launch_bounds__(256) void unmold_prepare_kernel(const float* __restrict__ mrcnn_mask, const int* __restrict__ dets,
                                                             int n_rows, int MHW, int C, float* __restrict__ compact,
                                                             float* __restrict__ stats) {
    __shared__ float s_min[256], s_max[256];
    __shared__ int s_nan[256];
    const int d = blockIdx.x, tid = threadIdx.x;
    int cls = dets[d * 6 + 4], row = dets[d * 6 + 5];
    cls = cls < 0 ? 0 : (cls >= C ? C - 1 : cls);            // the host wrapper validates; never index outside the tensor
    row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
    const float* src = mrcnn_mask + ((size_t)row * MHW) * C + cls;
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    for (int i = tid; i < MHW; i += 256) {
        const float v = src[(size_t)i * C];
        compact[(size_t)d * MHW + i] = v;
        nan |= (v != v);
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    s_min[tid] = mn; s_max[tid] = mx; s_nan[tid] = nan;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            s_min[tid] = fminf(s_min[tid], s_min[tid + s]);
            s_max[tid] = fmaxf(s_max[tid], s_max[tid + s]);
            s_nan[tid] |= s_nan[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        stats[d * 2 + 0] = s_nan[0] ? NAN : s_min[0];
        stats[d * 2 + 1] = s_nan[0] ? NAN : s_max[0];
    }
}

// utils.resize(mask, (oh, ow)) sampled at output pixel (oy, ox), then `>= 0.5`; float64, the host's operation order.
__device__ __forceinline__ unsigned unmold_sample(const float* __restrict__ m, int MH, int MW, int oh, int ow, int oy, int ox,
                                                  double mn, double mx) {
    const double ry = ((double)oy + 0.5) * ((double)MH / (double)oh) - 0.5;
    const double rx = ((double)ox + 0.5) * ((double)MW / (double)ow) - 0.5;
    const double fy0 = floor(ry), fx0 = floor(rx);
    const int y0 = (int)fy0, x0 = (int)fx0;
    const double fy = ry - fy0, fx = rx - fx0;
    const bool yt = y0 >= 0 && y0 < MH, yb = y0 + 1 >= 0 && y0 + 1 < MH;
    const bool xl = x0 >= 0 && x0 < MW, xr = x0 + 1 >= 0 && x0 + 1 < MW;
    const double tl = (yt && xl) ? (double)m[y0 * MW + x0] : 0.0;
    const double tr = (yt && xr) ? (double)m[y0 * MW + x0 + 1] : 0.0;
    const double bl = (yb && xl) ? (double)m[(y0 + 1) * MW + x0] : 0.0;
    const double br = (yb && xr) ? (double)m[(y0 + 1) * MW + x0 + 1] : 0.0;
    double o = (tl * (1.0 - fx) + tr * fx) * (1.0 - fy) + (bl * (1.0 - fx) + br * fx) * fy;
    o = o < mn ? mn : o;                                     // np.clip(out, min, max); NaN bounds are handled by the caller
    o = o > mx ? mx : o;
    return o >= 0.5 ? 1u : 0u;
}

__device__ __forceinline__ unsigned unmold_cell(const float* __restrict__ compact, const float* __restrict__ stats,
                                                const int* __restrict__ dets, int d, int y, int x, int MH, int MW) {
    const int y1 = dets[d * 6 + 0], x1 = dets[d * 6 + 1], y2 = dets[d * 6 + 2], x2 = dets[d * 6 + 3];
    if (y < y1 || y >= y2 || x < x1 || x >= x2) return 0u;
    const float mn = stats[d * 2 + 0], mx = stats[d * 2 + 1];
    if (mn != mn) return 0u;
    return unmold_sample(compact + (size_t)d * MH * MW, MH, MW, y2 - y1, x2 - x1, y - y1, x - x1, (double)mn, (double)mx);
}

// PACKED = false: out [H, W, n] uint8 (0 / 1), the layout of the reference's result (np.stack(full_masks, axis=-1)).
// PACKED = true:  out [H, W, ceil(n / 8)]: bit (d & 7) of byte d >> 3 = detection d (numpy.unpackbits(..., axis=-1,
//                 count=n, bitorder="little") restores the [H, W, n] array).
template <bool PACKED>
__global__ __launch_bounds__(256) void unmold_paste_kernel(const float* __restrict__ compact, const float* __restrict__ stats,
                                                           const int* __restrict__ dets, int n, int MH, int MW, int H, int W,
                                                           unsigned char* __restrict__ out) {
    const int per = PACKED ? (n + 7) / 8 : n;
    const long long total = (long long)H * W * per;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long pix = i / per;
    const int j = (int)(i - pix * per);
    const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    if (!PACKED) {
        out[i] = (unsigned char)unmold_cell(compact, stats, dets, j, y, x, MH, MW);
    } else {
        unsigned b = 0;
        for (int k = 0; k < 8; ++k) {
            const int d = j * 8 + k;
            if (d < n) b |= unmold_cell(compact, stats, dets, d, y, x, MH, MW) << k;
        }
        out[i] = (unsigned char)b;
    }
}

extern "C" size_t mrcnn_unmold_masks_workspace(int n, int MH, int MW) {
    if (n <= 0 || MH <= 0 || MW <= 0) return 0;
    return ((size_t)n * MH * MW + (size_t)n * 2) * sizeof(float);
}

extern "C" int mrcnn_unmold_masks(const float* mrcnn_mask, int n_rows, int MH, int MW, int C, const int32_t* dets, int n,
                                  int H, int W, int packed, void* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (n == 0) return MRCNN_OK;
    if (!mrcnn_mask || !dets || !out || n < 0 || n_rows <= 0 || MH <= 0 || MW <= 0 || C <= 0 || H <= 0 || W <= 0)
        return MRCNN_ERR_ARG;
    if (!workspace || workspace_bytes < mrcnn_unmold_masks_workspace(n, MH, MW)) return MRCNN_ERR_WORKSPACE;
    float* compact = (float*)workspace;
    float* stats = compact + (size_t)n * MH * MW;
    hipLaunchKernelGGL(unmold_prepare_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, mrcnn_mask, dets, n_rows,
                       MH * MW, C, compact, stats);
    const long long total = (long long)H * W * (packed ? (n + 7) / 8 : n);
    const long long blocks = cdiv64(total, 256);
    if (blocks > 0x7fffffffLL) return MRCNN_ERR_ARG;
    if (packed)
        hipLaunchKernelGGL(unmold_paste_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, compact, stats,
                           dets, n, MH, MW, H, W, (unsigned char*)out);
    else
        hipLaunchKernelGGL(unmold_paste_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, compact, stats,
                           dets, n, MH, MW, H, W, (unsigned char*)out);
    return mrcnn_launch_status();
}
