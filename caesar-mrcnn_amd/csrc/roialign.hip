// PyramidROIAlign (mrcnn/model.py:428-534) as one HBM-bound gather kernel (and its scatter-add
// adjoint).  The reference splits ROIs by pyramid level, runs tf.image.crop_and_resize per level,
// concatenates and re-sorts; here every output bin picks its level in registers and the result is
// written straight in the original ROI order.
//
// Work mapping: one wave per output bin (roi, py, px).  The C channels of a bin are contiguous in
// NHWC, so the four bilinear corners are four fully coalesced row reads (C=256: 1 KiB = 64 lanes x
// float4) and the bin is one coalesced 1 KiB store.  crop_and_resize semantics ([3P] TF 1.13
// CropAndResize CPU functor): in = lo*(D-1) + i*(hi-lo)*(D-1)/(P-1) (P>1) or 0.5*(lo+hi)*(D-1);
// outside [0, D-1] -> 0; lerp x first, then y.
#include "common.h"

struct RoiArgs {
    const float* boxes; const float* fm[4]; float* out; int32_t* level_out;
    const float* dout; float* dfm[4];
    int B, R, P, C;
    int H[4], W[4];
    float image_area;
};

__device__ __forceinline__ int roi_level(float y1, float x1, float y2, float x2, float image_area) {
    // log2_graph: tf.log(x) / tf.log(2.0) in float32; tf.round = half-to-even
    float h = y2 - y1, w = x2 - x1;
    float v = logf(sqrtf(h * w) / (224.0f / sqrtf(image_area))) / logf(2.0f);
    if (!(fabsf(v) <= 1e30f)) return 2;          // -inf / nan -> int32 min in TF -> clamps to 2
    int l = 4 + (int)rintf(v);
    return l < 2 ? 2 : (l > 5 ? 5 : l);
}

template <bool BWD>
__global__ __launch_bounds__(256) void roialign_kernel(const RoiArgs p) {
    const int lane = threadIdx.x & 63;
    const int64_t bin = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nbins = (int64_t)p.B * p.R * p.P * p.P;
    if (bin >= nbins) return;
    const int px = (int)(bin % p.P);
    const int py = (int)((bin / p.P) % p.P);
    const int64_t roi = bin / (p.P * p.P);       // b*R + r
    const int b = (int)(roi / p.R);
    const float* bx = p.boxes + roi * 4;
    const float y1 = bx[0], x1 = bx[1], y2 = bx[2], x2 = bx[3];
    const int lvl = roi_level(y1, x1, y2, x2, p.image_area);
    const int li = lvl - 2;
    const int H = p.H[li], W = p.W[li];
    if (!BWD && p.level_out && py == 0 && px == 0 && lane == 0) p.level_out[roi] = lvl;

    float in_y, in_x;
    if (p.P > 1) {
        const float hs = (y2 - y1) * (float)(H - 1) / (float)(p.P - 1);
        const float ws = (x2 - x1) * (float)(W - 1) / (float)(p.P - 1);
        in_y = y1 * (float)(H - 1) + (float)py * hs;
        in_x = x1 * (float)(W - 1) + (float)px * ws;
    } else {
        in_y = 0.5f * (y1 + y2) * (float)(H - 1);
        in_x = 0.5f * (x1 + x2) * (float)(W - 1);
    }
    const bool inside = !(in_y < 0.f || in_y > (float)(H - 1) || in_x < 0.f || in_x > (float)(W - 1));
    const int c4n = p.C >> 2;
    if (!BWD) {
        f32x4* o = (f32x4*)(p.out + bin * p.C);
        if (!inside) {
            for (int c = lane; c < c4n; c += 64) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            return;
        }
        const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
        const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
        const float yl = in_y - (float)top, xl = in_x - (float)lef;
        const float* base = p.fm[li] + (int64_t)b * H * W * p.C;
        const f32x4* tl = (const f32x4*)(base + ((int64_t)top * W + lef) * p.C);
        const f32x4* tr = (const f32x4*)(base + ((int64_t)top * W + rig) * p.C);
        const f32x4* bl = (const f32x4*)(base + ((int64_t)bot * W + lef) * p.C);
        const f32x4* br = (const f32x4*)(base + ((int64_t)bot * W + rig) * p.C);
        for (int c = lane; c < c4n; c += 64) {
            f32x4 a = tl[c], bq = tr[c], cq = bl[c], dq = br[c], r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = a[e] + (bq[e] - a[e]) * xl;
                float u = cq[e] + (dq[e] - cq[e]) * xl;
                r[e] = t + (u - t) * yl;
            }
            o[c] = r;
        }
    } else {
        if (!inside) return;
        const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
        const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
        const float yl = in_y - (float)top, xl = in_x - (float)lef;
        float* base = p.dfm[li] + (int64_t)b * H * W * p.C;
        float* tl = base + ((int64_t)top * W + lef) * p.C;
        float* tr = base + ((int64_t)top * W + rig) * p.C;
        float* bl = base + ((int64_t)bot * W + lef) * p.C;
        float* br = base + ((int64_t)bot * W + rig) * p.C;
        const float* g = p.dout + bin * p.C;
        const float wtl = (1.f - yl) * (1.f - xl), wtr = (1.f - yl) * xl, wbl = yl * (1.f - xl), wbr = yl * xl;
        // one dword per lane per atomic instruction: 256 contiguous bytes per wave-instruction
        // exact zeros are skipped: ROIs that carry no gradient (zero-padded / non-positive rows of the
        // mask head) and integer-aligned samples (weight 0 corners) would only add 0.0f -- and the
        // padded ROIs all hit pixel (0,0) of P2, which serialises the atomics on one row
        for (int c = lane; c < p.C; c += 64) {
            float gv = g[c];
            if (gv == 0.f) continue;
            if (wtl != 0.f) atomicAdd(tl + c, gv * wtl);
            if (wtr != 0.f) atomicAdd(tr + c, gv * wtr);
            if (wbl != 0.f) atomicAdd(bl + c, gv * wbl);
            if (wbr != 0.f) atomicAdd(br + c, gv * wbr);
        }
    }
}

static int fill_roi_args(const mrcnn_roialign_desc* d, RoiArgs& a) {
    if (!d || d->B <= 0 || d->R <= 0 || d->P <= 0 || d->C <= 0 || (d->C & 3)) return MRCNN_ERR_ARG;
    a.B = d->B; a.R = d->R; a.P = d->P; a.C = d->C; a.image_area = d->image_area;
    for (int i = 0; i < 4; ++i) {
        if (d->H[i] <= 0 || d->W[i] <= 0) return MRCNN_ERR_ARG;
        a.H[i] = d->H[i]; a.W[i] = d->W[i];
    }
    return MRCNN_OK;
}

extern "C" int mrcnn_roialign_fwd(const mrcnn_roialign_desc* d, const float* boxes, const float* fm2,
                                  const float* fm3, const float* fm4, const float* fm5, float* out,
                                  int32_t* level_out, void* stream) {
    RoiArgs a = {};
    int rc = fill_roi_args(d, a);
    if (rc) return rc;
    if (!boxes || !fm2 || !fm3 || !fm4 || !fm5 || !out) return MRCNN_ERR_ARG;
    a.boxes = boxes; a.fm[0] = fm2; a.fm[1] = fm3; a.fm[2] = fm4; a.fm[3] = fm5; a.out = out; a.level_out = level_out;
    int64_t nbins = (int64_t)a.B * a.R * a.P * a.P;
    hipLaunchKernelGGL(roialign_kernel<false>, dim3((unsigned)cdiv64(nbins, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_roialign_bwd(const mrcnn_roialign_desc* d, const float* boxes, const float* dout,
                                  float* dfm2, float* dfm3, float* dfm4, float* dfm5, void* stream) {
    RoiArgs a = {};
    int rc = fill_roi_args(d, a);
    if (rc) return rc;
    if (!boxes || !dout || !dfm2 || !dfm3 || !dfm4 || !dfm5) return MRCNN_ERR_ARG;
    a.boxes = boxes; a.dout = dout; a.dfm[0] = dfm2; a.dfm[1] = dfm3; a.dfm[2] = dfm4; a.dfm[3] = dfm5;
    int64_t nbins = (int64_t)a.B * a.R * a.P * a.P;
    hipLaunchKernelGGL(roialign_kernel<true>, dim3((unsigned)cdiv64(nbins, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Gather form of the adjoint (mrcnn_roialign_bwd_sorted).  In training every one of the B*R class-head ROIs
// carries gradient and they pile up on a few thousand pyramid pixels (512 ROIs per image on 64^2 + 32^2 + 16^2
// pixels at 256^2 inputs): the scatter form issues 4 * 256 float atomics per bin (103 M per step) with ~20-fold
// average and several-hundred-fold peak contention per address -- 1.3 ms alone, 2.5 ms beside the mask head.
// Here the (bin, corner) records are bucketed by destination pixel with a counting sort (histogram, scan,
// scatter), then one wave per 32 consecutive records sums  w * dout[bin, :]  in registers and issues one row of
// atomics per run of equal pixels: ~(pixels + chunks) * 256 atomics instead of records * 256.
// Summation order inside a pixel follows the scatter cursor (atomics), so -- exactly like the scatter form -- the
// result is not bitwise reproducible between runs.
struct RoiSortArgs {
    RoiArgs r;
    int* cursor;                         // [npix + 1]: histogram -> exclusive prefix -> bucket ends; [npix] = total
    int* rec_pix; int* rec_bin; float* rec_w;
    int lvl_off[5];                      // first global pixel id of P2..P5, [4] = npix
    int npix, nbins;
};

// the four (pixel, weight) pairs of a bin; false when the sample lies outside the map (no gradient)
__device__ __forceinline__ bool roi_bin_corners(const RoiSortArgs& q, int bin, int pix[4], float w[4]) {
    const RoiArgs& p = q.r;
    const int px = bin % p.P;
    const int py = (bin / p.P) % p.P;
    const int roi = bin / (p.P * p.P);
    const int b = roi / p.R;
    const float* bx = p.boxes + (int64_t)roi * 4;
    const float y1 = bx[0], x1 = bx[1], y2 = bx[2], x2 = bx[3];
    const int li = roi_level(y1, x1, y2, x2, p.image_area) - 2;
    const int H = p.H[li], W = p.W[li];
    float in_y, in_x;
    if (p.P > 1) {
        const float hs = (y2 - y1) * (float)(H - 1) / (float)(p.P - 1);
        const float ws = (x2 - x1) * (float)(W - 1) / (float)(p.P - 1);
        in_y = y1 * (float)(H - 1) + (float)py * hs;
        in_x = x1 * (float)(W - 1) + (float)px * ws;
    } else {
        in_y = 0.5f * (y1 + y2) * (float)(H - 1);
        in_x = 0.5f * (x1 + x2) * (float)(W - 1);
    }
    if (in_y < 0.f || in_y > (float)(H - 1) || in_x < 0.f || in_x > (float)(W - 1) || in_y != in_y || in_x != in_x) return false;
    const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
    const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
    const float yl = in_y - (float)top, xl = in_x - (float)lef;
    const int base = q.lvl_off[li] + b * H * W;
    pix[0] = base + top * W + lef; w[0] = (1.f - yl) * (1.f - xl);
    pix[1] = base + top * W + rig; w[1] = (1.f - yl) * xl;
    pix[2] = base + bot * W + lef; w[2] = yl * (1.f - xl);
    pix[3] = base + bot * W + rig; w[3] = yl * xl;
    return true;
}

__global__ void roi_count_kernel(const RoiSortArgs q) {
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= q.nbins) return;
    int pix[4]; float w[4];
    if (!roi_bin_corners(q, bin, pix, w)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (w[k] != 0.f) atomicAdd(&q.cursor[pix[k]], 1);
}

// exclusive prefix sum of cursor[0..npix) in place, total -> cursor[npix]; one workgroup, 4 entries per thread per pass
__global__ __launch_bounds__(1024) void roi_scan_kernel(int* cursor, int npix) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < npix; base += 4096) {
        const int i0 = base + tid * 4;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < npix ? cursor[i0 + k] : 0;
        const int mine = v[0] + v[1] + v[2] + v[3];
        int inc = mine;                                        // inclusive scan over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        int run = carry_s + woff + inc - mine;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < npix) cursor[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) cursor[npix] = carry_s;
}

__global__ void roi_scatter_kernel(const RoiSortArgs q) {
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= q.nbins) return;
    int pix[4]; float w[4];
    if (!roi_bin_corners(q, bin, pix, w)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (w[k] != 0.f) {
            const int pos = atomicAdd(&q.cursor[pix[k]], 1);
            q.rec_pix[pos] = pix[k]; q.rec_bin[pos] = bin; q.rec_w[pos] = w[k];
        }
}

// C == 256: lane owns channels 4*lane .. 4*lane+3
__global__ __launch_bounds__(256) void roi_gather_kernel(const RoiSortArgs q) {
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = q.cursor[q.npix];
    const int r0 = chunk * 32;
    if (r0 >= total) return;
    const int n = total - r0 < 32 ? total - r0 : 32;
    int my_pix = -1, my_bin = 0; float my_w = 0.f;
    if (lane < n) { my_pix = q.rec_pix[r0 + lane]; my_bin = q.rec_bin[r0 + lane]; my_w = q.rec_w[r0 + lane]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
        int li = 0;
        while (li < 3 && cur >= q.lvl_off[li + 1]) ++li;
        float* dst = q.r.dfm[li] + (int64_t)(cur - q.lvl_off[li]) * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (acc[e] != 0.f) atomicAdd(dst + e, acc[e]);
    };
    for (int i = 0; i < n; i += 4) {
        f32x4 v[4]; int pix[4]; float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = i + u < n ? i + u : n - 1;          // uniform
            const int bin = __builtin_amdgcn_readlane(my_bin, j);
            pix[u] = __builtin_amdgcn_readlane(my_pix, j);
            w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
            v[u] = *(const f32x4*)(q.r.dout + (int64_t)bin * 256 + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u < n) {
                if (pix[u] != cur) {
                    flush();
                    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    cur = pix[u];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += w[u] * v[u][e];
            }
        }
    }
    flush();
}

static bool roi_sorted_plan(const mrcnn_roialign_desc* d, long long& npix, long long& nbins, int lvl_off[5]) {
    if (!d || d->B <= 0 || d->R <= 0 || d->P <= 0 || d->C != 256) return false;
    npix = 0;
    for (int i = 0; i < 4; ++i) {
        if (d->H[i] <= 0 || d->W[i] <= 0) return false;
        lvl_off[i] = (int)npix;
        npix += (long long)d->B * d->H[i] * d->W[i];
    }
    nbins = (long long)d->B * d->R * d->P * d->P;
    if (npix >= (1ll << 30) || nbins * 4 >= (1ll << 30) || nbins * 256 >= (1ll << 40)) return false;
    lvl_off[4] = (int)npix;
    return true;
}

extern "C" size_t mrcnn_roialign_bwd_sorted_workspace(const mrcnn_roialign_desc* d) {
    long long npix, nbins; int off[5];
    if (!roi_sorted_plan(d, npix, nbins, off)) return 0;
    return (size_t)(((npix + 1) * 4 + 255) & ~255ll) + (size_t)nbins * 4 * 12;
}

extern "C" int mrcnn_roialign_bwd_sorted(const mrcnn_roialign_desc* d, const float* boxes, const float* dout, float* dfm2,
                                         float* dfm3, float* dfm4, float* dfm5, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    RoiSortArgs q = {};
    int rc = fill_roi_args(d, q.r);
    if (rc) return rc;
    if (!boxes || !dout || !dfm2 || !dfm3 || !dfm4 || !dfm5) return MRCNN_ERR_ARG;
    long long npix, nbins;
    if (!roi_sorted_plan(d, npix, nbins, q.lvl_off)) return MRCNN_ERR_UNSUPPORTED;
    const size_t need = mrcnn_roialign_bwd_sorted_workspace(d);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) ||
        (reinterpret_cast<uintptr_t>(dout) & 15))
        return MRCNN_ERR_ARG;
    q.r.boxes = boxes; q.r.dout = dout; q.r.dfm[0] = dfm2; q.r.dfm[1] = dfm3; q.r.dfm[2] = dfm4; q.r.dfm[3] = dfm5;
    q.npix = (int)npix; q.nbins = (int)nbins;
    char* ws = (char*)workspace;
    const size_t cur_bytes = (size_t)(((npix + 1) * 4 + 255) & ~255ll);
    q.cursor = (int*)ws;
    q.rec_pix = (int*)(ws + cur_bytes);
    q.rec_bin = q.rec_pix + nbins * 4;
    q.rec_w = (float*)(q.rec_bin + nbins * 4);
    hipStream_t s = (hipStream_t)stream;
    rc = mrcnn_fill_zero(q.cursor, cur_bytes, stream);
    if (rc) return rc;
    const unsigned gb = (unsigned)cdiv64(nbins, 256);
    hipLaunchKernelGGL(roi_count_kernel, dim3(gb), dim3(256), 0, s, q);
    hipLaunchKernelGGL(roi_scan_kernel, dim3(1), dim3(1024), 0, s, q.cursor, q.npix);
    hipLaunchKernelGGL(roi_scatter_kernel, dim3(gb), dim3(256), 0, s, q);
    const long long chunks = cdiv64(nbins * 4, 32);
    hipLaunchKernelGGL(roi_gather_kernel, dim3((unsigned)cdiv64(chunks, 4)), dim3(256), 0, s, q);
    return mrcnn_launch_status();
}
