// PyramidROIAlign (mrcnn/model.py:428-534) as one HBM-bound gather kernel (and its scatter-add
// adjoint).  The reference splits ROIs by pyramid level, runs tf.image.crop_and_resize per level,
// concatenates and re-sorts; here every output bin picks its level in registers and the result is
// written straight in the original ROI order.
//
// Work mapping: one wave per ROW of output bins (roi, py), walking px (round 3; one wave per bin before).  The C channels of a bin are contiguous in
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

// Round 3: a wave owns one ROW of bins (roi, py) and walks px: the ROI's level (logf / sqrtf), the row's y sample and every
// integer division are done once per P bins instead of once per bin -- with one wave per bin those ~300 instructions were
// what a 1 KiB bin cost (1 300 SIMD cycles per bin in the isolated launch), not its five memory instructions.  The float
// expressions are the ones of the per-bin form (same operands, same order): results are bit-identical.
template <bool BWD>
__global__ __launch_bounds__(256) void roialign_kernel(const RoiArgs p) {
    const int lane = threadIdx.x & 63;
    const unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6);               // (roi, py); the host checks B * R * P * P * C < 2^40, rows < 2^31
    const unsigned nrows = (unsigned)p.B * (unsigned)p.R * (unsigned)p.P;
    if (row >= nrows) return;
    const unsigned roi = row / (unsigned)p.P;                                // b * R + r
    const int py = (int)(row - roi * (unsigned)p.P);
    const int b = (int)(roi / (unsigned)p.R);
    const float* bx = p.boxes + (int64_t)roi * 4;
    const float y1 = bx[0], x1 = bx[1], y2 = bx[2], x2 = bx[3];
    const int lvl = roi_level(y1, x1, y2, x2, p.image_area);
    const int li = lvl - 2;
    const int H = p.H[li], W = p.W[li];
    if (!BWD && p.level_out && py == 0 && lane == 0) p.level_out[roi] = lvl;

    float in_y, ws = 0.f, x0;
    if (p.P > 1) {
        const float hs = (y2 - y1) * (float)(H - 1) / (float)(p.P - 1);
        ws = (x2 - x1) * (float)(W - 1) / (float)(p.P - 1);
        in_y = y1 * (float)(H - 1) + (float)py * hs;
        x0 = x1 * (float)(W - 1);
    } else {
        in_y = 0.5f * (y1 + y2) * (float)(H - 1);
        x0 = 0.5f * (x1 + x2) * (float)(W - 1);
    }
    const bool y_inside = !(in_y < 0.f || in_y > (float)(H - 1));
    const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
    const float yl = in_y - (float)top;
    const int c4n = p.C >> 2;
    const int64_t bin0 = (int64_t)row * p.P;
    if (!BWD) {
        const float* __restrict__ base = p.fm[li] + (int64_t)b * H * W * p.C;
        const float* __restrict__ rtop = base + (int64_t)top * W * p.C;
        const float* __restrict__ rbot = base + (int64_t)bot * W * p.C;
        float* __restrict__ orow = p.out + bin0 * p.C;
#pragma unroll 2
        for (int px = 0; px < p.P; ++px) {
            const float in_x = p.P > 1 ? x0 + (float)px * ws : x0;
            f32x4* o = (f32x4*)(orow + (int64_t)px * p.C);
            if (!y_inside || in_x < 0.f || in_x > (float)(W - 1)) {
                for (int c = lane; c < c4n; c += 64) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
            const float xl = in_x - (float)lef;
            const f32x4* tl = (const f32x4*)(rtop + (int64_t)lef * p.C);
            const f32x4* tr = (const f32x4*)(rtop + (int64_t)rig * p.C);
            const f32x4* bl = (const f32x4*)(rbot + (int64_t)lef * p.C);
            const f32x4* br = (const f32x4*)(rbot + (int64_t)rig * p.C);
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
        }
    } else {
        if (!y_inside) return;
        float* base = p.dfm[li] + (int64_t)b * H * W * p.C;
        float* rtop = base + (int64_t)top * W * p.C;
        float* rbot = base + (int64_t)bot * W * p.C;
        const float* grow = p.dout + bin0 * p.C;
        for (int px = 0; px < p.P; ++px) {
            const float in_x = p.P > 1 ? x0 + (float)px * ws : x0;
            if (in_x < 0.f || in_x > (float)(W - 1)) continue;
            const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
            const float xl = in_x - (float)lef;
            float* tl = rtop + (int64_t)lef * p.C;
            float* tr = rtop + (int64_t)rig * p.C;
            float* bl = rbot + (int64_t)lef * p.C;
            float* br = rbot + (int64_t)rig * p.C;
            const float* g = grow + (int64_t)px * p.C;
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
    const int64_t nrows = (int64_t)a.B * a.R * a.P;
    if (nrows >= (1LL << 31)) return MRCNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(roialign_kernel<false>, dim3((unsigned)cdiv64(nrows, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_roialign_bwd(const mrcnn_roialign_desc* d, const float* boxes, const float* dout,
                                  float* dfm2, float* dfm3, float* dfm4, float* dfm5, void* stream) {
    RoiArgs a = {};
    int rc = fill_roi_args(d, a);
    if (rc) return rc;
    if (!boxes || !dout || !dfm2 || !dfm3 || !dfm4 || !dfm5) return MRCNN_ERR_ARG;
    a.boxes = boxes; a.dout = dout; a.dfm[0] = dfm2; a.dfm[1] = dfm3; a.dfm[2] = dfm4; a.dfm[3] = dfm5;
    const int64_t nrows = (int64_t)a.B * a.R * a.P;
    if (nrows >= (1LL << 31)) return MRCNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(roialign_kernel<true>, dim3((unsigned)cdiv64(nrows, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// 16-bit form (BASELINE configs[4]: 16-bit activations): the pyramid levels and the pooled output are float16 / bfloat16,
// the interpolation runs in float32 and is rounded once.  Same wave-per-bin mapping; a bin's C channels are C * 2 bytes
// (256 channels: 64 lanes x 8 bytes), so the gather moves half the bytes of the float32 form and the cast passes around it
// (float32 copies of the pyramid, pooled output back to 16 bits, the gradient up to float32) disappear.  The adjoint reads
// the 16-bit gradient (scaled by the float16 loss scale; `mul` = 1 / scale divides it out) and adds float32 atomics into
// the float32 pyramid gradients, zero rows skipped as in the float32 form.
struct RoiH16Args {
    const float* boxes; const void* fm[4]; void* out;
    const void* dout; float* dfm[4];
    int B, R, P, C;
    int H[4], W[4];
    float image_area, mul;
};

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roialign_h16_kernel(const RoiH16Args p) {      // a wave per row of bins, as roialign_kernel
    typedef T t4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned nrows = (unsigned)p.B * (unsigned)p.R * (unsigned)p.P;
    if (row >= nrows) return;
    const unsigned roi = row / (unsigned)p.P;
    const int py = (int)(row - roi * (unsigned)p.P);
    const int b = (int)(roi / (unsigned)p.R);
    const float* bx = p.boxes + (int64_t)roi * 4;
    const float y1 = bx[0], x1 = bx[1], y2 = bx[2], x2 = bx[3];
    const int li = roi_level(y1, x1, y2, x2, p.image_area) - 2;
    const int H = p.H[li], W = p.W[li];
    float in_y, ws = 0.f, x0;
    if (p.P > 1) {
        const float hs = (y2 - y1) * (float)(H - 1) / (float)(p.P - 1);
        ws = (x2 - x1) * (float)(W - 1) / (float)(p.P - 1);
        in_y = y1 * (float)(H - 1) + (float)py * hs;
        x0 = x1 * (float)(W - 1);
    } else {
        in_y = 0.5f * (y1 + y2) * (float)(H - 1);
        x0 = 0.5f * (x1 + x2) * (float)(W - 1);
    }
    const bool y_inside = !(in_y < 0.f || in_y > (float)(H - 1));
    const int top = y_inside ? (int)floorf(in_y) : 0, bot = y_inside ? (int)ceilf(in_y) : 0;
    const float yl = in_y - (float)top;
    const int c4n = p.C >> 2;
    const int64_t bin0 = (int64_t)row * p.P;
    if (!BWD) {
        const T* __restrict__ base = (const T*)p.fm[li] + (int64_t)b * H * W * p.C;
        const T* __restrict__ rtop = base + (int64_t)top * W * p.C;
        const T* __restrict__ rbot = base + (int64_t)bot * W * p.C;
        T* __restrict__ orow = (T*)p.out + bin0 * p.C;
#pragma unroll 2
        for (int px = 0; px < p.P; ++px) {
            const float in_x = p.P > 1 ? x0 + (float)px * ws : x0;
            t4* o = (t4*)(orow + (int64_t)px * p.C);
            if (!y_inside || in_x < 0.f || in_x > (float)(W - 1)) {
                for (int c = lane; c < c4n; c += 64) o[c] = (t4){(T)0.f, (T)0.f, (T)0.f, (T)0.f};
                continue;
            }
            const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
            const float xl = in_x - (float)lef;
            const t4* tl = (const t4*)(rtop + (int64_t)lef * p.C);
            const t4* tr = (const t4*)(rtop + (int64_t)rig * p.C);
            const t4* bl = (const t4*)(rbot + (int64_t)lef * p.C);
            const t4* br = (const t4*)(rbot + (int64_t)rig * p.C);
            for (int c = lane; c < c4n; c += 64) {
                const t4 a = tl[c], bq = tr[c], cq = bl[c], dq = br[c];
                t4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = (float)a[e], cv = (float)cq[e];
                    const float t = av + ((float)bq[e] - av) * xl;
                    const float u = cv + ((float)dq[e] - cv) * xl;
                    r[e] = (T)(t + (u - t) * yl);
                }
                o[c] = r;
            }
        }
    } else {
        if (!y_inside) return;
        float* base = p.dfm[li] + (int64_t)b * H * W * p.C;
        float* rtop = base + (int64_t)top * W * p.C;
        float* rbot = base + (int64_t)bot * W * p.C;
        const T* grow = (const T*)p.dout + bin0 * p.C;
        for (int px = 0; px < p.P; ++px) {
            const float in_x = p.P > 1 ? x0 + (float)px * ws : x0;
            if (in_x < 0.f || in_x > (float)(W - 1)) continue;
            const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
            const float xl = in_x - (float)lef;
            float* tl = rtop + (int64_t)lef * p.C;
            float* tr = rtop + (int64_t)rig * p.C;
            float* bl = rbot + (int64_t)lef * p.C;
            float* br = rbot + (int64_t)rig * p.C;
            const T* g = grow + (int64_t)px * p.C;
            const float wtl = (1.f - yl) * (1.f - xl), wtr = (1.f - yl) * xl, wbl = yl * (1.f - xl), wbr = yl * xl;
            for (int c = lane; c < p.C; c += 64) {
                const float gv = (float)g[c] * p.mul;
                if (gv == 0.f) continue;
                if (wtl != 0.f) atomicAdd(tl + c, gv * wtl);
                if (wtr != 0.f) atomicAdd(tr + c, gv * wtr);
                if (wbl != 0.f) atomicAdd(bl + c, gv * wbl);
                if (wbr != 0.f) atomicAdd(br + c, gv * wbr);
            }
        }
    }
}

static int fill_roi_h16_args(const mrcnn_roialign_desc* d, RoiH16Args& a) {
    if (!d || d->B <= 0 || d->R <= 0 || d->P <= 0 || d->C <= 0 || (d->C & 3)) return MRCNN_ERR_ARG;
    a.B = d->B; a.R = d->R; a.P = d->P; a.C = d->C; a.image_area = d->image_area;
    for (int i = 0; i < 4; ++i) {
        if (d->H[i] <= 0 || d->W[i] <= 0) return MRCNN_ERR_ARG;
        a.H[i] = d->H[i]; a.W[i] = d->W[i];
    }
    return MRCNN_OK;
}

extern "C" int mrcnn_roialign_fwd_h16(const mrcnn_roialign_desc* d, int dtype, const float* boxes, const void* fm2, const void* fm3,
                                      const void* fm4, const void* fm5, void* out, void* stream) {
    RoiH16Args a = {};
    int rc = fill_roi_h16_args(d, a);
    if (rc) return rc;
    if (!boxes || !fm2 || !fm3 || !fm4 || !fm5 || !out || (dtype != MRCNN_DTYPE_F16 && dtype != MRCNN_DTYPE_BF16)) return MRCNN_ERR_ARG;
    a.boxes = boxes; a.fm[0] = fm2; a.fm[1] = fm3; a.fm[2] = fm4; a.fm[3] = fm5; a.out = out;
    const int64_t nbins = (int64_t)a.B * a.R * a.P;          // rows of bins: a wave each
    if (nbins >= (1LL << 31)) return MRCNN_ERR_UNSUPPORTED;
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL((roialign_h16_kernel<_Float16, false>), dim3((unsigned)cdiv64(nbins, 4)), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((roialign_h16_kernel<__bf16, false>), dim3((unsigned)cdiv64(nbins, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}

extern "C" int mrcnn_roialign_bwd_h16(const mrcnn_roialign_desc* d, int dtype, const float* boxes, const void* dout, float multiplier,
                                      float* dfm2, float* dfm3, float* dfm4, float* dfm5, void* stream) {
    RoiH16Args a = {};
    int rc = fill_roi_h16_args(d, a);
    if (rc) return rc;
    if (!boxes || !dout || !dfm2 || !dfm3 || !dfm4 || !dfm5 || (dtype != MRCNN_DTYPE_F16 && dtype != MRCNN_DTYPE_BF16)) return MRCNN_ERR_ARG;
    a.boxes = boxes; a.dout = dout; a.mul = multiplier; a.dfm[0] = dfm2; a.dfm[1] = dfm3; a.dfm[2] = dfm4; a.dfm[3] = dfm5;
    const int64_t nbins = (int64_t)a.B * a.R * a.P;          // rows of bins: a wave each
    if (nbins >= (1LL << 31)) return MRCNN_ERR_UNSUPPORTED;
    if (dtype == MRCNN_DTYPE_F16)
        hipLaunchKernelGGL((roialign_h16_kernel<_Float16, true>), dim3((unsigned)cdiv64(nbins, 4)), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((roialign_h16_kernel<__bf16, true>), dim3((unsigned)cdiv64(nbins, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Gather form of the adjoint (mrcnn_roialign_bwd_gather).  In training every one of the B*R class-head ROIs
// carries gradient and they pile up on a few thousand pyramid pixels (512 ROIs per image on 64^2 + 32^2 + 16^2
// pixels at 256^2 inputs): the scatter form issues 4 * 256 float atomics per bin (103 M per step) with ~20-fold
// average and several-hundred-fold peak contention per address -- 1.3 ms alone, 2.5 ms beside the mask head, and
// on the critical path of the step.
// Bilinear sampling is separable, so the adjoint can be read off per destination pixel without any sorting:
// pixel (y, x) of level l receives  sum over ROIs of level l, over the samples (py, px) whose floor or ceil row /
// column is (y, x), of  wy * wx * dout[roi, py, px, :].  One wave per pixel: lanes test 64 ROIs of the pixel's
// image at a time (level, then which of the P sample rows / columns touch y / x: two P-bit masks), the hits are
// walked wave-uniformly and their rows accumulated in registers (lane = 4 channels), one row of atomics at the
// end (the mask head's adjoint may be adding to the same maps on another stream).  Weights are the very factors of
// the scatter form ((1 - yl) for the floor row, yl for the ceil row), so the two forms differ by summation order only.
// A pixel under hundreds of ROIs is a long dependent chain if rows are fetched one by one, so the hits of 64 ROIs are
// first staged as (row, weight) entries in LDS (lane-parallel, fixed (ROI, py, px) order) and then read eight rows at a
// time; ROI_GATHER_SPLIT waves share a pixel (every 4th block of 64 ROIs each).
struct RoiGatherArgs {
    RoiArgs r;
    int lvl_off[5];                      // first wave (pixel) id of P2..P5, [4] = number of pixels
};

#define ROI_GATHER_CAP 256        // (row, weight) entries staged per wave and round
#define ROI_GATHER_SPLIT 4        // waves per pixel: each takes every 4th block of 64 ROIs (hot pixels set the tail)

__global__ __launch_bounds__(256) void roialign_bwd_gather_kernel(const RoiGatherArgs q) {
    __shared__ int q_row[4][ROI_GATHER_CAP];
    __shared__ float q_w[4][ROI_GATHER_CAP];
    const RoiArgs& p = q.r;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wid = blockIdx.x * 4 + wv;
    const int pix = wid / ROI_GATHER_SPLIT, part = wid - pix * ROI_GATHER_SPLIT;
    if (pix >= q.lvl_off[4]) return;
    int li = 0;
    while (li < 3 && pix >= q.lvl_off[li + 1]) ++li;
    const int H = p.H[li], W = p.W[li];
    const int local = pix - q.lvl_off[li];
    const int b = local / (H * W);
    const int yx = local - b * H * W;
    const int y = yx / W, x = yx - y * W;
    const int P = p.P;
    const float fh = (float)(H - 1), fw = (float)(W - 1);
    int* qr = q_row[wv];
    float* qw = q_w[wv];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = part * 64; r0 < p.R; r0 += 64 * ROI_GATHER_SPLIT) {
        // ---- lanes test 64 ROIs: level, then which sample rows / columns have (y, x) as floor or ceil
        const int r = r0 + lane;
        unsigned my = 0u, mx = 0u;
        float y0 = 0.f, hs = 0.f, x0 = 0.f, ws = 0.f;
        if (r < p.R) {
            const float* bx = p.boxes + ((int64_t)b * p.R + r) * 4;
            const float y1 = bx[0], x1 = bx[1], y2 = bx[2], x2 = bx[3];
            if (roi_level(y1, x1, y2, x2, p.image_area) - 2 == li) {
                if (P > 1) {
                    hs = (y2 - y1) * fh / (float)(P - 1);
                    ws = (x2 - x1) * fw / (float)(P - 1);
                    y0 = y1 * fh;
                    x0 = x1 * fw;
                } else {
                    y0 = 0.5f * (y1 + y2) * fh;
                    x0 = 0.5f * (x1 + x2) * fw;
                }
                for (int k = 0; k < P; ++k) {
                    const float in_y = P > 1 ? y0 + (float)k * hs : y0;
                    const float in_x = P > 1 ? x0 + (float)k * ws : x0;
                    if (!(in_y < 0.f || in_y > fh) && ((int)floorf(in_y) == y || (int)ceilf(in_y) == y)) my |= 1u << k;
                    if (!(in_x < 0.f || in_x > fw) && ((int)floorf(in_x) == x || (int)ceilf(in_x) == x)) mx |= 1u << k;
                }
            }
        }
        const int nx = __popc(mx);
        const int cnt = __popc(my) * nx;
        if (__ballot(cnt != 0) == 0ull) continue;
        int inc = cnt;                                            // inclusive prefix over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        const int total = __builtin_amdgcn_readlane(inc, 63);
        const int first = inc - cnt;                              // this lane's entries: [first, first + cnt) in (py, px) order
        const int row0 = (((int)b * p.R + r) * P) * P;
        for (int base = 0; base < total; base += ROI_GATHER_CAP) {
            // ---- stage this round's (row, weight) entries in LDS (lane-parallel), fixed (ROI, py, px) order
            if (cnt != 0 && first < base + ROI_GATHER_CAP && first + cnt > base) {
                int e = first;
                for (unsigned ym = my; ym; ym &= ym - 1) {
                    if (e + nx <= base || e >= base + ROI_GATHER_CAP) { e += nx; continue; }
                    const int py = __ffs((int)ym) - 1;
                    const float in_y = P > 1 ? y0 + (float)py * hs : y0;
                    const int top = (int)floorf(in_y);
                    const float yl = in_y - (float)top;
                    const float wy = top == y ? 1.f - yl : yl;   // floor row, else the ceil row
                    for (unsigned xm = mx; xm; xm &= xm - 1, ++e) {
                        if (e < base || e >= base + ROI_GATHER_CAP) continue;
                        const int px = __ffs((int)xm) - 1;
                        const float in_x = P > 1 ? x0 + (float)px * ws : x0;
                        const int lef = (int)floorf(in_x);
                        const float xl = in_x - (float)lef;
                        const float wx = lef == x ? 1.f - xl : xl;
                        qr[e - base] = row0 + py * P + px;
                        qw[e - base] = wy * wx;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- eight rows in flight per batch
            const int n = total - base < ROI_GATHER_CAP ? total - base : ROI_GATHER_CAP;
            for (int i = 0; i < n; i += 8) {
                f32x4 g[8]; float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = i + u < n ? i + u : n - 1;
                    w[u] = i + u < n ? qw[j] : 0.f;
                    g[u] = *(const f32x4*)(p.dout + (int64_t)qr[j] * 256 + lane * 4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += w[u] * g[u][e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    float* dst = p.dfm[li] + (int64_t)local * 256 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (acc[e] != 0.f) atomicAdd(dst + e, acc[e]);
}

extern "C" int mrcnn_roialign_bwd_gather(const mrcnn_roialign_desc* d, const float* boxes, const float* dout, float* dfm2,
                                         float* dfm3, float* dfm4, float* dfm5, void* stream) {
    RoiGatherArgs q = {};
    int rc = fill_roi_args(d, q.r);
    if (rc) return rc;
    if (!boxes || !dout || !dfm2 || !dfm3 || !dfm4 || !dfm5) return MRCNN_ERR_ARG;
    if (d->C != 256 || d->P > 32 || (reinterpret_cast<uintptr_t>(dout) & 15)) return MRCNN_ERR_UNSUPPORTED;
    long long npix = 0;
    for (int i = 0; i < 4; ++i) {
        q.lvl_off[i] = (int)npix;
        npix += (long long)d->B * d->H[i] * d->W[i];
    }
    if (npix >= (1ll << 30) || (long long)d->B * d->R * d->P * d->P >= (1ll << 30)) return MRCNN_ERR_UNSUPPORTED;
    q.lvl_off[4] = (int)npix;
    q.r.boxes = boxes; q.r.dout = dout; q.r.dfm[0] = dfm2; q.r.dfm[1] = dfm3; q.r.dfm[2] = dfm4; q.r.dfm[3] = dfm5;
    hipLaunchKernelGGL(roialign_bwd_gather_kernel, dim3((unsigned)cdiv64(npix * ROI_GATHER_SPLIT, 4)), dim3(256), 0, (hipStream_t)stream, q);
    return mrcnn_launch_status();
}
