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
