// DetectionTargetLayer (mrcnn/model.py:570-763): per image, drop zero-padded proposals / GT, IoU
// matrix, positive (IoU >= 0.5) / negative (< 0.5, no crowd) split, random subsample to the 1:2 quota,
// GT assignment, box-refinement targets and 28x28 mask targets (tf.image.crop_and_resize + round).
//
// tf.random.shuffle is replaced by caller-supplied uniform keys (one per proposal row): candidates are
// taken in increasing key order, ties by lower row.  One 1024-thread workgroup per image does the
// matching and the two key sorts in LDS; a second kernel samples the mask targets.
#include "common.h"

#define DT_CAP 2048        // max proposals per image (POST_NMS_ROIS_TRAINING = 2000)
#define DT_GCAP 512        // max GT instances (MAX_GT_INSTANCES = 300 on the run.py path)
#define DT_THREADS 1024

struct DtArgs {
    const float* proposals; const int32_t* gt_class_ids; const float* gt_boxes; const uint8_t* gt_masks;
    const float* rand_keys;
    float* rois; int32_t* tcls; float* tbbox; float* tmask; int32_t* assign; int32_t* counts;
    int B, R, G, T, MH, MW, mh, mw, pos_quota, mini;
    float neg_r, s0, s1, s2, s3;
};

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int n, int tid, int nthreads) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n >> 1); t += nthreads) {
                int lo = ((t / stride) * stride * 2) + (t % stride);
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                unsigned long long x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(DT_THREADS) void detection_targets_kernel(const DtArgs p) {
    __shared__ unsigned long long pkeys[DT_CAP], nkeys[DT_CAP];
    __shared__ float gbox[DT_GCAP * 4];
    __shared__ int gcls[DT_GCAP], gorig[DT_GCAP];
    __shared__ float cbox[DT_GCAP * 4];
    __shared__ short s_argmax[DT_CAP];
    __shared__ int s_ng, s_nc, s_np, s_nn;
    const int b = blockIdx.x, tid = threadIdx.x, R = p.R, G = p.G, T = p.T;
    const float* props = p.proposals + (int64_t)b * R * 4;

    // ---- compact GT (order preserving): non-zero box, then class > 0 (instances) / < 0 (crowds) ----
    if (tid == 0) {
        int ng = 0, nc = 0;
        for (int g = 0; g < G; ++g) {
            const float* gb = p.gt_boxes + ((int64_t)b * G + g) * 4;
            float s = fabsf(gb[0]) + fabsf(gb[1]) + fabsf(gb[2]) + fabsf(gb[3]);
            if (s == 0.f) continue;                       // trim_zeros_graph
            int c = p.gt_class_ids[(int64_t)b * G + g];
            if (c > 0) {
                gbox[ng * 4] = gb[0]; gbox[ng * 4 + 1] = gb[1]; gbox[ng * 4 + 2] = gb[2]; gbox[ng * 4 + 3] = gb[3];
                gcls[ng] = c; gorig[ng] = g; ++ng;
            } else if (c < 0) {
                cbox[nc * 4] = gb[0]; cbox[nc * 4 + 1] = gb[1]; cbox[nc * 4 + 2] = gb[2]; cbox[nc * 4 + 3] = gb[3];
                ++nc;
            }
        }
        s_ng = ng; s_nc = nc; s_np = 0; s_nn = 0;
    }
    for (int i = tid; i < DT_CAP; i += DT_THREADS) { pkeys[i] = ~0ull; nkeys[i] = ~0ull; }
    __syncthreads();
    const int ng = s_ng, nc = s_nc;

    // ---- overlaps_graph + positive / negative classification ------------------------------------------
    for (int r = tid; r < R; r += DT_THREADS) {
        const float* pb = props + (int64_t)r * 4;
        const float y1 = pb[0], x1 = pb[1], y2 = pb[2], x2 = pb[3];
        if (fabsf(y1) + fabsf(x1) + fabsf(y2) + fabsf(x2) == 0.f) continue;      // zero padding
        const float parea = (y2 - y1) * (x2 - x1);
        float best = -INFINITY;
        int arg = 0;
        for (int g = 0; g < ng; ++g) {
            const float* gb = &gbox[g * 4];
            float iy1 = fmaxf(y1, gb[0]), ix1 = fmaxf(x1, gb[1]);
            float iy2 = fminf(y2, gb[2]), ix2 = fminf(x2, gb[3]);
            float inter = fmaxf(ix2 - ix1, 0.f) * fmaxf(iy2 - iy1, 0.f);
            float garea = (gb[2] - gb[0]) * (gb[3] - gb[1]);
            float iou = inter / (parea + garea - inter);
            if (iou > best) { best = iou; arg = g; }
        }
        float cbest = -INFINITY;
        for (int g = 0; g < nc; ++g) {
            const float* gb = &cbox[g * 4];
            float iy1 = fmaxf(y1, gb[0]), ix1 = fmaxf(x1, gb[1]);
            float iy2 = fminf(y2, gb[2]), ix2 = fminf(x2, gb[3]);
            float inter = fmaxf(ix2 - ix1, 0.f) * fmaxf(iy2 - iy1, 0.f);
            float garea = (gb[2] - gb[0]) * (gb[3] - gb[1]);
            float iou = inter / (parea + garea - inter);
            if (iou > cbest) cbest = iou;
        }
        s_argmax[r] = (short)arg;
        const unsigned kb = __float_as_uint(p.rand_keys[(int64_t)b * R + r]);    // keys are >= 0
        const unsigned long long key = ((unsigned long long)kb << 32) | (unsigned)r;
        if (best >= 0.5f) {
            pkeys[r] = key;
            atomicAdd(&s_np, 1);
        } else if (best < 0.5f && cbest < 0.001f) {
            nkeys[r] = key;
            atomicAdd(&s_nn, 1);
        }
    }
    __syncthreads();
    int n = 1;
    while (n < R) n <<= 1;
    bitonic_sort_u64(pkeys, n, tid, DT_THREADS);
    bitonic_sort_u64(nkeys, n, tid, DT_THREADS);

    int P = s_np < p.pos_quota ? s_np : p.pos_quota;
    int N = (int)(p.neg_r * (float)P) - P;       // tf.cast(r * tf.cast(P, float32), int32) - P
    if (N < 0) N = 0;
    if (N > s_nn) N = s_nn;
    if (P > T) P = T;
    if (P + N > T) N = T - P;
    if (tid == 0) { p.counts[b * 2] = P; p.counts[b * 2 + 1] = N; }

    // ---- outputs ---------------------------------------------------------------------------------------
    for (int t = tid; t < T; t += DT_THREADS) {
        float* ro = p.rois + ((int64_t)b * T + t) * 4;
        float* bo = p.tbbox + ((int64_t)b * T + t) * 4;
        int cls = 0, asg = -1;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        if (t < P + N) {
            const int r = (int)((t < P ? pkeys[t] : nkeys[t - P]) & 0xFFFFFFFFull);
            const float* pb = props + (int64_t)r * 4;
            r0 = pb[0]; r1 = pb[1]; r2 = pb[2]; r3 = pb[3];
            if (t < P) {
                const int g = s_argmax[r];
                const float* gb = &gbox[g * 4];
                cls = gcls[g];
                asg = gorig[g];
                // utils.box_refinement_graph, then / BBOX_STD_DEV
                float h = r2 - r0, w = r3 - r1;
                float cy = r0 + 0.5f * h, cx = r1 + 0.5f * w;
                float gh = gb[2] - gb[0], gw = gb[3] - gb[1];
                float gcy = gb[0] + 0.5f * gh, gcx = gb[1] + 0.5f * gw;
                d0 = ((gcy - cy) / h) / p.s0;
                d1 = ((gcx - cx) / w) / p.s1;
                d2 = logf(gh / h) / p.s2;
                d3 = logf(gw / w) / p.s3;
            }
        }
        ro[0] = r0; ro[1] = r1; ro[2] = r2; ro[3] = r3;
        bo[0] = d0; bo[1] = d1; bo[2] = d2; bo[3] = d3;
        p.tcls[(int64_t)b * T + t] = cls;
        p.assign[(int64_t)b * T + t] = asg;
    }
}

// grid (T, B), 256 threads: mask target of ROI t = crop_and_resize(gt mask of the assigned instance)
__global__ __launch_bounds__(256) void mask_targets_kernel(const DtArgs p) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int npix = p.mh * p.mw;
    float* o = p.tmask + ((int64_t)b * p.T + t) * npix;
    const int g = p.assign[(int64_t)b * p.T + t];
    if (g < 0) {
        for (int i = threadIdx.x; i < npix; i += 256) o[i] = 0.f;
        return;
    }
    const float* ro = p.rois + ((int64_t)b * p.T + t) * 4;
    float y1 = ro[0], x1 = ro[1], y2 = ro[2], x2 = ro[3];
    if (p.mini) {
        const float* gb = p.gt_boxes + ((int64_t)b * p.G + g) * 4;
        float gh = gb[2] - gb[0], gw = gb[3] - gb[1];
        y1 = (y1 - gb[0]) / gh; x1 = (x1 - gb[1]) / gw;
        y2 = (y2 - gb[0]) / gh; x2 = (x2 - gb[1]) / gw;
    }
    const int H = p.MH, W = p.MW;
    const uint8_t* m = p.gt_masks + (int64_t)b * H * W * p.G + g;     // [H, W, G] layout
    const float hs = p.mh > 1 ? (y2 - y1) * (float)(H - 1) / (float)(p.mh - 1) : 0.f;
    const float ws = p.mw > 1 ? (x2 - x1) * (float)(W - 1) / (float)(p.mw - 1) : 0.f;
    for (int i = threadIdx.x; i < npix; i += 256) {
        const int py = i / p.mw, px = i - py * p.mw;
        const float in_y = p.mh > 1 ? y1 * (float)(H - 1) + (float)py * hs : 0.5f * (y1 + y2) * (float)(H - 1);
        const float in_x = p.mw > 1 ? x1 * (float)(W - 1) + (float)px * ws : 0.5f * (x1 + x2) * (float)(W - 1);
        float v = 0.f;
        if (!(in_y < 0.f || in_y > (float)(H - 1) || in_x < 0.f || in_x > (float)(W - 1))) {
            const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
            const int lef = (int)floorf(in_x), rig = (int)ceilf(in_x);
            const float yl = in_y - (float)top, xl = in_x - (float)lef;
            const float tl = m[((int64_t)top * W + lef) * p.G] ? 1.f : 0.f;
            const float tr = m[((int64_t)top * W + rig) * p.G] ? 1.f : 0.f;
            const float bl = m[((int64_t)bot * W + lef) * p.G] ? 1.f : 0.f;
            const float br = m[((int64_t)bot * W + rig) * p.G] ? 1.f : 0.f;
            const float tp = tl + (tr - tl) * xl;
            const float bt = bl + (br - bl) * xl;
            v = tp + (bt - tp) * yl;
        }
        o[i] = rintf(v);                          // tf.round: half to even
    }
}

extern "C" int mrcnn_detection_targets(const mrcnn_dettarget_desc* d, const float* proposals,
                                       const int32_t* gt_class_ids, const float* gt_boxes,
                                       const uint8_t* gt_masks, const float* rand_keys, float* rois,
                                       int32_t* target_class_ids, float* target_bbox, float* target_mask,
                                       int32_t* roi_gt_assignment, int32_t* counts, void* stream) {
    if (!d || !proposals || !gt_class_ids || !gt_boxes || !gt_masks || !rand_keys || !rois ||
        !target_class_ids || !target_bbox || !target_mask || !roi_gt_assignment || !counts)
        return MRCNN_ERR_ARG;
    if (d->B <= 0 || d->R <= 0 || d->R > DT_CAP || d->G <= 0 || d->G > DT_GCAP || d->T <= 0 || d->MH <= 0 ||
        d->MW <= 0 || d->mask_h <= 0 || d->mask_w <= 0 || d->positive_count < 0)
        return MRCNN_ERR_ARG;
    DtArgs a;
    a.proposals = proposals; a.gt_class_ids = gt_class_ids; a.gt_boxes = gt_boxes; a.gt_masks = gt_masks;
    a.rand_keys = rand_keys; a.rois = rois; a.tcls = target_class_ids; a.tbbox = target_bbox;
    a.tmask = target_mask; a.assign = roi_gt_assignment; a.counts = counts;
    a.B = d->B; a.R = d->R; a.G = d->G; a.T = d->T; a.MH = d->MH; a.MW = d->MW; a.mh = d->mask_h; a.mw = d->mask_w;
    a.pos_quota = d->positive_count; a.mini = d->use_mini_mask; a.neg_r = d->negative_ratio_r;
    a.s0 = d->bbox_std_dev[0]; a.s1 = d->bbox_std_dev[1]; a.s2 = d->bbox_std_dev[2]; a.s3 = d->bbox_std_dev[3];
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(detection_targets_kernel, dim3(d->B), dim3(DT_THREADS), 0, s, a);
    hipLaunchKernelGGL(mask_targets_kernel, dim3(d->T, d->B), dim3(256), 0, s, a);
    return mrcnn_launch_status();
}
