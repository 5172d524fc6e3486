// DetectionLayer / refine_detections_graph (mrcnn/model.py:770-909), one 1024-thread workgroup per
// image: class argmax, class-specific box decode, clip to the image window, background/confidence
// filter, per-class greedy NMS (at most max_instances per class), top max_instances by score, zero pad.
//
// The reference loops tf.image.non_max_suppression over the unique class ids with tf.map_fn and
// intersects index sets; here all candidates are sorted once by (score desc, roi index asc) and a
// single greedy scan applies "same class and IoU > thr" suppression with a per-class quota.  Because a
// box can only be suppressed by a higher-ranked box of its own class, the first max_instances boxes
// selected in global score order are exactly the top-k of the union of the per-class results.
#include "common.h"

#define DET_CAP 2048
#define DET_THREADS 1024
#define DET_MAX_CLASSES 256

struct DetArgs {
    const float* rois; const float* probs; const float* deltas; const float* windows; float* det;
    unsigned long long* mask_ws;   // [B, R, nwords]
    int B, R, C, maxi, nwords;
    float minconf, thr, s0, s1, s2, s3;
};

__global__ __launch_bounds__(DET_THREADS) void detection_kernel(const DetArgs p) {
    __shared__ unsigned long long keys[DET_CAP];
    __shared__ float sbox[DET_CAP * 4];
    __shared__ float sscore[DET_CAP];
    __shared__ int scls[DET_CAP];
    __shared__ int cls_cnt[DET_MAX_CLASSES];
    __shared__ int keep[DET_CAP];
    __shared__ int s_nvalid, s_total;
    const int b = blockIdx.x, tid = threadIdx.x, R = p.R, C = p.C;
    const float* win = p.windows + b * 4;
    if (tid == 0) s_nvalid = 0;
    for (int c = tid; c < DET_MAX_CLASSES; c += DET_THREADS) cls_cnt[c] = 0;
    for (int i = tid; i < DET_CAP; i += DET_THREADS) keys[i] = ~0ull;
    __syncthreads();

    // ---- 1. per-ROI class, score, refined box ---------------------------------------------------
    for (int r = tid; r < R; r += DET_THREADS) {
        const float* pr = p.probs + ((int64_t)b * R + r) * C;
        int cls = 0;
        float best = pr[0];
        for (int c = 1; c < C; ++c) {
            float v = pr[c];
            if (v > best) { best = v; cls = c; }
        }
        const float* dl = p.deltas + (((int64_t)b * R + r) * C + cls) * 4;
        decode_clip_box(p.rois + ((int64_t)b * R + r) * 4, dl[0] * p.s0, dl[1] * p.s1, dl[2] * p.s2, dl[3] * p.s3,
                        win[0], win[1], win[2], win[3], &sbox[r * 4]);
        scls[r] = cls;
        sscore[r] = best;
        bool ok = cls > 0;
        if (p.minconf != 0.f) ok = ok && (best >= p.minconf);
        if (ok) {
            unsigned u = __float_as_uint(best);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            keys[r] = ((unsigned long long)(~u) << 32) | (unsigned)r;
            atomicAdd(&s_nvalid, 1);
        }
    }
    __syncthreads();
    const int nvalid = s_nvalid;

    // ---- 2. sort candidates: score descending, roi index ascending ---------------------------------
    int n = 1;
    while (n < R) n <<= 1;
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n >> 1); t += DET_THREADS) {
                int lo = ((t / stride) * stride * 2) + (t % stride);
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                unsigned long long x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
    }

    // ---- 3. suppression bit matrix over sorted positions -----------------------------------------
    const int nw = p.nwords;
    unsigned long long* mask = p.mask_ws + (int64_t)b * R * nw;
    for (int i = tid; i < nvalid; i += DET_THREADS) {
        const int ri = (int)(keys[i] & 0xFFFFFFFFull);
        const int ci = scls[ri];
        for (int w = 0; w < nw; ++w) {
            unsigned long long bits = 0ull;
            if (w >= (i >> 6)) {
                const int j0 = w * 64;
                const int jn = (nvalid - j0) < 64 ? (nvalid - j0) : 64;
                for (int j = 0; j < jn; ++j) {
                    const int jj = j0 + j;
                    if (jj <= i) continue;
                    const int rj = (int)(keys[jj] & 0xFFFFFFFFull);
                    if (scls[rj] == ci && iou_gt(&sbox[ri * 4], &sbox[rj * 4], p.thr)) bits |= 1ull << j;
                }
            }
            mask[(int64_t)i * nw + w] = bits;
        }
    }
    __threadfence_block();
    __syncthreads();

    // ---- 4. greedy scan by wave 0 ----------------------------------------------------------------------
    if (tid < 64) {
        const int lane = tid;
        unsigned long long rem[DET_CAP / 64 / 64 > 0 ? DET_CAP / 64 / 64 : 1];   // words lane + 64*k
        rem[0] = 0ull;
        int total = 0;
        const int nchunks = (nvalid + 63) >> 6;
        for (int c = 0; c < nchunks && total < p.maxi; ++c) {
            const int i = c * 64 + lane;
            unsigned long long diag = (i < nvalid) ? mask[(int64_t)i * nw + c] : 0ull;
            int mycls = (i < nvalid) ? scls[(int)(keys[i] & 0xFFFFFFFFull)] : 0;
            unsigned long long remc = shfl64(rem[0], c & 63);
            const int valid_n = (nvalid - c * 64) < 64 ? (nvalid - c * 64) : 64;
            unsigned long long alive = ~remc;
            if (valid_n < 64) alive &= (1ull << valid_n) - 1ull;
            unsigned long long kept = 0ull;
            for (int t = 0; t < valid_n; ++t) {
                unsigned long long d = shfl64(diag, t);
                int cl = __shfl(mycls, t, 64);
                if (((alive >> t) & 1ull) && total < p.maxi) {
                    // per-class quota of tf.image.non_max_suppression(max_output_size)
                    int cnt = cls_cnt[cl];
                    if (cnt < p.maxi) {
                        if (lane == 0) cls_cnt[cl] = cnt + 1;
                        kept |= 1ull << t;
                        ++total;
                        alive &= ~d;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if ((kept >> lane) & 1ull) {
                int pos = total - __popcll(kept) + __popcll(kept & ((1ull << lane) - 1ull));
                keep[pos] = i;
            }
            unsigned long long k2 = kept;
            while (k2) {
                int t = __ffsll((long long)k2) - 1;
                k2 &= k2 - 1ull;
                const unsigned long long* row = mask + (int64_t)(c * 64 + t) * nw;
                if (lane > c && lane < nw) rem[0] |= row[lane];
            }
        }
        if (lane == 0) s_total = total;
    }
    __syncthreads();
    const int total = s_total;
    for (int q = tid; q < p.maxi; q += DET_THREADS) {
        float* o = p.det + ((int64_t)b * p.maxi + q) * 6;
        if (q < total) {
            const int r = (int)(keys[keep[q]] & 0xFFFFFFFFull);
            o[0] = sbox[r * 4]; o[1] = sbox[r * 4 + 1]; o[2] = sbox[r * 4 + 2]; o[3] = sbox[r * 4 + 3];
            o[4] = (float)scls[r];
            o[5] = sscore[r];
        } else {
            o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.f;
        }
    }
}

extern "C" size_t mrcnn_detection_workspace(const mrcnn_detection_desc* d) {
    if (!d || d->B <= 0 || d->R <= 0) return 0;
    size_t nw = ((size_t)d->R + 63) / 64;
    return (size_t)d->B * d->R * nw * sizeof(unsigned long long) + 256;
}

extern "C" int mrcnn_detection_fwd(const mrcnn_detection_desc* d, const float* rois, const float* probs,
                                   const float* deltas, const float* windows, float* detections,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !rois || !probs || !deltas || !windows || !detections || !workspace) return MRCNN_ERR_ARG;
    // one wave holds the removed-bitmap: R <= 64*64 words... we keep one word per lane => R <= 4096,
    // and the LDS arrays cap R at DET_CAP
    if (d->B <= 0 || d->R <= 0 || d->R > DET_CAP || d->C <= 1 || d->C > DET_MAX_CLASSES || d->max_instances <= 0 ||
        d->max_instances > DET_CAP)
        return MRCNN_ERR_ARG;
    if (workspace_bytes < mrcnn_detection_workspace(d)) return MRCNN_ERR_WORKSPACE;
    DetArgs a;
    a.rois = rois; a.probs = probs; a.deltas = deltas; a.windows = windows; a.det = detections;
    a.mask_ws = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.B = d->B; a.R = d->R; a.C = d->C; a.maxi = d->max_instances; a.nwords = (d->R + 63) / 64;
    a.minconf = d->min_confidence; a.thr = d->nms_threshold;
    a.s0 = d->bbox_std_dev[0]; a.s1 = d->bbox_std_dev[1]; a.s2 = d->bbox_std_dev[2]; a.s3 = d->bbox_std_dev[3];
    hipLaunchKernelGGL(detection_kernel, dim3(d->B), dim3(DET_THREADS), 0, (hipStream_t)stream, a);
    return mrcnn_launch_status();
}
