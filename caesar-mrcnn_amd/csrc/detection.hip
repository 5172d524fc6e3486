// DetectionLayer / refine_detections_graph (mrcnn/model.py:770-909): class argmax, class-specific box decode,
// clip to the image window, background/confidence filter, per-class greedy NMS (at most max_instances per
// class), top max_instances by score, zero pad.
//
// The reference loops tf.image.non_max_suppression over the unique class ids with tf.map_fn and intersects
// index sets; here all candidates are sorted once by (score desc, roi index asc) and a single greedy scan applies
// "same class and IoU > thr" suppression.  Because a box can only be suppressed by a higher-ranked box of its own
// class, the first max_instances boxes selected in global score order are exactly the top-k of the union of the
// per-class results; the per-class quota of non_max_suppression equals that overall limit (both
// DETECTION_MAX_INSTANCES, model.py:838-851, 869-873) and can therefore never bind first.
//
//   K1 detection_prepare_kernel  one 1024-thread workgroup per image: argmax / decode / filter, compaction of the
//        candidates, sort by rank counting (keys are distinct -- the ROI index is part of the key -- so the rank of a
//        key is the number of smaller keys: ~n broadcast LDS reads per thread, no barrier network), sorted boxes /
//        classes / scores to the workspace.
//   K2 detection_mask_kernel     64x64 tiles of the upper-triangular suppression bit matrix over all CUs (with
//        DETECTION_MIN_CONFIDENCE = 0, the repo's setting, ~all 1000 ROIs are candidates: ~0.5 M class+IoU tests,
//        which one workgroup cannot do in less than ~100 us).
//   K3 detection_scan_kernel     one wave per image: chunked greedy scan (scalar unit, alive candidates only,
//        find-first-set + v_readlane), rows of kept boxes fetched sixteen at a time, gather + zero pad.
#include "common.h"

#define DET_CAP 2048
#define DET_THREADS 1024
#define DET_MAX_CLASSES 256

struct DetArgs {
    const float* rois; const float* probs; const float* deltas; const float* windows; float* det;
    unsigned long long* mask_ws;   // [B, R, nwords]
    float* qbox;                   // [B, R, 4]  candidates in sorted order
    int* qcls;                     // [B, R]
    float* qscore;                 // [B, R]
    int* nvalid;                   // [B]
    int B, R, C, maxi, nwords;
    float minconf, thr, s0, s1, s2, s3;
};

__global__ __launch_bounds__(DET_THREADS) void detection_prepare_kernel(const DetArgs p) {
    __shared__ unsigned long long packed[DET_CAP];
    __shared__ float sbox[DET_CAP * 4];
    __shared__ float sscore[DET_CAP];
    __shared__ int scls[DET_CAP];
    __shared__ int s_nvalid;
    const int b = blockIdx.x, tid = threadIdx.x, R = p.R, C = p.C;
    const float* win = p.windows + b * 4;
    if (tid == 0) s_nvalid = 0;
    __syncthreads();

    // ---- per-ROI class, score, refined box; candidates compacted (any order) ---------------------------
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
            packed[atomicAdd(&s_nvalid, 1)] = ((unsigned long long)(~u) << 32) | (unsigned)r;   // score desc, index asc
        }
    }
    __syncthreads();
    const int nvalid = s_nvalid;
    if (tid == 0) p.nvalid[b] = nvalid;

    // ---- rank = number of smaller keys; gather into sorted order ------------------------------------------
    float* qbox = p.qbox + (int64_t)b * R * 4;
    int* qcls = p.qcls + (int64_t)b * R;
    float* qscore = p.qscore + (int64_t)b * R;
    for (int i0 = 0; i0 < nvalid; i0 += DET_THREADS) {
        const int i = i0 + tid;
        const unsigned long long mine = i < nvalid ? packed[i] : 0ull;
        int rank = 0;
        for (int j = 0; j < nvalid; ++j) rank += packed[j] < mine;
        if (i < nvalid) {
            const int r = (int)(mine & 0xFFFFFFFFull);
            qbox[rank * 4 + 0] = sbox[r * 4 + 0]; qbox[rank * 4 + 1] = sbox[r * 4 + 1];
            qbox[rank * 4 + 2] = sbox[r * 4 + 2]; qbox[rank * 4 + 3] = sbox[r * 4 + 3];
            qcls[rank] = scls[r];
            qscore[rank] = sscore[r];
        }
    }
}

// word (row tile br, column tile bc) of the suppression matrix: bit j of row i = same class and IoU > thr, j > i
__global__ __launch_bounds__(64) void detection_mask_kernel(const DetArgs p) {
    const int bc = blockIdx.x, br = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int nvalid = p.nvalid[b];
    if (bc < br || br * 64 >= nvalid || bc * 64 >= nvalid) return;
    __shared__ float cb[64 * 4];
    __shared__ int cc[64];
    const float* qbox = p.qbox + (int64_t)b * p.R * 4;
    const int* qcls = p.qcls + (int64_t)b * p.R;
    const int j = bc * 64 + lane;
    if (j < nvalid) {
        cb[lane * 4 + 0] = qbox[j * 4 + 0]; cb[lane * 4 + 1] = qbox[j * 4 + 1];
        cb[lane * 4 + 2] = qbox[j * 4 + 2]; cb[lane * 4 + 3] = qbox[j * 4 + 3];
        cc[lane] = qcls[j];
    }
    __syncthreads();
    const int i = br * 64 + lane;
    if (i >= nvalid) return;
    float mine[4] = {qbox[i * 4], qbox[i * 4 + 1], qbox[i * 4 + 2], qbox[i * 4 + 3]};
    const int ci = qcls[i];
    const int jn = (nvalid - bc * 64) < 64 ? (nvalid - bc * 64) : 64;
    unsigned long long bits = 0ull;
    for (int t = 0; t < jn; ++t)
        if (bc * 64 + t > i && cc[t] == ci && iou_gt(mine, &cb[t * 4], p.thr)) bits |= 1ull << t;
    p.mask_ws[((int64_t)b * p.R + i) * p.nwords + bc] = bits;
}

__global__ __launch_bounds__(64) void detection_scan_kernel(const DetArgs p) {
    __shared__ int keep[DET_CAP];
    const int b = blockIdx.x, lane = threadIdx.x, nw = p.nwords;
    const int nvalid = p.nvalid[b];
    const unsigned long long* mask = p.mask_ws + (int64_t)b * p.R * nw;
    unsigned long long rem = 0ull;                           // removed bits, word `lane` (nw <= 32)
    int total = 0;
    const int nchunks = (nvalid + 63) >> 6;
    unsigned long long diag = (lane < nvalid) ? mask[(int64_t)lane * nw] : 0ull;
    for (int c = 0; c < nchunks && total < p.maxi; ++c) {
        const int i = c * 64 + lane;
        unsigned long long diag_next = 0ull;
        if (c + 1 < nchunks && i + 64 < nvalid) diag_next = mask[(int64_t)(i + 64) * nw + c + 1];
        const unsigned long long remc = readlane64(rem, c & 63);
        const int valid_n = (nvalid - c * 64) < 64 ? (nvalid - c * 64) : 64;
        unsigned long long alive = ~remc;
        if (valid_n < 64) alive &= (1ull << valid_n) - 1ull;
        unsigned long long kept = 0ull;
        while (alive && total < p.maxi) {
            const int t = __ffsll((long long)alive) - 1;
            kept |= 1ull << t;
            ++total;
            alive &= ~readlane64(diag, t);
            alive &= ~(1ull << t);
        }
        if ((kept >> lane) & 1ull) {
            int pos = total - __popcll(kept) + __popcll(kept & ((1ull << lane) - 1ull));
            keep[pos] = i;
        }
        unsigned long long k2 = kept;
        const bool wv = lane > c && lane < nchunks;
        while (k2) {                                         // rows of the kept boxes, sixteen loads in flight
            unsigned long long v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v[u] = 0ull;
                if (k2) {
                    const int t = __ffsll((long long)k2) - 1;
                    k2 &= k2 - 1ull;
                    if (wv) v[u] = mask[(int64_t)(c * 64 + t) * nw + lane];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) rem |= v[u];
        }
        diag = diag_next;
    }
    __syncthreads();
    const float* qbox = p.qbox + (int64_t)b * p.R * 4;
    const int* qcls = p.qcls + (int64_t)b * p.R;
    const float* qscore = p.qscore + (int64_t)b * p.R;
    for (int q = lane; q < p.maxi; q += 64) {
        float* o = p.det + ((int64_t)b * p.maxi + q) * 6;
        if (q < total) {
            const int k = keep[q];
            o[0] = qbox[k * 4]; o[1] = qbox[k * 4 + 1]; o[2] = qbox[k * 4 + 2]; o[3] = qbox[k * 4 + 3];
            o[4] = (float)qcls[k];
            o[5] = qscore[k];
        } else {
            o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.f;
        }
    }
}

static size_t det_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t mrcnn_detection_workspace(const mrcnn_detection_desc* d) {
    if (!d || d->B <= 0 || d->R <= 0) return 0;
    const size_t nw = ((size_t)d->R + 63) / 64, BR = (size_t)d->B * d->R;
    return det_align(BR * nw * sizeof(unsigned long long)) + det_align(BR * 4 * sizeof(float)) + det_align(BR * sizeof(int)) +
           det_align(BR * sizeof(float)) + det_align((size_t)d->B * sizeof(int)) + 256;
}

extern "C" int mrcnn_detection_fwd(const mrcnn_detection_desc* d, const float* rois, const float* probs,
                                   const float* deltas, const float* windows, float* detections,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !rois || !probs || !deltas || !windows || !detections || !workspace) return MRCNN_ERR_ARG;
    // the scan wave keeps one removed-bitmap word per lane and the prepare kernel's LDS arrays cap R at DET_CAP
    if (d->B <= 0 || d->R <= 0 || d->R > DET_CAP || d->C <= 1 || d->C > DET_MAX_CLASSES || d->max_instances <= 0 ||
        d->max_instances > DET_CAP)
        return MRCNN_ERR_ARG;
    if (workspace_bytes < mrcnn_detection_workspace(d)) return MRCNN_ERR_WORKSPACE;
    DetArgs a;
    a.rois = rois; a.probs = probs; a.deltas = deltas; a.windows = windows; a.det = detections;
    const size_t nw = ((size_t)d->R + 63) / 64, BR = (size_t)d->B * d->R;
    char* q = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.mask_ws = reinterpret_cast<unsigned long long*>(q); q += det_align(BR * nw * sizeof(unsigned long long));
    a.qbox = reinterpret_cast<float*>(q); q += det_align(BR * 4 * sizeof(float));
    a.qcls = reinterpret_cast<int*>(q); q += det_align(BR * sizeof(int));
    a.qscore = reinterpret_cast<float*>(q); q += det_align(BR * sizeof(float));
    a.nvalid = reinterpret_cast<int*>(q);
    a.B = d->B; a.R = d->R; a.C = d->C; a.maxi = d->max_instances; a.nwords = (int)nw;
    a.minconf = d->min_confidence; a.thr = d->nms_threshold;
    a.s0 = d->bbox_std_dev[0]; a.s1 = d->bbox_std_dev[1]; a.s2 = d->bbox_std_dev[2]; a.s3 = d->bbox_std_dev[3];
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(detection_prepare_kernel, dim3(d->B), dim3(DET_THREADS), 0, s, a);
    hipLaunchKernelGGL(detection_mask_kernel, dim3((unsigned)nw, (unsigned)nw, d->B), dim3(64), 0, s, a);
    hipLaunchKernelGGL(detection_scan_kernel, dim3(d->B), dim3(64), 0, s, a);
    return mrcnn_launch_status();
}
