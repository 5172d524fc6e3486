// ProposalLayer (mrcnn/model.py:329-406): top-k by foreground score, box decode, clip, greedy NMS,
// zero padding -- per image, index-exact.
//
//  K1 select_sort_decode (one 1024-thread workgroup per image)
//       4-pass MSB-first radix select of the k-th largest score over the A anchors (LDS histograms),
//       compaction of the k winners (ties at the threshold taken in ascending anchor index, which is
//       tf.nn.top_k's order), bitonic sort of 64-bit (score desc, index asc) keys -- 8 keys per thread in registers,
//       wave-level exchanges by shuffle, LDS only for the strides that cross waves -- then
//       apply_box_deltas_graph + clip_boxes_graph on the sorted winners.
//  K2 nms_mask: 64x64 tiles of the upper-triangular suppression bit matrix (IoU > threshold).
//  K3 nms_scan (one wave per image): chunked greedy scan over the bit matrix, gather + zero pad.
//
// IoU follows TF's non_max_suppression kernel [3P]: corners canonicalised with min/max, 0 when an area
// is <= 0, inter / (area_i + area_j - inter), suppress when IoU > threshold.  Compiled with
// -ffp-contract=off so the float32 operation order is the reference's (no fused multiply-add).
#include "common.h"

#define SORT_CAP 8192           // bitonic capacity (pre_nms_limit <= SORT_CAP)
#ifdef MRCNN_PROP_STAMPS        // debug build (build(extra_flags=["-DMRCNN_PROP_STAMPS"])): section times of select_sort_decode, image 0
#define PROP_STAMP(i) do { if (tid == 0 && blockIdx.x == 0) st_[i] = wall_clock64(); } while (0)
#else
#define PROP_STAMP(i) do { } while (0)
#endif
#define K1_THREADS 1024

__device__ __forceinline__ unsigned f2key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending unsigned == ascending float
}

struct PropArgs {
    const float* probs; const float* deltas; const float* anchors;
    float* rois; int32_t* top_idx; int32_t* keep_idx; int32_t* num_keep;
    float* boxes_ws;               // [B, K, 4]
    unsigned long long* mask_ws;   // [B, K, nwords]
    unsigned* pre_ws;              // [B, PRE_WORDS]: three histograms, per-workgroup tie counts, counter, candidate count
    unsigned long long* cand_ws;   // [B, SORT_CAP] candidate keys of the multi-workgroup pre-selection
    int pre_groups;                // workgroups per image of the pre-selection (0 = off)
    int B, A, K, proposal_count, nwords;
    float thr, s0, s1, s2, s3;
};

// ---- multi-workgroup selection ----------------------------------------------------------------------------
// One workgroup walking all A scores four times is what the selection costs on large images (A = 261 888 at 1024^2:
// 0.9 ms).  For A >= 32 768 the exact selection runs over many workgroups in five small launches:
//   topk_hist_kernel x3   histograms of key bits [31:20], [19:8], [7:0] (LDS per workgroup, then global atomics), each
//                         restricted to the threshold prefix found in the previous ones -> the K-th largest key T and
//                         how many anchors equal to T belong to the top K
//   topk_ties_kernel      anchors == T per workgroup; workgroups own CONTIGUOUS anchor ranges, so tie order = index order
//   topk_collect_kernel   keys > T (any order) and the first need_eq ties in anchor-index order (tf.nn.top_k's tie
//                         rule), exactly K keys -> the single-workgroup kernel only sorts and decodes them.
#define PRE_THREADS 256
#define PRE_BINS 4096
#define PRE_MAXG 64
#define PRE_TIE (2 * PRE_BINS + 256)               // word offsets inside pre_ws
#define PRE_COUNTER (PRE_TIE + PRE_MAXG)
#define PRE_NCAND (PRE_COUNTER + 1)
#define PRE_ERR (PRE_NCAND + 1)                     // stores refused by a bounds guard (0 on a healthy run; host-readable)
#define PRE_WORDS (PRE_ERR + 1)

// bin d of hist[0..nbins) with count(bins > d) < need <= count(bins >= d); *left = need - count(bins > d)
__device__ int pre_find_bin(const unsigned* __restrict__ hist, int nbins, unsigned need, unsigned* left, unsigned* s_part, int* s_res) {
    const int tid = threadIdx.x, per = nbins / PRE_THREADS;       // nbins = 4096 or 256
    unsigned sum = 0;
    for (int i = 0; i < per; ++i) sum += hist[tid * per + i];
    __syncthreads();
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned cum = 0;
        int t = PRE_THREADS - 1;
        for (; t > 0; --t) {
            if (cum + s_part[t] >= need) break;
            cum += s_part[t];
        }
        int d = t * per + per - 1;
        for (; d > t * per; --d) {
            if (cum + hist[d] >= need) break;
            cum += hist[d];
        }
        s_res[0] = d;
        s_res[1] = (int)(need - cum);
    }
    __syncthreads();
    *left = (unsigned)s_res[1];
    return s_res[0];
}

__global__ void topk_zero_kernel(unsigned* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

struct PreThreshold { unsigned prefix, shift, need; };   // keys with (key >> shift) == prefix are still undecided

// thresholds of the first `passes` histograms (workgroup-wide call)
__device__ PreThreshold pre_threshold(const unsigned* pre, int K, int passes, unsigned* s_part, int* s_res) {
    PreThreshold t;
    t.prefix = 0; t.shift = 32; t.need = (unsigned)K;
    if (passes >= 1) { const unsigned b1 = (unsigned)pre_find_bin(pre, PRE_BINS, t.need, &t.need, s_part, s_res); t.prefix = b1; t.shift = 20; }
    if (passes >= 2) { const unsigned b2 = (unsigned)pre_find_bin(pre + PRE_BINS, PRE_BINS, t.need, &t.need, s_part, s_res); t.prefix = (t.prefix << 12) | b2; t.shift = 8; }
    if (passes >= 3) { const unsigned b3 = (unsigned)pre_find_bin(pre + 2 * PRE_BINS, 256, t.need, &t.need, s_part, s_res); t.prefix = (t.prefix << 8) | b3; t.shift = 0; }
    return t;
}

__global__ __launch_bounds__(PRE_THREADS) void topk_hist_kernel(const PropArgs p, const int pass) {
    __shared__ unsigned hist[PRE_BINS];
    __shared__ unsigned s_part[PRE_THREADS];
    __shared__ int s_res[2];
    const int b = blockIdx.y, tid = threadIdx.x;
    unsigned* pre = p.pre_ws + (int64_t)b * PRE_WORDS;
    const float* sc = p.probs + (int64_t)b * p.A * 2 + 1;
    const PreThreshold t = pre_threshold(pre, p.K, pass, s_part, s_res);
    const int nbins = pass == 2 ? 256 : PRE_BINS;
    const int bshift = pass == 0 ? 20 : (pass == 1 ? 8 : 0);
    for (int i = tid; i < nbins; i += PRE_THREADS) hist[i] = 0;
    __syncthreads();
    for (int a = blockIdx.x * PRE_THREADS + tid; a < p.A; a += gridDim.x * PRE_THREADS) {
        const unsigned u = f2key(sc[(int64_t)a * 2]);
        if (pass == 0 || (u >> t.shift) == t.prefix) atomicAdd(&hist[(u >> bshift) & (nbins - 1)], 1u);
    }
    __syncthreads();
    unsigned* out = pre + pass * PRE_BINS;
    for (int i = tid; i < nbins; i += PRE_THREADS)
        if (hist[i]) atomicAdd(&out[i], hist[i]);
}

// workgroup g owns anchors [g*chunk, (g+1)*chunk)
__global__ __launch_bounds__(PRE_THREADS) void topk_ties_kernel(const PropArgs p, const int chunk) {
    __shared__ unsigned s_part[PRE_THREADS];
    __shared__ int s_res[2];
    __shared__ unsigned s_cnt;
    const int b = blockIdx.y, tid = threadIdx.x, g = blockIdx.x;
    unsigned* pre = p.pre_ws + (int64_t)b * PRE_WORDS;
    const float* sc = p.probs + (int64_t)b * p.A * 2 + 1;
    const PreThreshold t = pre_threshold(pre, p.K, 3, s_part, s_res);
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const int a1 = min(p.A, (g + 1) * chunk);
    unsigned mine = 0;
    for (int a = g * chunk + tid; a < a1; a += PRE_THREADS) mine += f2key(sc[(int64_t)a * 2]) == t.prefix;
    atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0) pre[PRE_TIE + g] = s_cnt;
}

__global__ __launch_bounds__(PRE_THREADS) void topk_collect_kernel(const PropArgs p, const int chunk) {
    __shared__ unsigned s_part[PRE_THREADS];
    __shared__ int s_res[2];
    __shared__ unsigned s_wsum[PRE_THREADS / 64];
    const int b = blockIdx.y, tid = threadIdx.x, g = blockIdx.x;
    unsigned* pre = p.pre_ws + (int64_t)b * PRE_WORDS;
    const float* sc = p.probs + (int64_t)b * p.A * 2 + 1;
    const PreThreshold t = pre_threshold(pre, p.K, 3, s_part, s_res);
    const unsigned T = t.prefix, need_eq = t.need;
    unsigned base = 0;                                   // ties in the workgroups before this one
    for (int q = 0; q < g; ++q) base += pre[PRE_TIE + q];
    if (g == 0 && tid == 0) pre[PRE_NCAND] = (unsigned)p.K;
    unsigned long long* cand = p.cand_ws + (int64_t)b * SORT_CAP;
    const int a1 = min(p.A, (g + 1) * chunk);
    for (int a0 = g * chunk; a0 < a1; a0 += PRE_THREADS) {
        const int a = a0 + tid;
        unsigned u = 0;
        bool gt = false, eq = false;
        if (a < a1) { u = f2key(sc[(int64_t)a * 2]); gt = u > T; eq = u == T; }
        // rank of this tie among all ties in anchor order
        const unsigned long long bal = __ballot(eq);
        const unsigned wrank = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if ((tid & 63) == 0) s_wsum[tid >> 6] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = base, tot = 0;
        for (int w = 0; w < PRE_THREADS / 64; ++w) { if (w < (tid >> 6)) before += s_wsum[w]; tot += s_wsum[w]; }
        if (gt || (eq && before + wrank < need_eq)) {
            // the slot comes from a counter in global memory: never trust it as an index.  A counter that did not start
            // at zero (or inconsistent histograms) would otherwise walk off the candidate array; the refused store is
            // counted, and select_sort_decode_kernel sees PRE_COUNTER != K and runs its own selection instead.
            const unsigned slot = atomicAdd(&pre[PRE_COUNTER], 1u);
            if (slot < (unsigned)SORT_CAP) cand[slot] = ((unsigned long long)(~u) << 32) | (unsigned)a;
            else atomicAdd(&pre[PRE_ERR], 1u);
        }
        base += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(K1_THREADS) void select_sort_decode_kernel(const PropArgs p) {
    __shared__ unsigned long long keys[SORT_CAP];
    __shared__ unsigned long long xkeys[SORT_CAP];            // second exchange buffer of the sort's cross-wave stages
    __shared__ unsigned hist[256], s_suf[256];
    __shared__ unsigned s_prefix, s_need, s_count, s_wsum[16], s_base;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sc = p.probs + (int64_t)b * p.A * 2 + 1;     // foreground probability, stride 2
    const int A = p.A, K = p.K;
#ifdef MRCNN_PROP_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PROP_STAMP(0);

    // the pre-selection must have delivered exactly the K keys it announced; anything else (counter not reset, refused
    // stores) is discarded and this workgroup selects for itself -- slower, never wrong and never out of bounds
    const unsigned* pre = p.pre_ws + (int64_t)b * PRE_WORDS;
    const unsigned ncand = p.pre_groups ? pre[PRE_NCAND] : ~0u;
    const bool preselected = p.pre_groups && ncand == (unsigned)K && pre[PRE_COUNTER] == (unsigned)K;   // workgroup-uniform

    // ---- radix select: largest K keys ------------------------------------------------------------
    if (tid == 0) { s_prefix = 0; s_need = (unsigned)K; }
    __syncthreads();
    unsigned cnt_eq = 0;
    // round 3: the scores of a 256 x 256 image (A = 16 368) are read ONCE -- a thread's 16 keys stay in registers for the four
    // passes and the compaction (five strided walks over the scores before; larger A without a pre-selection keeps those)
    constexpr int KC = 16;
    const bool cached = !preselected && A <= KC * K1_THREADS;
    unsigned uk[KC];
    if (cached) {
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            const int a = i * K1_THREADS + tid;
            uk[i] = a < A ? f2key(sc[(int64_t)a * 2]) : 0u;      // key 0 is below every real key (f2key sets or flips the top bit)
        }
    }
    for (int pass = 0; pass < 4 && !preselected; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        if (cached) {
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                const unsigned u = uk[i];
                if (i * K1_THREADS + tid < A && (u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
            }
        } else {
            for (int a = tid; a < A; a += K1_THREADS) {
                unsigned u = f2key(sc[(int64_t)a * 2]);
                if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        // the digit d with count(digits > d) < need <= count(digits >= d): suffix sums of the 256 bins by the first four waves
        // (shuffle scan inside a wave, wave totals through LDS) -- one thread walking the bins took ~8 us per pass
        {
            const unsigned need = s_need;
            unsigned suf = 0;
            if (tid < 256) {
                suf = hist[tid];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned v = __shfl_down(suf, o, 64);
                    if ((tid & 63) + o < 64) suf += v;
                }
                if ((tid & 63) == 0) s_wsum[tid >> 6] = suf;           // total of this wave's 64 bins
            }
            __syncthreads();
            if (tid < 256) {
                for (int w = (tid >> 6) + 1; w < 4; ++w) suf += s_wsum[w];
                s_suf[tid] = suf;                                       // count(digits >= tid)
            }
            __syncthreads();
            if (tid < 256) {
                const unsigned above = tid < 255 ? s_suf[tid + 1] : 0u;    // count(digits > tid)
                // suffix sums fall with the digit: exactly one digit has above < need <= suf (digit 0 takes the rest, as the walk did)
                if ((suf >= need || tid == 0) && above < need) {
                    s_prefix = prefix | ((unsigned)tid << shift);
                    s_need = need - above;
                    s_count = hist[tid];
                }
            }
        }
        __syncthreads();
        cnt_eq = s_count;
    }
    PROP_STAMP(1);
    const unsigned T = s_prefix;       // k-th largest key
    const unsigned need_eq = s_need;   // how many keys == T belong to the top K (>= 1)

    // ---- compaction into keys[] --------------------------------------------------------------------
    __syncthreads();                   // everyone has read s_count/s_prefix/s_need
    for (int i = tid; i < SORT_CAP; i += K1_THREADS) keys[i] = ~0ull;
    if (tid == 0) { s_count = 0; s_base = 0; }
    __syncthreads();
    const bool ordered = cnt_eq > need_eq;      // more ties than slots: lowest anchor index first
    if (preselected) {                          // >= K candidates, every one of the top K among them: sort them all
        const unsigned long long* cand = p.cand_ws + (int64_t)b * SORT_CAP;
        for (int i = tid; i < (int)ncand; i += K1_THREADS) keys[i] = cand[i];
    }
    for (int a0 = 0, it = 0; a0 < A && !preselected; a0 += K1_THREADS, ++it) {
        const int a = a0 + tid;
        unsigned u = 0;
        bool gt = false, eq = false;
        if (a < A) {
            if (cached) {
                u = uk[0];
#pragma unroll
                for (int i = 1; i < KC; ++i) u = it == i ? uk[i] : u;      // (a select chain: a run-time index would put uk[] in scratch)
            } else {
                u = f2key(sc[(int64_t)a * 2]);
            }
            gt = u > T;
            eq = u == T;
        }
        if (gt || (eq && !ordered)) {
            unsigned slot = atomicAdd(&s_count, 1u);          // < K by construction of the radix select; guarded all the same
            if (slot < (unsigned)SORT_CAP) keys[slot] = ((unsigned long long)(~u) << 32) | (unsigned)a;
        }
        if (ordered) {
            // block-wide exclusive rank of `eq` in index order
            unsigned long long bal = __ballot(eq);
            unsigned wrank = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if ((tid & 63) == 0) s_wsum[tid >> 6] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = s_base;
            for (int w = 0; w < (tid >> 6); ++w) before += s_wsum[w];
            if (eq && before + wrank < need_eq) {
                unsigned slot = atomicAdd(&s_count, 1u);
                if (slot < (unsigned)SORT_CAP) keys[slot] = ((unsigned long long)(~u) << 32) | (unsigned)a;
            }
            __syncthreads();
            if (tid == 0) {
                unsigned tot = 0;
                for (int w = 0; w < K1_THREADS / 64; ++w) tot += s_wsum[w];
                s_base += tot;
            }
            __syncthreads();
        }
    }
    __syncthreads();

    PROP_STAMP(2);
    // ---- bitonic sort (ascending 64-bit keys == score descending, index ascending) ----------------
    // round 3: a thread holds 8 consecutive keys in registers (e = 8 tid + j).  Strides 1 / 2 / 4 are exchanges inside the thread,
    // strides 8 .. 256 lane exchanges inside the wave (__shfl_xor), only strides >= 512 cross waves and go through LDS (two
    // buffers in turn: one barrier per such stage): 10 barrier stages for 8 192 keys instead of 91.  The network and its
    // comparisons are the ones of the LDS form (same keys, same order: the result is a sort either way).
    int n = 1;
    const int nsort = preselected && (int)ncand > K ? (int)ncand : K;
    while (n < nsort) n <<= 1;
    if (n < 8) n = 8;
    unsigned long long k8[8];
    const bool mine = tid * 8 < n;                             // threads past n hold padding only
#pragma unroll
    for (int j = 0; j < 8; ++j) k8[j] = mine ? keys[tid * 8 + j] : ~0ull;
    __syncthreads();                                           // keys[] is free: it becomes the first exchange buffer
    auto cs = [](unsigned long long& lo, unsigned long long& hi, const bool up) {
        const bool sw = (lo > hi) == up;
        const unsigned long long x = sw ? hi : lo, y = sw ? lo : hi;
        lo = x; hi = y;
    };
    auto intra = [&](const int stride, const bool up0, const int size) {       // stride 1, 2 or 4; `up` per element when size < 8
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if ((j & stride) == 0) {
                const bool up = size >= 8 ? up0 : ((j & size) == 0);
                cs(k8[j], k8[j | stride], up);
            }
        }
    };
    const int lane = tid & 63;
    int xbuf = 0;
    for (int size = 2; size <= n; size <<= 1) {
        const bool up0 = ((tid * 8) & size) == 0;              // direction of this thread's elements for size >= 8
        for (int stride = size >> 1; stride >= 512; stride >>= 1) {      // partner thread in another wave: through LDS
            unsigned long long* buf = xbuf ? xkeys : keys;
            xbuf ^= 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) buf[tid * 8 + j] = k8[j];
            __syncthreads();
            const int pt = tid ^ (stride >> 3);
            const bool is_lo = (tid & (stride >> 3)) == 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned long long o = buf[pt * 8 + j];
                const bool take_min = is_lo == up0;
                k8[j] = take_min ? (o < k8[j] ? o : k8[j]) : (o > k8[j] ? o : k8[j]);
            }
        }
        for (int stride = (size >> 1) < 256 ? (size >> 1) : 256; stride >= 8; stride >>= 1) {   // partner lane in this wave
            const int lm = stride >> 3;
            const bool is_lo = (lane & lm) == 0;
            const bool take_min = is_lo == up0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned long long o = __shfl_xor(k8[j], lm, 64);
                k8[j] = take_min ? (o < k8[j] ? o : k8[j]) : (o > k8[j] ? o : k8[j]);
            }
        }
        if (size >= 8) { intra(4, up0, size); intra(2, up0, size); intra(1, up0, size); }
        else if (size == 4) { intra(2, up0, 4); intra(1, up0, 4); }
        else intra(1, up0, 2);
    }

    PROP_STAMP(3);
    // ---- decode + clip (apply_box_deltas_graph, clip_boxes_graph with window [0,0,1,1]) ------------
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid * 8 + j;
        if (i >= K) continue;
        const unsigned a = (unsigned)(k8[j] & 0xFFFFFFFFull);
        if (p.top_idx) p.top_idx[(int64_t)b * K + i] = (int)a;
        const float* an = p.anchors + (int64_t)a * 4;
        const float* dl = p.deltas + ((int64_t)b * A + a) * 4;
        float d0 = dl[0] * p.s0, d1 = dl[1] * p.s1, d2 = dl[2] * p.s2, d3 = dl[3] * p.s3;
        decode_clip_box(an, d0, d1, d2, d3, 0.f, 0.f, 1.f, 1.f, p.boxes_ws + ((int64_t)b * K + i) * 4);
    }
    PROP_STAMP(4);
#ifdef MRCNN_PROP_STAMPS
    if (tid == 0 && blockIdx.x == 0)
        printf("select_sort_decode (10 ns units): select %llu  compaction %llu  sort %llu  decode %llu\n", st_[1] - st_[0], st_[2] - st_[1],
               st_[3] - st_[2], st_[4] - st_[3]);
#endif
}

// grid (nwords, nwords, B); workgroup = 64 threads: row i = 64*blockIdx.y + lane against 64 columns
__global__ __launch_bounds__(64) void nms_mask_kernel(const PropArgs p) {
    const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x, lane = threadIdx.x;
    const int K = p.K;
    const int i = rb * 64 + lane;
    unsigned long long* mrow = p.mask_ws + ((int64_t)b * K + i) * p.nwords + cb;
    if (cb < rb) return;                 // strictly lower blocks are never read (nms_scan touches words >= its chunk only):
                                         // not written either -- 35.7 -> ~18 MB of stores per 6000-box image pair (profiles/r02_hbm_kernels.md)
    __shared__ float cbox[64 * 4];
    const float* boxes = p.boxes_ws + (int64_t)b * K * 4;
    const int j0 = cb * 64;
    if (j0 + lane < K) {
        const float* s = boxes + (int64_t)(j0 + lane) * 4;
        cbox[lane * 4 + 0] = s[0]; cbox[lane * 4 + 1] = s[1]; cbox[lane * 4 + 2] = s[2]; cbox[lane * 4 + 3] = s[3];
    }
    __syncthreads();
    if (i >= K) return;
    float me[4] = {boxes[(int64_t)i * 4], boxes[(int64_t)i * 4 + 1], boxes[(int64_t)i * 4 + 2], boxes[(int64_t)i * 4 + 3]};
    unsigned long long bits = 0ull;
    const int jn = (K - j0) < 64 ? (K - j0) : 64;
    for (int j = 0; j < jn; ++j) {
        if (j0 + j > i && iou_gt(me, &cbox[j * 4], p.thr)) bits |= 1ull << j;
    }
    *mrow = bits;
}

// one wave per image
__global__ __launch_bounds__(64) void nms_scan_kernel(const PropArgs p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int K = p.K, nw = p.nwords, maxk = p.proposal_count;
    const unsigned long long* mask = p.mask_ws + (int64_t)b * K * nw;
    const float* boxes = p.boxes_ws + (int64_t)b * K * 4;
    __shared__ int keep[SORT_CAP];
    unsigned long long rem0 = 0ull, rem1 = 0ull;     // removed bits: words lane and lane+64
    int total = 0;
    // The greedy decision inside a chunk is wave-uniform: it runs on the scalar unit over the ALIVE candidates only
    // (find-first-set, two v_readlane per kept box).  What is left on the critical path is memory latency: the
    // diagonal word of the next chunk is prefetched, and the rows of the boxes a chunk keeps are fetched sixteen at a
    // time so that their independent loads overlap.
    unsigned long long diag = (lane < K) ? mask[(int64_t)lane * nw] : 0ull;
    for (int c = 0; c < nw && total < maxk; ++c) {
        const int i = c * 64 + lane;
        unsigned long long diag_next = 0ull;
        if (c + 1 < nw && i + 64 < K) diag_next = mask[(int64_t)(i + 64) * nw + c + 1];
        const unsigned long long remc = readlane64(c < 64 ? rem0 : rem1, c & 63);
        const int valid_n = (K - c * 64) < 64 ? (K - c * 64) : 64;
        unsigned long long alive = ~remc;
        if (valid_n < 64) alive &= (1ull << valid_n) - 1ull;
        unsigned long long kept = 0ull;
        // No alive candidate of this chunk overlaps another alive one (the usual case away from object clusters): the
        // greedy scan would keep them all, one find-first-set round per box -- take them at once.
        const bool clean = __ballot(((alive >> lane) & 1ull) && (diag & alive) != 0ull) == 0ull;
        if (clean && __popcll(alive) <= maxk - total) {
            kept = alive;
            total += __popcll(alive);
            alive = 0ull;
        }
        while (alive && total < maxk) {
            const int t = __ffsll((long long)alive) - 1;
            kept |= 1ull << t;
            ++total;
            alive &= ~readlane64(diag, t);
            alive &= ~(1ull << t);
        }
        // record kept boxes of this chunk
        if ((kept >> lane) & 1ull) {
            int pos = total - __popcll(kept) + __popcll(kept & ((1ull << lane) - 1ull));
            keep[pos] = i;
        }
        // OR the rows of the kept boxes into the removed bitmap (only words > c matter)
        unsigned long long k2 = kept;
        const bool w0 = lane > c && lane < nw, w1 = lane + 64 > c && lane + 64 < nw;
        while (k2) {
            unsigned long long v0[16], v1[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v0[u] = 0ull; v1[u] = 0ull;
                if (k2) {
                    const int t = __ffsll((long long)k2) - 1;
                    k2 &= k2 - 1ull;
                    const unsigned long long* row = mask + (int64_t)(c * 64 + t) * nw;
                    if (w0) v0[u] = row[lane];
                    if (w1) v1[u] = row[lane + 64];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { rem0 |= v0[u]; rem1 |= v1[u]; }
        }
        diag = diag_next;
    }
    __syncthreads();
    if (p.num_keep && lane == 0) p.num_keep[b] = total;
    for (int q = lane; q < maxk; q += 64) {
        float* o = p.rois + ((int64_t)b * maxk + q) * 4;
        if (q < total) {
            const float* s = boxes + (int64_t)keep[q] * 4;
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
            if (p.keep_idx) p.keep_idx[(int64_t)b * maxk + q] = keep[q];
        } else {
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
            if (p.keep_idx) p.keep_idx[(int64_t)b * maxk + q] = -1;
        }
    }
}

static inline int prop_k(const mrcnn_proposal_desc* d) { return d->pre_nms_limit < d->A ? d->pre_nms_limit : d->A; }

extern "C" size_t mrcnn_proposal_workspace(const mrcnn_proposal_desc* d) {
    if (!d || d->B <= 0 || d->A <= 0) return 0;
    size_t K = (size_t)prop_k(d), nw = (K + 63) / 64;
    return (size_t)d->B * (K * 4 * sizeof(float) + K * nw * sizeof(unsigned long long) + SORT_CAP * sizeof(unsigned long long) +
                           PRE_WORDS * sizeof(unsigned)) + 512;
}

// Where the selection's health words live, for a host that has synchronised anyway (tests, debugging): byte offset from
// `workspace` of image 0's {collected, announced, refused} uint32 triple, *stride_bytes between images.  Healthy run:
// collected == announced == K (or both 0 when A < 32 768: no multi-workgroup selection), refused == 0.
extern "C" size_t mrcnn_proposal_status_offset(const mrcnn_proposal_desc* d, const void* workspace, size_t* stride_bytes) {
    if (!d || !workspace) return 0;
    const size_t K = (size_t)prop_k(d), nw = (K + 63) / 64;
    const uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    const uintptr_t pre = base + (size_t)d->B * (K * 4 * sizeof(float) + K * nw * sizeof(unsigned long long) +
                                                 SORT_CAP * sizeof(unsigned long long));
    if (stride_bytes) *stride_bytes = PRE_WORDS * sizeof(unsigned);
    return (size_t)(pre - reinterpret_cast<uintptr_t>(workspace)) + PRE_COUNTER * sizeof(unsigned);
}

extern "C" int mrcnn_proposal_fwd(const mrcnn_proposal_desc* d, const float* rpn_probs, const float* rpn_bbox,
                                  const float* anchors, float* rois, int32_t* top_idx, int32_t* keep_idx,
                                  int32_t* num_keep, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !rpn_probs || !rpn_bbox || !anchors || !rois || !workspace) return MRCNN_ERR_ARG;
    if (d->B <= 0 || d->A <= 0 || d->pre_nms_limit <= 0 || d->proposal_count <= 0) return MRCNN_ERR_ARG;
    const int K = prop_k(d);
    if (K > SORT_CAP || d->proposal_count > SORT_CAP) return MRCNN_ERR_ARG;
    if (workspace_bytes < mrcnn_proposal_workspace(d)) return MRCNN_ERR_WORKSPACE;
    PropArgs a;
    a.probs = rpn_probs; a.deltas = rpn_bbox; a.anchors = anchors; a.rois = rois; a.top_idx = top_idx;
    a.keep_idx = keep_idx; a.num_keep = num_keep;
    a.B = d->B; a.A = d->A; a.K = K; a.proposal_count = d->proposal_count; a.nwords = (K + 63) / 64;
    a.thr = d->nms_threshold; a.s0 = d->std_dev[0]; a.s1 = d->std_dev[1]; a.s2 = d->std_dev[2]; a.s3 = d->std_dev[3];
    uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    a.boxes_ws = reinterpret_cast<float*>(base);
    a.mask_ws = reinterpret_cast<unsigned long long*>(base + (size_t)d->B * K * 4 * sizeof(float));
    a.cand_ws = a.mask_ws + (size_t)d->B * K * a.nwords;
    a.pre_ws = reinterpret_cast<unsigned*>(a.cand_ws + (size_t)d->B * SORT_CAP);
    hipStream_t s = (hipStream_t)stream;
    // pre-selection over many workgroups when one workgroup would have to walk > 32 anchors per thread four times
    static const long long pre_min = getenv("MRCNN_TOPK_PRE_MIN") ? atoll(getenv("MRCNN_TOPK_PRE_MIN")) : 32 * K1_THREADS;   // A/B
    a.pre_groups = d->A >= pre_min && d->A > K ? (int)((d->A + 4095) / 4096 < PRE_MAXG ? (d->A + 4095) / 4096 : PRE_MAXG) : 0;
    if (a.pre_groups) {
        const int chunk = (d->A + a.pre_groups - 1) / a.pre_groups;
        const dim3 grid(a.pre_groups, d->B);
        const int nz = d->B * PRE_WORDS;            // a kernel, not a memset node: the call may be inside a graph capture
        // mrcnn_tuning_set("proposal_skip_zero", 1) (tests only; explicit process state, not an environment variable a
        // production run could inherit) leaves the previous call's counters in place: the fault-injection twin of "the reset
        // did not happen", which the guards above must survive with the exact result
        if (!g_mrcnn_proposal_skip_zero)
            hipLaunchKernelGGL(topk_zero_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, s, a.pre_ws, nz);
        for (int pass = 0; pass < 3; ++pass) hipLaunchKernelGGL(topk_hist_kernel, grid, dim3(PRE_THREADS), 0, s, a, pass);
        hipLaunchKernelGGL(topk_ties_kernel, grid, dim3(PRE_THREADS), 0, s, a, chunk);
        hipLaunchKernelGGL(topk_collect_kernel, grid, dim3(PRE_THREADS), 0, s, a, chunk);
    }
    hipLaunchKernelGGL(select_sort_decode_kernel, dim3(d->B), dim3(K1_THREADS), 0, s, a);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(a.nwords, a.nwords, d->B), dim3(64), 0, s, a);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(d->B), dim3(64), 0, s, a);
    return mrcnn_launch_status();
}
