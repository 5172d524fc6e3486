// The five Mask R-CNN losses and their gradients w.r.t. the network outputs
// (mrcnn/model.py:1098-1270; K.sparse_categorical_crossentropy / tf.nn.sparse_softmax_cross_entropy /
// K.binary_crossentropy semantics are [3P] Keras 2.2.4 + TF 1.13).
//
//   phase 1 (reduce): per-image RPN pass (ordered rank of positive anchors == batch_pack_graph order)
//                     and per-ROI head pass; block sums -> float atomics on 16 scalars.
//   phase 2 (grads):  elementwise passes that read the scalars (means) and write dense gradients.
#include "common.h"

enum { S_RC_SUM = 0, S_RC_CNT, S_RB_SUM, S_RB_CNT, S_C_SUM, S_C_PA, S_B_SUM, S_POS, S_M_SUM, S_D_Y, S_D_P, S_NUM = 16 };

struct LossArgs {
    const int32_t* rpn_match; const float* rpn_bbox_t; const float* rpn_logits; const float* rpn_bbox;
    const int32_t* tcls; const float* tbbox; const float* tmask; const int32_t* active;
    const float* cls_logits; const float* mbbox; const float* mmask;
    float* losses; float* d_rpn_logits; float* d_rpn_bbox; float* d_cls_logits; float* d_mbbox; float* d_mmask;
    float* scal; int32_t* rank;
    int B, A, T, C, mh, mw, max_rpn_pos, dice;
    float w0, w1, w2, w3, w4;
};

__device__ __forceinline__ float block_sum(float v, float* sbuf) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sbuf[w];
    return t;
}

__device__ __forceinline__ float smooth_l1(float t, float p) {
    float diff = fabsf(t - p);
    return diff < 1.f ? 0.5f * diff * diff : diff - 0.5f;
}
__device__ __forceinline__ float smooth_l1_grad(float t, float p) {   // d/dp
    float e = p - t, diff = fabsf(e);
    float s = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
    return diff < 1.f ? diff * s : s;
}

// ---- phase 1a: RPN ----------------------------------------------------------------------------
// Round 3: the match codes of a 1024-anchor slice used to be read, ranked (two barriers + a 16-term sum per thread) and consumed
// slice by slice -- 48 barriers and 16 dependent load round trips for A = 16 368 (139 us inside the step).  Now a thread reads
// its (up to RPN_MAXIT) match codes at once, the per-wave positive counts of every slice meet in LDS once, and the ranks follow
// from that table: two barriers in all.  Per thread the terms are added in the same slice order as before: same sums, bit for bit.
#define RPN_MAXIT 32            // slices of 1024 anchors held in registers (A <= 32 768); larger A walks in groups of RPN_MAXIT slices
__global__ __launch_bounds__(1024) void rpn_loss_reduce_kernel(const LossArgs p) {
    __shared__ float sbuf[16];
    __shared__ unsigned s_cnt[RPN_MAXIT][16];     // positives of (slice, wave)
    __shared__ unsigned s_pre[RPN_MAXIT][16];     // positives before (slice, wave) in anchor order, within this group of slices
    __shared__ unsigned s_base, s_group;
    const int b = blockIdx.x, tid = threadIdx.x, A = p.A;
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_base = 0;
    float csum = 0.f, ccnt = 0.f, bsum = 0.f, bcnt = 0.f;
    for (int g0 = 0; g0 < A; g0 += RPN_MAXIT * 1024) {
        int m[RPN_MAXIT];
        unsigned wrank[RPN_MAXIT];
#pragma unroll
        for (int i = 0; i < RPN_MAXIT; ++i) {
            const int a = g0 + i * 1024 + tid;
            m[i] = a < A ? p.rpn_match[(int64_t)b * A + a] : 0;
        }
#pragma unroll
        for (int i = 0; i < RPN_MAXIT; ++i) {
            const unsigned long long bal = __ballot(m[i] == 1);
            wrank[i] = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_cnt[i][wave] = (unsigned)__popcll(bal);
        }
        __syncthreads();
        if (tid < RPN_MAXIT * 16) {                              // exclusive prefix over the (slice, wave) table in anchor order
            // 512 entries at most: a wave-level scan per 64 entries, then the eight wave totals
            unsigned v = s_cnt[tid >> 4][tid & 15], inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(inc, o, 64);
                if (lane >= o) inc += u;
            }
            s_pre[tid >> 4][tid & 15] = inc - v;                 // exclusive inside this wave's 64 entries
            if (lane == 63) sbuf[wave] = __uint_as_float(inc);   // (sbuf borrowed as eight unsigned totals)
        }
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < RPN_MAXIT * 16 / 64; ++w) tot += __float_as_uint(sbuf[w]);
            s_group = tot;
        }
        const unsigned base = s_base;
#pragma unroll
        for (int i = 0; i < RPN_MAXIT; ++i) {
            const int a = g0 + i * 1024 + tid;
            if (a >= A) continue;
            if (m[i] != 0) {
                const float l0 = p.rpn_logits[((int64_t)b * A + a) * 2], l1 = p.rpn_logits[((int64_t)b * A + a) * 2 + 1];
                const float mx = fmaxf(l0, l1);
                const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
                csum += lse - (m[i] == 1 ? l1 : l0);
                ccnt += 1.f;
            }
            const bool pos = (m[i] == 1);
            const int e = i * 16 + wave;                         // this thread's table entry
            unsigned before = base + s_pre[i][wave];
            for (int w = 0; w < (e >> 6); ++w) before += __float_as_uint(sbuf[w]);     // the 64-entry groups before this entry's
            const int rank = (int)(before + wrank[i]);
            p.rank[(int64_t)b * A + a] = pos ? rank : -1;
            if (pos && rank < p.max_rpn_pos) {
                const float* t = p.rpn_bbox_t + ((int64_t)b * p.max_rpn_pos + rank) * 4;
                const float* q = p.rpn_bbox + ((int64_t)b * A + a) * 4;
                bsum += smooth_l1(t[0], q[0]) + smooth_l1(t[1], q[1]) + smooth_l1(t[2], q[2]) + smooth_l1(t[3], q[3]);
                bcnt += 1.f;
            }
        }
        __syncthreads();                                         // everyone has read s_base / sbuf / s_pre of this group
        if (tid == 0) s_base = base + s_group;
        __syncthreads();
    }
    csum = block_sum(csum, sbuf); ccnt = block_sum(ccnt, sbuf);
    bsum = block_sum(bsum, sbuf); bcnt = block_sum(bcnt, sbuf);
    if (tid == 0) {
        atomicAdd(&p.scal[S_RC_SUM], csum); atomicAdd(&p.scal[S_RC_CNT], ccnt);
        atomicAdd(&p.scal[S_RB_SUM], bsum); atomicAdd(&p.scal[S_RB_CNT], bcnt);
    }
}

// ---- phase 1b: heads; one workgroup per ROI row --------------------------------------------------
__global__ __launch_bounds__(256) void heads_loss_reduce_kernel(const LossArgs p) {
    __shared__ float sbuf[4];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, C = p.C;
    const int cls = p.tcls[row];
    if (tid == 0) {
        const float* l = p.cls_logits + row * C;
        float mx = l[0];
        int am = 0;
        for (int c = 1; c < C; ++c) if (l[c] > mx) { mx = l[c]; am = c; }
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(l[c] - mx);
        const float ce = (mx + logf(s)) - l[cls];
        const float pa = (float)p.active[am];            // active_class_ids[0] (model.py:1182)
        atomicAdd(&p.scal[S_C_SUM], ce * pa);
        atomicAdd(&p.scal[S_C_PA], pa);
        if (cls > 0) {
            const float* t = p.tbbox + row * 4;
            const float* q = p.mbbox + (row * C + cls) * 4;
            atomicAdd(&p.scal[S_B_SUM], smooth_l1(t[0], q[0]) + smooth_l1(t[1], q[1]) + smooth_l1(t[2], q[2]) + smooth_l1(t[3], q[3]));
            atomicAdd(&p.scal[S_POS], 1.f);
        }
    }
    if (cls <= 0) return;
    const int npix = p.mh * p.mw;
    const float eps = 1e-7f;
    float ms = 0.f, sy = 0.f, sp = 0.f;
    for (int i = tid; i < npix; i += 256) {
        const float y = p.tmask[row * npix + i];
        const float q = p.mmask[(row * npix + i) * C + cls];
        if (p.dice) {
            ms += y * q; sy += y; sp += q;
        } else {
            const float qc = fminf(fmaxf(q, eps), 1.f - eps);
            const float x = logf(qc / (1.f - qc));
            ms += fmaxf(x, 0.f) - x * y + logf(1.f + expf(-fabsf(x)));
        }
    }
    ms = block_sum(ms, sbuf);
    if (p.dice) { sy = block_sum(sy, sbuf); sp = block_sum(sp, sbuf); }
    if (tid == 0) {
        atomicAdd(&p.scal[S_M_SUM], ms);
        if (p.dice) { atomicAdd(&p.scal[S_D_Y], sy); atomicAdd(&p.scal[S_D_P], sp); }
    }
}

// ---- phase 2a: RPN gradients ------------------------------------------------------------------
__global__ void rpn_loss_grad_kernel(const LossArgs p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.B * p.A) return;
    const int b = (int)(i / p.A);
    const int m = p.rpn_match[i];
    const float ccnt = p.scal[S_RC_CNT], bcnt = p.scal[S_RB_CNT];
    float g0 = 0.f, g1 = 0.f;
    if (m != 0) {
        const float l0 = p.rpn_logits[i * 2], l1 = p.rpn_logits[i * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), s = e0 + e1;
        const float k = p.w0 / ccnt;
        g0 = (e0 / s - (m == 1 ? 0.f : 1.f)) * k;
        g1 = (e1 / s - (m == 1 ? 1.f : 0.f)) * k;
    }
    p.d_rpn_logits[i * 2] = g0;
    p.d_rpn_logits[i * 2 + 1] = g1;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    const int rk = p.rank[i];
    if (rk >= 0 && rk < p.max_rpn_pos) {
        const float* t = p.rpn_bbox_t + ((int64_t)b * p.max_rpn_pos + rk) * 4;
        const float* q = p.rpn_bbox + i * 4;
        const float k = p.w1 / (bcnt * 4.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = smooth_l1_grad(t[e], q[e]) * k;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) p.d_rpn_bbox[i * 4 + e] = d[e];
}

// ---- phase 2b: head gradients + final loss values -------------------------------------------------
__global__ __launch_bounds__(256) void heads_loss_grad_kernel(const LossArgs p) {
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, C = p.C;
    const int cls = p.tcls[row];
    const float pa_sum = p.scal[S_C_PA], npos = p.scal[S_POS];
    const int npix = p.mh * p.mw;
    if (row == 0 && tid == 0) {
        const float ccnt = p.scal[S_RC_CNT], bcnt = p.scal[S_RB_CNT];
        p.losses[0] = ccnt > 0.f ? p.scal[S_RC_SUM] / ccnt : 0.f;
        p.losses[1] = bcnt > 0.f ? p.scal[S_RB_SUM] / (bcnt * 4.f) : 0.f;
        p.losses[2] = p.scal[S_C_SUM] / pa_sum;
        p.losses[3] = npos > 0.f ? p.scal[S_B_SUM] / (npos * 4.f) : 0.f;
        if (p.dice) {
            const float sm = 1e-7f;
            p.losses[4] = npos > 0.f ? 1.f - (2.f * p.scal[S_M_SUM] + sm) / (p.scal[S_D_Y] + p.scal[S_D_P] + sm) : 0.f;
        } else {
            p.losses[4] = npos > 0.f ? p.scal[S_M_SUM] / (npos * (float)npix) : 0.f;
        }
    }
    if (tid == 0) {
        const float* l = p.cls_logits + row * C;
        float mx = l[0];
        int am = 0;
        for (int c = 1; c < C; ++c) if (l[c] > mx) { mx = l[c]; am = c; }
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(l[c] - mx);
        const float k = p.w2 * (float)p.active[am] / pa_sum;
        for (int c = 0; c < C; ++c)
            p.d_cls_logits[row * C + c] = (expf(l[c] - mx) / s - (c == cls ? 1.f : 0.f)) * k;
    }
    for (int i = tid; i < C * 4; i += 256) {
        float g = 0.f;
        const int c = i >> 2, e = i & 3;
        if (cls > 0 && c == cls) g = smooth_l1_grad(p.tbbox[row * 4 + e], p.mbbox[(row * C + c) * 4 + e]) * (p.w3 / (npos * 4.f));
        p.d_mbbox[row * C * 4 + i] = g;
    }
    const float eps = 1e-7f;
    const float dsm = 1e-7f;
    const float D = p.scal[S_D_Y] + p.scal[S_D_P] + dsm, I2 = 2.f * p.scal[S_M_SUM] + dsm;
    for (int i = tid; i < npix * C; i += 256) {
        float g = 0.f;
        const int c = i % C;
        if (cls > 0 && c == cls) {
            const int pix = i / C;
            const float y = p.tmask[row * npix + pix];
            const float q = p.mmask[row * npix * C + i];
            if (p.dice) {
                g = -(2.f * y * D - I2) / (D * D) * p.w4;
            } else if (q >= eps && q <= 1.f - eps) {
                // TF autodiff of max(x,0) - x*y + log1p(exp(-|x|)) with x = log(q/(1-q)):
                // relu'(0) = 0 and sign(0) = 0, so the value at x == 0 is -y (not sigmoid(0) - y)
                const float x = logf(q / (1.f - q));
                const float e = expf(-fabsf(x));
                const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
                const float gx = (x > 0.f ? 1.f : 0.f) - y - sg * e / (1.f + e);
                g = gx / (q * (1.f - q)) * (p.w4 / (npos * (float)npix));
            }
        }
        p.d_mmask[row * npix * C + i] = g;
    }
}

__global__ void loss_zero_kernel(float* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.f;
}

extern "C" size_t mrcnn_losses_workspace(const mrcnn_loss_desc* d) {
    if (!d || d->B <= 0 || d->A <= 0) return 0;
    return 256 + S_NUM * sizeof(float) + (size_t)d->B * d->A * sizeof(int32_t);
}

extern "C" int mrcnn_losses_fwd_bwd(const mrcnn_loss_desc* d, const int32_t* rpn_match, const float* rpn_bbox_t,
                                    const float* rpn_class_logits, const float* rpn_bbox,
                                    const int32_t* target_class_ids, const float* target_bbox,
                                    const float* target_mask, const int32_t* active_class_ids,
                                    const float* mrcnn_class_logits, const float* mrcnn_bbox,
                                    const float* mrcnn_mask, float* losses, float* d_rpn_class_logits,
                                    float* d_rpn_bbox, float* d_mrcnn_class_logits, float* d_mrcnn_bbox,
                                    float* d_mrcnn_mask, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !rpn_match || !rpn_bbox_t || !rpn_class_logits || !rpn_bbox || !target_class_ids || !target_bbox ||
        !target_mask || !active_class_ids || !mrcnn_class_logits || !mrcnn_bbox || !mrcnn_mask || !losses ||
        !d_rpn_class_logits || !d_rpn_bbox || !d_mrcnn_class_logits || !d_mrcnn_bbox || !d_mrcnn_mask || !workspace)
        return MRCNN_ERR_ARG;
    if (d->B <= 0 || d->A <= 0 || d->T <= 0 || d->C <= 1 || d->mask_h <= 0 || d->mask_w <= 0 || d->max_rpn_pos <= 0)
        return MRCNN_ERR_ARG;
    if (workspace_bytes < mrcnn_losses_workspace(d)) return MRCNN_ERR_WORKSPACE;
    LossArgs a;
    a.rpn_match = rpn_match; a.rpn_bbox_t = rpn_bbox_t; a.rpn_logits = rpn_class_logits; a.rpn_bbox = rpn_bbox;
    a.tcls = target_class_ids; a.tbbox = target_bbox; a.tmask = target_mask; a.active = active_class_ids;
    a.cls_logits = mrcnn_class_logits; a.mbbox = mrcnn_bbox; a.mmask = mrcnn_mask; a.losses = losses;
    a.d_rpn_logits = d_rpn_class_logits; a.d_rpn_bbox = d_rpn_bbox; a.d_cls_logits = d_mrcnn_class_logits;
    a.d_mbbox = d_mrcnn_bbox; a.d_mmask = d_mrcnn_mask;
    uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    a.scal = reinterpret_cast<float*>(base);
    a.rank = reinterpret_cast<int32_t*>(base + S_NUM * sizeof(float));
    a.B = d->B; a.A = d->A; a.T = d->T; a.C = d->C; a.mh = d->mask_h; a.mw = d->mask_w;
    a.max_rpn_pos = d->max_rpn_pos; a.dice = d->mask_loss_dice;
    a.w0 = d->w[0]; a.w1 = d->w[1]; a.w2 = d->w[2]; a.w3 = d->w[3]; a.w4 = d->w[4];
    hipStream_t s = (hipStream_t)stream;
    // a kernel, not hipMemsetAsync: this call sits inside captured training steps, and a captured memset node writes a
    // garbage value from the second replay on (ROCm 7.2; tools/graph_memset_probe.py)
    hipLaunchKernelGGL(loss_zero_kernel, dim3(1), dim3(64), 0, s, a.scal, S_NUM);
    hipLaunchKernelGGL(rpn_loss_reduce_kernel, dim3(d->B), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(heads_loss_reduce_kernel, dim3((unsigned)(d->B * d->T)), dim3(256), 0, s, a);
    const int64_t na = (int64_t)d->B * d->A;
    hipLaunchKernelGGL(rpn_loss_grad_kernel, dim3((unsigned)cdiv64(na, 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(heads_loss_grad_kernel, dim3((unsigned)(d->B * d->T)), dim3(256), 0, s, a);
    return mrcnn_launch_status();
}
