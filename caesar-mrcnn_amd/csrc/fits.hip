// FITS tile -> network input image on the device: the numeric part of utils.read_fits (mrcnn/utils.py:1033-1163) as the
// run.py path calls it (stretch = normalize = convertToRGB = to_uint8 = True, no bias / contrast stretch):
//   out = data.astype(float32); out[isnan(out)] = nanmin(out)                                   (:1088-1091)
//   per channel c (zscale contrast zc):  ZScaleInterval(contrast=zc)(out)  -> [0, 1]            (:1101-1111, stretch_img :1166-1172)
//                                        / max                                                  (normalize_img :1182-1188)
//   uint8 RGB = round(255 * channel)                                                            (gray2rgb :1190-1208)
// astropy's ZScaleInterval [3P, SURVEY App. C-6] is the restatement of caesar-mrcnn_amd/fits.py:zscale_limits (<= 1000 strided
// samples of the finite values, sorted; <= 5 rounds of a straight-line fit with 2.5 sigma rejection grown by 1 % of the samples;
// limits = median -/+ slope / contrast, clamped to the sample range).  Three launches: per-chunk statistics (+ byte swap of the
// big-endian FITS floats), ONE workgroup for sampling / sort / fit, one pass that writes the bytes.  The per-pixel arithmetic is
// the host's, type for type (float32 subtraction, float64 division and clip, float32 normalisation, round-half-even), so the
// uint8 image is identical to fits.read_fits's; the fit uses the closed form of the weighted least-squares line where the host
// goes through numpy.polyfit's SVD (agreement ~1e-15 relative in the limits: far below half a grey level).
#include "common.h"
#include <math.h>

#define FITS_CHUNK 1024          // pixels per statistics chunk
#define FITS_MAX_CHUNKS 4096     // one workgroup scans the chunk table in LDS: tiles up to 2048 x 2048
#define FITS_NSAMPLES 1000

struct FitsChunk { float mn, mx; int n_nan, n_inf; };
struct FitsChannel { float vmin32; int divide; double range; float cmax; int pad; };
struct FitsParams { float fill; int all_nan; FitsChannel ch[3]; };

__device__ __forceinline__ float fits_load(const unsigned* raw, long long i, int big_endian) {
    unsigned u = raw[i];
    if (big_endian) u = __builtin_bswap32(u);
    return __uint_as_float(u);
}

// chunk statistics; vals gets the native-endian copy
__global__ __launch_bounds__(256) void fits_stats_kernel(const unsigned* __restrict__ raw, float* __restrict__ vals, long long n,
                                                         int big_endian, FitsChunk* __restrict__ chunks) {
    __shared__ float s_mn[256], s_mx[256];
    __shared__ int s_nan[256], s_inf[256];
    const int tid = threadIdx.x;
    const long long base = (long long)blockIdx.x * FITS_CHUNK;
    float mn = INFINITY, mx = -INFINITY;
    int nn = 0, ni = 0;
    for (int k = tid; k < FITS_CHUNK; k += 256) {
        const long long i = base + k;
        if (i >= n) break;
        const float v = fits_load(raw, i, big_endian);
        vals[i] = v;
        if (v != v) { ++nn; continue; }
        if (isinf(v)) ++ni;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    s_mn[tid] = mn; s_mx[tid] = mx; s_nan[tid] = nn; s_inf[tid] = ni;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            s_mn[tid] = fminf(s_mn[tid], s_mn[tid + s]);
            s_mx[tid] = fmaxf(s_mx[tid], s_mx[tid + s]);
            s_nan[tid] += s_nan[tid + s];
            s_inf[tid] += s_inf[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) chunks[blockIdx.x] = {s_mn[0], s_mx[0], s_nan[0], s_inf[0]};
}

__device__ __forceinline__ double fits_block_sum(double v, double* red, int tid) {
    red[tid] = v;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// one channel's value of a pixel after stretch + normalisation input (stretch_img then .astype(float32))
__device__ __forceinline__ float fits_stretch(float v, const FitsChannel& c) {
    double o = (double)(v - c.vmin32);                  // np.subtract(float32 array, python float) stays float32, then .astype(float64)
    if (c.divide) o = o / c.range;
    o = o < 0.0 ? 0.0 : o;                              // np.clip(out, 0, 1)
    o = o > 1.0 ? 1.0 : o;
    return (float)o;
}

// sampling, sort, iterative fit and limits: ONE workgroup of 1024 threads
__global__ __launch_bounds__(1024) void fits_zscale_kernel(const float* __restrict__ vals, long long n, int nchunks,
                                                           const FitsChunk* __restrict__ chunks, double c0, double c1, double c2,
                                                           FitsParams* __restrict__ params) {
    __shared__ int s_prefix[FITS_MAX_CHUNKS + 1];
    __shared__ float s_samp[1024];
    __shared__ double s_red[1024];
    __shared__ unsigned char s_bad[1024], s_bad2[1024];
    const int tid = threadIdx.x;
    // ---- global min / max / counts ------------------------------------------------------------------------------------
    float mn = INFINITY, mx = -INFINITY;
    int nn = 0;
    for (int k = tid; k < nchunks; k += 1024) {
        const FitsChunk c = chunks[k];
        mn = fminf(mn, c.mn); mx = fmaxf(mx, c.mx); nn += c.n_nan;
    }
    s_samp[tid] = mn;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) s_samp[tid] = fminf(s_samp[tid], s_samp[tid + s]); __syncthreads(); }
    const float fill = s_samp[0];                       // np.nanmin(out): +inf only when every pixel is NaN
    __syncthreads();
    s_samp[tid] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) s_samp[tid] = fmaxf(s_samp[tid], s_samp[tid + s]); __syncthreads(); }
    const float dmax = s_samp[0];
    __syncthreads();
    const long long total_nan = (long long)fits_block_sum((double)nn, s_red, tid);
    if (total_nan == n) {                               // nothing to scale: the host gets NaN everywhere (uint8 of NaN: 0)
        if (tid == 0) { params->all_nan = 1; params->fill = fill; }
        return;
    }
    const bool fill_finite = !isinf(fill);
    // ---- finite pixels per chunk, exclusive prefix ------------------------------------------------------------------------
    for (int k = tid; k < nchunks; k += 1024) {
        const long long left = n - (long long)k * FITS_CHUNK;
        const int size = left < FITS_CHUNK ? (int)left : FITS_CHUNK;
        s_prefix[k + 1] = size - chunks[k].n_inf - (fill_finite ? 0 : chunks[k].n_nan);
    }
    if (tid == 0) s_prefix[0] = 0;
    __syncthreads();
    if (tid == 0) for (int k = 0; k < nchunks; ++k) s_prefix[k + 1] += s_prefix[k];     // <= 4096 adds: microseconds
    __syncthreads();
    const long long nfinite = s_prefix[nchunks];
    int npix = 0, stride = 1;
    if (nfinite > 0) {
        stride = (int)fmax(1.0, (double)nfinite / (double)FITS_NSAMPLES);
        const long long avail = (nfinite + stride - 1) / stride;
        npix = avail < FITS_NSAMPLES ? (int)avail : FITS_NSAMPLES;
    }
    // ---- values[isfinite][::stride][:nsamples] ---------------------------------------------------------------------------
    float sv = INFINITY;
    if (tid < npix) {
        const long long rank = (long long)tid * stride;
        int lo = 0, hi = nchunks;                       // last chunk with prefix <= rank
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_prefix[mid] <= rank) lo = mid; else hi = mid; }
        int need = (int)(rank - s_prefix[lo]);
        const long long base = (long long)lo * FITS_CHUNK;
        const long long left = n - base;
        const int size = left < FITS_CHUNK ? (int)left : FITS_CHUNK;
        if (s_prefix[lo + 1] - s_prefix[lo] == size) {
            float v = vals[base + need];
            sv = v != v ? fill : v;
        } else {
            for (int k = 0; k < size; ++k) {
                float v = vals[base + k];
                if (v != v) v = fill;
                if (isinf(v)) continue;
                if (need-- == 0) { sv = v; break; }
            }
        }
    }
    s_samp[tid] = sv;
    __syncthreads();
    // ---- bitonic sort, ascending; the +inf padding ends up behind the npix samples ----------------------------------------
    for (int k = 2; k <= 1024; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = tid ^ j;
            if (p > tid) {
                const float a = s_samp[tid], b = s_samp[p];
                const bool up = (tid & k) == 0;
                if ((a > b) == up) { s_samp[tid] = b; s_samp[p] = a; }
            }
            __syncthreads();
        }
    // ---- ZScaleInterval.get_limits -------------------------------------------------------------------------------------------
    double slope = 0.0;
    int ngood = npix;
    if (npix > 0) {
        const int minpix = max(5, (int)(npix * 0.5));
        const int ngrow = max(1, (int)(npix * 0.01));
        const int back = ngrow / 2, fwd = (ngrow - 1) / 2;             // np.convolve(bad, ones(ngrow), 'same'): out[i] = OR bad[i - back .. i + fwd]
        int last = npix + 1;
        const bool in = tid < npix;
        const double x = (double)tid, y = in ? (double)s_samp[tid] : 0.0;
        s_bad[tid] = 0;
        __syncthreads();
        for (int it = 0; it < 5; ++it) {
            if (ngood >= last || ngood < minpix) break;
            const double w = (in && !s_bad[tid]) ? 1.0 : 0.0;
            // weighted least-squares line through the good samples (np.polyfit(x, samples, 1, w = good)), centred form
            const double sw = fits_block_sum(w, s_red, tid);
            const double xm = fits_block_sum(w * x, s_red, tid) / sw, ym = fits_block_sum(w * y, s_red, tid) / sw;
            const double sxx = fits_block_sum(w * (x - xm) * (x - xm), s_red, tid);
            const double sxy = fits_block_sum(w * (x - xm) * (y - ym), s_red, tid);
            slope = sxx > 0.0 ? sxy / sxx : 0.0;
            const double icpt = ym - slope * xm;
            const double flat = y - (slope * x + icpt);
            const double fm = fits_block_sum(w * flat, s_red, tid) / sw;
            const double var = fits_block_sum(w * (flat - fm) * (flat - fm), s_red, tid) / sw;
            const double thr = 2.5 * sqrt(var);
            if (in && (flat < -thr || flat > thr)) s_bad[tid] = 1;
            __syncthreads();
            unsigned char b = 0;
            if (in) {
                const int a0 = max(0, tid - back), a1 = min(npix - 1, tid + fwd);
                for (int k = a0; k <= a1; ++k) b |= s_bad[k];
            }
            s_bad2[tid] = b;
            __syncthreads();
            s_bad[tid] = s_bad2[tid];
            last = ngood;
            ngood = (int)fits_block_sum((in && !s_bad2[tid]) ? 1.0 : 0.0, s_red, tid);
        }
        if (tid == 0) {
            const float smin = s_samp[0], smax = s_samp[npix - 1];
            float median;                                               // np.median of a float32 array: float32
            if (npix & 1) median = s_samp[npix / 2];
            else median = (s_samp[npix / 2 - 1] + s_samp[npix / 2]) / 2.0f;
            const int center = (npix - 1) / 2;
            const double contrast[3] = {c0, c1, c2};
            for (int c = 0; c < 3; ++c) {
                // vmin / vmax are numpy float32 scalars (the sample ends) or float64 (the fitted limits): the type of their
                // difference -- float32 subtraction or float64 -- follows
                double vmin = (double)smin, vmax = (double)smax;
                bool vmin32 = true, vmax32 = true;
                if (ngood >= minpix) {
                    double sl = slope;
                    if (contrast[c] > 0.0) sl = sl / contrast[c];
                    const double lo = (double)median - (double)(center - 1) * sl, hi = (double)median + (double)(npix - center) * sl;
                    if (lo > vmin) { vmin = lo; vmin32 = false; }         // max(vmin, lo): the first argument wins a tie
                    if (hi < vmax) { vmax = hi; vmax32 = false; }
                }
                FitsChannel ch;
                ch.vmin32 = (float)vmin;
                ch.range = (vmin32 && vmax32) ? (double)(smax - smin) : vmax - vmin;
                ch.divide = ch.range != 0.0;
                ch.pad = 0;
                ch.cmax = 0.f;
                ch.cmax = fits_stretch(dmax, ch);                       // every step of the map is monotone: max(channel) = channel(max)
                params->ch[c] = ch;
            }
            params->fill = fill;
            params->all_nan = 0;
        }
    } else if (tid == 0) {                                                // no finite pixel at all (only +-inf): numpy's sort of an empty
        params->all_nan = 1;                                               // sample raises on the host; the device writes zeros
        params->fill = fill;
    }
}

__global__ __launch_bounds__(256) void fits_rgb_kernel(const float* __restrict__ vals, long long n, const FitsParams* __restrict__ params,
                                                       unsigned char* __restrict__ rgb) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const FitsParams p = *params;
    unsigned char o[3] = {0, 0, 0};
    if (!p.all_nan) {
        float v = vals[i];
        if (v != v) v = p.fill;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = fits_stretch(v, p.ch[c]);
            const float q = rintf((s / p.ch[c].cmax) * 255.f);           // normalize_img (float32), gray2rgb: (c * 255).round()
            o[c] = (q >= 0.f && q <= 255.f) ? (unsigned char)q : (unsigned char)0;     // 0 / 0 (constant tile): NaN -> 0
        }
    }
    rgb[i * 3 + 0] = o[0]; rgb[i * 3 + 1] = o[1]; rgb[i * 3 + 2] = o[2];
}

extern "C" size_t mrcnn_fits_workspace(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    const long long n = (long long)H * W;
    const long long chunks = (n + FITS_CHUNK - 1) / FITS_CHUNK;
    return (size_t)n * sizeof(float) + (size_t)chunks * sizeof(FitsChunk) + 256;
}

extern "C" int mrcnn_fits_to_rgb(const void* raw, int big_endian, int H, int W, const double* zscale_contrasts, void* rgb, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!raw || !rgb || !zscale_contrasts || H <= 0 || W <= 0) return MRCNN_ERR_ARG;
    const long long n = (long long)H * W;
    const long long chunks = (n + FITS_CHUNK - 1) / FITS_CHUNK;
    if (chunks > FITS_MAX_CHUNKS) return MRCNN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < mrcnn_fits_workspace(H, W)) return MRCNN_ERR_WORKSPACE;
    float* vals = (float*)workspace;
    FitsChunk* ch = (FitsChunk*)(vals + n);
    FitsParams* params = (FitsParams*)(((uintptr_t)(ch + chunks) + 63) & ~(uintptr_t)63);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fits_stats_kernel, dim3((unsigned)chunks), dim3(256), 0, s, (const unsigned*)raw, vals, n, big_endian, ch);
    hipLaunchKernelGGL(fits_zscale_kernel, dim3(1), dim3(1024), 0, s, vals, n, (int)chunks, ch, zscale_contrasts[0], zscale_contrasts[1],
                       zscale_contrasts[2], params);
    hipLaunchKernelGGL(fits_rgb_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, vals, n, params, (unsigned char*)rgb);
    return mrcnn_launch_status();
}
