// Operand staging alone, two traffic patterns of a 16-bit 3x3 mask-head layer (2048 ROIs of 14 x 14, 256 -> 256 channels), no MFMA and
// no operand reads: what the stream into LDS costs by itself.
//   mode 0  what the 16-bit kernels do today: every K-step (tap, 64-channel chunk) stages 256 pixel rows x 128 B (shifted by the tap)
//           and 256 weight rows x 128 B: 36 x 64 KiB per 256 x 256 tile, 3.7 GB per layer
//   mode 1  the pixel slab of a channel chunk staged ONCE with its halo (256 + 2 x 15 rows -> 288 x 128 B), then the nine taps'
//           weight stages only: 4 x (36 + 9 x 32) KiB per tile, 2.03 GB per layer
// One persistent workgroup per CU (128 KiB ring), 1 KiB LDS-DMA pieces (8 rows x 128 B per wave instruction), stage k + 1 issued
// before stage k is waited for, one barrier per stage -- the skeleton of conv_fwd_h16_kernel<T, 4, 2, 2, 4>.
// usage: h16_stage_probe [rois]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int MODE>
__global__ __launch_bounds__(256, 1) void stage_kernel(const char* x, const char* w, unsigned x_bytes, unsigned w_bytes, int tiles, int* sink) {
    __shared__ __attribute__((aligned(16))) char lds[147456];      // weight / pixel ring 2 x 32 (64) KiB, then (mode 1) two 36 KiB slabs
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    const unsigned lane_row = lane >> 3, lane_chunk = (lane & 7) * 16;
    int stage = 0;
    auto issue_rows = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, long long row0, unsigned row_bytes, unsigned col_byte, int pieces) {
        // `pieces` 1 KiB pieces of 8 rows x 128 B, this wave takes pieces wave, wave + 4, ...
        for (int p = wave; p < pieces; p += 4) {
            const long long row = row0 + p * 8 + lane_row;
            const unsigned off = row < 0 ? 0xFFFFFFF0u : (unsigned)(row * row_bytes + col_byte + lane_chunk);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(dst + p * 1024), 16, off, 0, 0, 0);
        }
    };
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long long m0 = (long long)t * 256;
        if (MODE == 1) issue_rows(rx, lds + 65536, m0 - 16, 512, 0, 36);            // the tile's first slab
        for (int chunk = 0; chunk < 4; ++chunk) {
            for (int tap = 0; tap < 9; ++tap) {
                char* buf = lds + (stage & 1) * (MODE == 1 ? 32768 : 65536);
                if (MODE == 0) issue_rows(rx, buf, m0 + (tap / 3 - 1) * 14 + (tap % 3 - 1), 512, chunk * 128, 32);
                issue_rows(rw, buf + (MODE == 0 ? 32768 : 0), 0, 4608, (tap * 256 + chunk * 64) * 2, 32);
                ++stage;
                const bool slab = MODE == 1 && tap == 0 && chunk < 3;
                if (slab) issue_rows(rx, lds + 65536 + ((chunk + 1) & 1) * 36864, m0 - 16, 512, (chunk + 1) * 128, 36);   // next chunk's slab: 8 K-steps to land
                // everything but what was issued in this step has landed
                if (MODE == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (slab) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && lds[blockIdx.x & 1023] == 77) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int rois = argc > 1 ? atoi(argv[1]) : 2048;
    const long long M = (long long)rois * 196;
    const int tiles = (int)((M + 255) / 256);
    const size_t xb = (size_t)M * 512, wb = (size_t)256 * 4608;
    char *x, *w; int* sink;
    hipMalloc(&x, xb + 65536); hipMalloc(&w, wb); hipMalloc(&sink, 4);
    hipMemset(x, 1, xb + 65536); hipMemset(w, 1, wb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb[2] = {tiles * 36.0 * 65536 / 1e9, tiles * 4.0 * (36864 + 9 * 32768) / 1e9};
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (mode == 0) stage_kernel<0><<<256, 256>>>(x, w, (unsigned)xb, (unsigned)wb, tiles, sink);
                else stage_kernel<1><<<256, 256>>>(x, w, (unsigned)xb, (unsigned)wb, tiles, sink);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("mode %d (%s): %.3f ms per layer, %.2f GB staged, %.2f TB/s into LDS\n", mode,
                   mode ? "pixel slab once per channel chunk + weights per tap" : "pixels and weights per K-step (today)", ms, gb[mode], gb[mode] / ms);
        }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
