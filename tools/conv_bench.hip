// Standalone timing harness for the implicit-GEMM convolution kernels (no torch): builds against the
// same source as libmrcnn_hip.so and times the C-ABI entry points with hipEvents.
//   conv_bench [N H W Cin Cout K stride reps]      default: the mask-head 3x3 shape of bench.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mrcnn_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    int N = 1024, H = 14, W = 14, Cin = 256, Cout = 256, K = 3, stride = 1, reps = 20;
    if (argc > 7) { N = atoi(argv[1]); H = atoi(argv[2]); W = atoi(argv[3]); Cin = atoi(argv[4]); Cout = atoi(argv[5]); K = atoi(argv[6]); stride = atoi(argv[7]); }
    if (argc > 8) reps = atoi(argv[8]);
    mrcnn_conv_desc d = {};
    d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.KH = K; d.KW = K; d.stride = stride;
    d.pad_t = d.pad_l = (K - 1) / 2; d.OH = (H + stride - 1) / stride; d.OW = (W + stride - 1) / stride;
    d.act = 1; d.res_mode = 0; d.out_mode = 0; d.cmod = Cout;
    d.out_w_stride = Cout; d.out_h_stride = (int64_t)d.OW * Cout; d.out_n_stride = (int64_t)d.OH * d.OW * Cout;
    size_t nx = (size_t)N * H * W * Cin, nw = (size_t)K * K * Cin * Cout, no = (size_t)N * d.OH * d.OW * Cout;
    std::vector<float> hx(nx), hw(nw);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    float *x, *w, *b, *o, *dw, *ws;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&b, Cout * 4)); CK(hipMalloc(&o, no * 4));
    CK(hipMalloc(&dw, nw * 4));
    size_t wsb = mrcnn_conv2d_wgrad_workspace(&d);
    size_t wsf = mrcnn_conv2d_fwd_workspace(&d);
    if (wsf > wsb) wsb = wsf;
    CK(hipMalloc(&ws, wsb + 256));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b, 0, Cout * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double flop = 2.0 * N * d.OH * d.OW * (double)Cout * K * K * Cin;
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 3; ++i) {
            int rc = mode == 0 ? mrcnn_conv2d_fwd_ws(&d, x, w, b, nullptr, nullptr, nullptr, o, nullptr, ws, wsb, nullptr)
                               : mrcnn_conv2d_wgrad(&d, x, o, dw, ws, wsb, 0, nullptr);
            if (rc) { printf("rc=%d\n", rc); return 1; }
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) {
            if (mode == 0) mrcnn_conv2d_fwd_ws(&d, x, w, b, nullptr, nullptr, nullptr, o, nullptr, ws, wsb, nullptr);
            else mrcnn_conv2d_wgrad(&d, x, o, dw, ws, wsb, 0, nullptr);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%s N=%d %dx%d Cin=%d Cout=%d k=%d s=%d : %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", mode ? "wgrad" : "fwd  ", N, H, W,
               Cin, Cout, K, stride, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    }
    return 0;
}
