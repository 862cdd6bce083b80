// Micro-benchmark: per-CU streaming rates of loads only, stores only and both, one 1024-thread
// workgroup per 4096-cell meridian (32 contiguous bytes per lane), as in miz_step_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0 loads+stores, 1 loads only (one tiny store), 2 stores only
__global__ void __launch_bounds__(1024) k(double *state, long long fstride, int pitch, double *sink) {
    const int t = threadIdx.x, col = blockIdx.x;
    double *base = state + (size_t)col * pitch + 4 * t;
    double acc = 0.0;
    double2 v[6][2];
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        if (MODE != 2) {
            v[f][0] = *reinterpret_cast<const double2 *>(base + f * fstride);
            v[f][1] = *reinterpret_cast<const double2 *>(base + f * fstride + 2);
        } else {
            v[f][0].x = v[f][0].y = v[f][1].x = v[f][1].y = (double)(t + f);
        }
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        if (MODE != 1) {
            *reinterpret_cast<double2 *>(base + f * fstride) = v[f][0];
            *reinterpret_cast<double2 *>(base + f * fstride + 2) = v[f][1];
        } else {
            acc += v[f][0].x + v[f][0].y + v[f][1].x + v[f][1].y;
        }
    }
    if (MODE == 1 && acc == 12345.678) sink[0] = acc;
}

int main() {
    const int nlat = 4096, ncol = 2048;
    const long long fstride = (long long)nlat * ncol;
    double *d, *sink;
    CHK(hipMalloc(&d, sizeof(double) * fstride * 6));
    CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(d, 0, sizeof(double) * fstride * 6));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char *names[3] = {"loads+stores (96 B/cell)", "loads only (48 B/cell)", "stores only (48 B/cell)"};
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) {
                if (mode == 0) k<0><<<ncol, 1024>>>(d, fstride, nlat, sink);
                else if (mode == 1) k<1><<<ncol, 1024>>>(d, fstride, nlat, sink);
                else k<2><<<ncol, 1024>>>(d, fstride, nlat, sink);
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = (mode == 0 ? 96.0 : 48.0) * nlat * ncol * 20;
            double us = ms * 1000 / 20;
            printf("%-28s %.1f us/launch, %.0f GB/s, %.1f B/cycle/CU @2.4GHz\n", names[mode], us,
                   bytes / (ms * 1e-3) / 1e9, bytes / 20 / 256 / (us * 1e-6 * 2.4e9));
        }
    // Per-CU rate when only a fraction of the CUs stream: one round of `n` workgroups.
    for (int n : {2048, 256, 128, 64, 16}) {
        for (int mode = 1; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHK(hipEventRecord(e0));
                if (mode == 1) k<1><<<n, 1024>>>(d, fstride, nlat, sink);
                else k<2><<<n, 1024>>>(d, fstride, nlat, sink);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double bytes = 48.0 * nlat * n;
            int cus = n < 256 ? n : 256;
            printf("%4d workgroups, %s: %.1f us, %.1f B/cycle per busy CU\n", n, mode == 1 ? "loads " : "stores",
                   best * 1000, bytes / cus / (best * 1e-3 * 2.4e9));
        }
    }
    return 0;
}
