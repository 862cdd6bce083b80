// Micro-benchmark: HBM throughput of a 6-field read + 6-field write stream (the MIZ state
// traffic) with (a) lane-consecutive 16-B accesses and (b) "chunk" accesses where each lane
// owns 32 contiguous bytes (two 16-B accesses at lane stride 32 B), 1024-thread workgroups,
// one workgroup per 4096-cell meridian, as in miz_step_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) stream_kernel(double *state, long long fstride, int pitch) {
    const int t = threadIdx.x, col = blockIdx.x;
    double *base = state + (size_t)col * pitch;
    double2 v[6][2];
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const double *p = base + f * fstride;
        if (MODE == 0) {   // coalesced: lane-consecutive 16 B, two passes of T*2 doubles
            v[f][0] = *reinterpret_cast<const double2 *>(p + 2 * t);
            v[f][1] = *reinterpret_cast<const double2 *>(p + 2 * (t + 1024));
        } else {           // chunk: 32 contiguous bytes per lane
            v[f][0] = *reinterpret_cast<const double2 *>(p + 4 * t);
            v[f][1] = *reinterpret_cast<const double2 *>(p + 4 * t + 2);
        }
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        double *p = base + f * fstride;
        double2 a = v[f][0], b = v[f][1];
        a.x += 1.0; a.y += 1.0; b.x += 1.0; b.y += 1.0;
        if (MODE == 0) {
            *reinterpret_cast<double2 *>(p + 2 * t) = a;
            *reinterpret_cast<double2 *>(p + 2 * (t + 1024)) = b;
        } else {
            *reinterpret_cast<double2 *>(p + 4 * t) = a;
            *reinterpret_cast<double2 *>(p + 4 * t + 2) = b;
        }
    }
}

int main() {
    const int nlat = 4096, ncol = 2048;
    const long long fstride = (long long)nlat * ncol;
    double *d;
    CHK(hipMalloc(&d, sizeof(double) * fstride * 6));
    CHK(hipMemset(d, 0, sizeof(double) * fstride * 6));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) {
                if (mode == 0) stream_kernel<0><<<ncol, 1024>>>(d, fstride, nlat);
                else stream_kernel<1><<<ncol, 1024>>>(d, fstride, nlat);
            }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = 96.0 * nlat * ncol * 20;
            printf("mode %d (%s): %.1f us/launch, %.0f GB/s\n", mode, mode ? "chunk 32B/lane" : "coalesced", ms * 1000 / 20, bytes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
