// Calibration of FETCH_SIZE for the access pattern of the L2 prefetch in miz_step_kernel: one 4-byte
// LDS-DMA load per 32-B sector (64 lanes x 32 B = 2 KiB per wave-instruction), data discarded.
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; the region is 1 GiB (4x the Infinity Cache),
// so every sector comes from HBM: FETCH_SIZE x 1024 / 2^30 is the counter's scale for this pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) touch(const char *base, size_t bytes) {
    __shared__ unsigned sink[256];
    const size_t per_block = 256 * 32 * 16;                       // 16 sectors per lane
    const char *p = base + (size_t)blockIdx.x * per_block + (size_t)threadIdx.x * 32;
    auto *dst = (__attribute__((address_space(3))) void *)(sink + (threadIdx.x & ~63));
    for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + (size_t)i * 256 * 32), dst, 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (bytes == 1 && sink[threadIdx.x] == 0x12345678u) ((volatile char *)base)[0] = 1;   // keep the loads
}

int main() {
    const size_t bytes = 1ull << 30;
    char *d;
    CHK(hipMalloc(&d, bytes));
    CHK(hipMemset(d, 0, bytes));
    const size_t per_block = 256 * 32 * 16;
    for (int rep = 0; rep < 3; ++rep) touch<<<(unsigned)(bytes / per_block), 256>>>(d, bytes);
    CHK(hipDeviceSynchronize());
    printf("touched %zu bytes per launch, one dword per 32-B sector\n", bytes);
    return 0;
}
