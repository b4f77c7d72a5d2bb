// Throughput of returnless global atomic adds on gfx950 by memory scope and by address pattern: what the pileup
// kernel's deviation path pays.  hipcc --offload-arch=gfx950 -O2 atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int SCOPE>
__global__ __launch_bounds__(512) void k(unsigned* buf, unsigned words, int per_thread, int mode, unsigned copies)
{
    unsigned x = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    unsigned xcc = 0;
    if (copies > 1) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
    }
    unsigned* base = buf + (size_t)(xcc % copies) * words;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        unsigned a = mode == 0 ? (x >> 8) % words            // scattered over the buffer
                   : mode == 1 ? ((x >> 8) % 900u) * 64u     // 900 hot addresses
                               : (threadIdx.x & 63) + 64u * ((x >> 8) % (words / 64u));  // a wave hits one 256 B run
        __hip_atomic_fetch_add(base + a, 1u, __ATOMIC_RELAXED, SCOPE);
    }
}

int main()
{
    const unsigned words = 2843 * 70;
    unsigned* d;
    hipMalloc(&d, (size_t)words * 8 * 4);
    hipMemset(d, 0, (size_t)words * 8 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int per = 256;
    const char* mnames[] = {"scattered", "900 hot addresses", "coalesced runs"};
    for (int mode = 0; mode < 3; ++mode)
        for (int variant = 0; variant < 3; ++variant) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(256), dim3(512), 0, 0, d, words, per, mode, 1u);
                if (variant == 1) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(256), dim3(512), 0, 0, d, words, per, mode, 1u);
                if (variant == 2) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(256), dim3(512), 0, 0, d, words, per, mode, 8u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = 256.0 * 512 * per;
            printf("%-20s %-34s %8.3f ms  %7.1f G atomics/s\n", mnames[mode],
                   variant == 0 ? "agent scope" : variant == 1 ? "workgroup scope (one copy: WRONG)" : "workgroup scope, copy per XCD", ms, n / ms / 1e6);
        }
    // correctness of the per-XCD copies: the copies must add up to the number of atomics issued
    hipMemset(d, 0, (size_t)words * 8 * 4);
    hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(256), dim3(512), 0, 0, d, words, per, 1, 8u);
    hipDeviceSynchronize();
    unsigned* h = new unsigned[(size_t)words * 8];
    hipMemcpy(h, d, (size_t)words * 8 * 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 0, per_copy[8] = {0};
    for (unsigned c = 0; c < 8; ++c)
        for (unsigned i = 0; i < words; ++i) {
            sum += h[(size_t)c * words + i];
            per_copy[c] += h[(size_t)c * words + i];
        }
    printf("per-XCD copies sum to %llu of %llu issued; per copy:", sum, 256ull * 512 * per);
    for (int c = 0; c < 8; ++c) printf(" %llu", per_copy[c]);
    printf("\n");
    return 0;
}
