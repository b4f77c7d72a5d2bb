// How fast does a mapped file reach the GPU?  (a) hipMemcpyAsync from the pageable mapping (what msq_bgzf_stage_file does),
// (b) T threads copy slices of the mapping into pinned slots, each slot then crosses by DMA.
// Build here, run on the GPU box: hipcc -O2 -pthread --offload-arch=gfx950 tools/probe/h2d_probe.cpp -o tools/probe/h2d_probe && tools/probe/h2d_probe file [threads] [slot MB] [slots]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* path = argv[1];
    const int threads = argc > 2 ? atoi(argv[2]) : 4;
    const size_t slot = (argc > 3 ? atoi(argv[3]) : 8) << 20;
    const int n_slots = argc > 4 ? atoi(argv[4]) : 8;
    int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const size_t n = st.st_size;
    char* d;
    CK(hipMalloc(&d, n + 4096));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char* pinned;
    double t0 = now();
    CK(hipHostMalloc(&pinned, slot * n_slots, hipHostMallocDefault));
    std::printf("hipHostMalloc %zu MB: %.2f ms\n", slot * n_slots >> 20, now() - t0);
    std::vector<hipEvent_t> ev(n_slots);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int rep = 0; rep < 3; ++rep) {
        const char* m = (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        t0 = now();
        CK(hipMemcpyAsync(d, m, n, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t = now() - t0;
        std::printf("pageable: %.2f ms  %.1f GB/s\n", t, n / t / 1e6);
        munmap((void*)m, n);
        m = (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        t0 = now();
        const size_t n_chunks = (n + slot - 1) / slot;
        // thread t copies the t-th part of every chunk; chunk c is ready when all threads have done their part, and may be
        // filled once chunk c - n_slots has crossed (fill_limit)
        std::vector<std::atomic<int>> ready(n_chunks);
        for (auto& r : ready) r.store(0);
        std::atomic<size_t> fill_limit{(size_t)n_slots};
        std::vector<std::thread> pool;
        for (int t2 = 0; t2 < threads; ++t2)
            pool.emplace_back([&, t2] {
                for (size_t c = 0; c < n_chunks; ++c) {
                    while (c >= fill_limit.load(std::memory_order_acquire)) std::this_thread::yield();
                    const size_t lo = c * slot, len = std::min(slot, n - lo);
                    const size_t a = len * t2 / threads, b = len * (t2 + 1) / threads;
                    std::memcpy(pinned + (c % n_slots) * slot + a, m + lo + a, b - a);
                    ready[c].fetch_add(1, std::memory_order_release);
                }
            });
        size_t oldest = 0, n_issued = 0;
        auto poll = [&] {
            while (oldest < n_issued && hipEventQuery(ev[oldest % n_slots]) == hipSuccess) {
                ++oldest;
                fill_limit.store(oldest + n_slots, std::memory_order_release);
            }
        };
        for (size_t c = 0; c < n_chunks; ++c) {
            while (ready[c].load(std::memory_order_acquire) < threads) {
                poll();
                std::this_thread::yield();
            }
            const size_t lo = c * slot, len = std::min(slot, n - lo);
            CK(hipMemcpyAsync(d + lo, pinned + (c % n_slots) * slot, len, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[c % n_slots], s));
            n_issued = c + 1;
            poll();
        }
        for (auto& th : pool) th.join();
        CK(hipStreamSynchronize(s));
        t = now() - t0;
        std::printf("threads %d slot %zu MB x %d: %.2f ms  %.1f GB/s\n", threads, slot >> 20, n_slots, t, n / t / 1e6);
        munmap((void*)m, n);
    }
    return 0;
}
