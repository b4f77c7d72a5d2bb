// Issue cost of the VALU / LDS instructions the pileup kernel leans on, gfx950: cycles per instruction for one wave
// per SIMD and for two (4 and 8 waves in one workgroup on one CU).  hipcc --offload-arch=gfx950 -O2 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, ASM)                                                                                         \
    __global__ void k_##name(unsigned long long* out, int iters)                                                \
    {                                                                                                           \
        __shared__ unsigned long long lds[1024];                                                                \
        unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = 1, f = 2, g = 3, h = 4;          \
        unsigned long long p = 0, q = 0;                                                                        \
        unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;                                    \
        (void)lds; (void)p; (void)q; (void)addr;                                                                \
        __syncthreads();                                                                                        \
        unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < iters; ++i) {                                                                       \
            REP16(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(addr) : "memory");) \
        }                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                       \
        if (a + b + c + d + e + f + g + h == 0x12345) out[0] = a;                                               \
    }

// four independent instructions per asm statement (no dependency stalls)
BODY(and, "v_and_b32 %0, %1, %0\n v_and_b32 %2, %3, %2\n v_and_b32 %4, %5, %4\n v_and_b32 %6, %7, %6")
BODY(bitop3, "v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96\n v_bitop3_b32 %2, %3, %2, %4 bitop3:0xe8\n v_bitop3_b32 %4, %5, %4, %6 bitop3:0x96\n v_bitop3_b32 %6, %7, %6, %0 bitop3:0xe8")
BODY(alignbit, "v_alignbit_b32 %0, %1, %0, 3\n v_alignbit_b32 %2, %3, %2, 5\n v_alignbit_b32 %4, %5, %4, 7\n v_alignbit_b32 %6, %7, %6, 9")
BODY(lshl, "v_lshlrev_b32 %0, %1, %0\n v_lshlrev_b32 %2, %3, %2\n v_lshlrev_b32 %4, %5, %4\n v_lshlrev_b32 %6, %7, %6")
BODY(bcnt, "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %4, %5, %4\n v_bcnt_u32_b32 %6, %7, %6")
BODY(cndmask, "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n v_cndmask_b32 %4, %5, %4, vcc\n v_cndmask_b32 %6, %7, %6, vcc")
BODY(add3, "v_add3_u32 %0, %1, %0, %2\n v_add3_u32 %2, %3, %2, %4\n v_add3_u32 %4, %5, %4, %6\n v_add3_u32 %6, %7, %6, %0")
BODY(dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %5, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %7, %6 row_shr:8 row_mask:0xf bank_mask:0xf")
BODY(dep_and, "v_and_b32 %0, %1, %0\n v_and_b32 %0, %2, %0\n v_and_b32 %0, %3, %0\n v_and_b32 %0, %4, %0")
BODY(readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9")
BODY(salu, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1")
BODY(mix_vs, "v_and_b32 %0, %1, %0\n s_add_u32 s20, s20, 1\n v_and_b32 %2, %3, %2\n s_add_u32 s21, s21, 1")
BODY(ds_or32, "ds_or_b32 %8, %0 offset:0\n ds_or_b32 %8, %2 offset:8\n ds_or_b32 %8, %4 offset:4096\n ds_or_b32 %8, %6 offset:4104")
BODY(ds_read32, "ds_read_b32 %0, %8 offset:0\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %6, %8 offset:4104")

BODY(cnd_sgpr, "v_cndmask_b32_e64 %0, %1, %0, s[20:21]\n v_cndmask_b32_e64 %2, %3, %2, s[20:21]\n v_cndmask_b32_e64 %4, %5, %4, s[22:23]\n v_cndmask_b32_e64 %6, %7, %6, s[22:23]")
BODY(and_sgpr, "v_and_b32 %0, s20, %0\n v_and_b32 %2, s21, %2\n v_and_b32 %4, s22, %4\n v_and_b32 %6, s23, %6")
BODY(lshl_sgpr, "v_lshlrev_b32 %0, s20, %0\n v_lshlrev_b32 %2, s21, %2\n v_lshlrev_b32 %4, s22, %4\n v_lshlrev_b32 %6, s23, %6")
BODY(cmp_vcc, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %6, %7")
BODY(cmp_sgpr, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %2, %3\n v_cmp_lt_u32_e64 s[24:25], %4, %5\n v_cmp_lt_u32_e64 s[26:27], %6, %7")
BODY(mov_sgpr, "v_mov_b32 %0, s20\n v_mov_b32 %2, s21\n v_mov_b32 %4, s22\n v_mov_b32 %6, s23")
BODY(rfl, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3")
BODY(and_const, "v_and_b32 %0, 0x12345678, %0\n v_and_b32 %2, 0x12345678, %2\n v_and_b32 %4, 0x12345678, %4\n v_and_b32 %6, 0x12345678, %6")
BODY(and_inline, "v_and_b32 %0, 31, %0\n v_and_b32 %2, 31, %2\n v_and_b32 %4, 31, %4\n v_and_b32 %6, 31, %6")
BODY(cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %3, %2, vcc\n v_and_b32 %4, %5, %4\n v_and_b32 %6, %7, %6")
BODY(mbcnt, "v_mbcnt_lo_u32_b32 %0, s20, %0\n v_mbcnt_hi_u32_b32 %2, s21, %2\n v_mbcnt_lo_u32_b32 %4, s22, %4\n v_mbcnt_hi_u32_b32 %6, s23, %6")
BODY(exec_mask, "s_mov_b64 s[20:21], exec\n v_and_b32 %0, %1, %0\n s_mov_b64 exec, s[20:21]\n v_and_b32 %2, %3, %2")

typedef void (*kern_t)(unsigned long long*, int);
struct Entry { const char* name; kern_t k; int per; };

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 4096 * 8);
    std::vector<unsigned long long> h(4096);
    const int iters = 2000;
    Entry es[] = {{"v_and_b32 (indep)", k_and, 64}, {"v_and_b32 (dependent)", k_dep_and, 64}, {"v_bitop3_b32", k_bitop3, 64},
                  {"v_alignbit_b32", k_alignbit, 64}, {"v_lshlrev_b32", k_lshl, 64}, {"v_bcnt_u32_b32", k_bcnt, 64},
                  {"v_cndmask_b32", k_cndmask, 64}, {"v_add3_u32", k_add3, 64}, {"v_add_u32_dpp", k_dpp, 64},
                  {"v_readlane_b32", k_readlane, 64}, {"s_add_u32", k_salu, 64}, {"v_and + s_add mix", k_mix_vs, 64},
                  {"v_cndmask (sgpr pair mask)", k_cnd_sgpr, 64}, {"v_and_b32 (sgpr operand)", k_and_sgpr, 64},
                  {"v_lshlrev_b32 (sgpr operand)", k_lshl_sgpr, 64}, {"v_cmp -> vcc", k_cmp_vcc, 64}, {"v_cmp -> sgpr pair", k_cmp_sgpr, 64},
                  {"v_mov_b32 v, s", k_mov_sgpr, 64}, {"v_readfirstlane", k_rfl, 64}, {"v_and_b32 (literal)", k_and_const, 64},
                  {"v_and_b32 (inline const)", k_and_inline, 64}, {"cmp+cndmask+2 and", k_cmp_cnd, 64}, {"v_mbcnt (sgpr operand)", k_mbcnt, 64},
                  {"exec save/restore + 2 and", k_exec_mask, 64},
                  {"ds_or_b32 (16B lane stride)", k_ds_or32, 64}, {"ds_read_b32 (16B lane stride)", k_ds_read32, 64}};
    for (auto& e : es) {
        printf("%-30s", e.name);
        for (int waves : {1, 4, 8, 12, 16}) {
            hipLaunchKernelGGL(e.k, dim3(1), dim3(64 * waves), 0, 0, d, iters);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int w = 0; w < waves; ++w) worst = h[w] > worst ? (double)h[w] : worst;
            printf("  %2dw: %6.2f", waves, worst / ((double)iters * e.per));
        }
        printf("   cycles/instr/wave\n");
    }
    return 0;
}
