// How many cycles does a SIMD of gfx950 need per 64-lane VALU instruction, by instruction class?  (What "VALU
// bound" means for the render kernel: its roofline.)  Every wave runs ITER x 16 instructions of one kind on 8
// independent register chains (or ONE dependent chain), W waves per SIMD on every SIMD of the chip.
// Output: one JSON line per (instruction, chains, W): shader-clock cycles (s_memtime ticks of wave 0) per instruction
// of one wave, and per wave-instruction of the SIMD (= that / W).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define KINDS(X)                                                                  \
    X(0, "v_add_f32", "v_add_f32 %0, %0, %1", a)                                 \
    X(1, "v_mul_f32", "v_mul_f32 %0, %0, %1", a)                                 \
    X(2, "v_sub_f32", "v_sub_f32 %0, %1, %0", a)                                 \
    X(3, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2", a)                             \
    X(4, "v_max_f32", "v_max_f32 %0, %0, %1", a)                                 \
    X(5, "v_min3_f32", "v_min3_f32 %0, %0, %1, %2", a)                           \
    X(6, "v_cndmask_b32 (sgpr cond)", "v_cndmask_b32 %0, %0, %1, s[20:21]", a)  \
    X(7, "v_cmp_lt_f32 (to sgpr pair)", "v_cmp_lt_f32 s[22:23], %0, %1", a)     \
    X(8, "v_add_u32", "v_add_u32 %0, %0, %1", u)                                 \
    X(9, "v_xor_b32", "v_xor_b32 %0, %0, %1", u)                                 \
    X(10, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1", u)                   \
    X(11, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", u)                          \
    X(12, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", u)                            \
    X(13, "v_rcp_f32", "v_rcp_f32 %0, %0", a)                                    \
    X(14, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2", p)                      \
    X(15, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", p)                          \
    X(16, "v_mov_b32", "v_mov_b32 %0, %1", a)                                    \
    X(17, "v_fma_f64", "v_fma_f64 %0, %0, %1, %2", d)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int kKind, bool kDependent>
__global__ void __launch_bounds__(256) rate_kernel(float *out, unsigned long long *ticks, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.0f, p5 = p1 + 1.0f, p6 = p2 + 1.0f, p7 = p3 + 1.0f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float ab = 1.0001f, ac = 0.5f;
    const unsigned ub = 3, uc = 5;
    const float2v pb = {ab, ab}, pc = {ac, ac};
    const double db = 1.0001, dc = 0.5;
    asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555" ::: "s20", "s21");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
    {
#define ONE(T, X, TEXT) asm volatile(TEXT : "+v"(T##X) : "v"(T##b), "v"(T##c) : "s22", "s23");
#define X(ID, NAME, TEXT, T)                                                                              \
    if (kKind == ID)                                                                                      \
    {                                                                                                     \
        if (kDependent)                                                                                   \
        {                                                                                                 \
            ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) \
            ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) ONE(T, 0, TEXT) \
        }                                                                                                 \
        else                                                                                              \
        {                                                                                                 \
            ONE(T, 0, TEXT) ONE(T, 1, TEXT) ONE(T, 2, TEXT) ONE(T, 3, TEXT) ONE(T, 4, TEXT) ONE(T, 5, TEXT) ONE(T, 6, TEXT) ONE(T, 7, TEXT) \
            ONE(T, 0, TEXT) ONE(T, 1, TEXT) ONE(T, 2, TEXT) ONE(T, 3, TEXT) ONE(T, 4, TEXT) ONE(T, 5, TEXT) ONE(T, 6, TEXT) ONE(T, 7, TEXT) \
        }                                                                                                 \
    }
        KINDS(X)
#undef X
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
                                                 float(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) + float(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *ticks = t1 - t0;
}

template <int kKind, bool kDependent>
void Run(const char *name, int cus, int waves_per_simd)
{
    const int iters = 8000, blocks = cus * waves_per_simd; // 256 lanes = 4 waves = one per SIMD
    float *out;
    unsigned long long *ticks, host_ticks = 0;
    (void)hipMalloc(&out, size_t(blocks) * 256 * 4);
    (void)hipMalloc(&ticks, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<kKind, kDependent>), dim3(blocks), dim3(256), 0, 0, out, ticks, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<kKind, kDependent>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&host_ticks, ticks, 8, hipMemcpyDeviceToHost);
    const double insts_per_wave = double(iters) * 16;
    const double clock_ghz = double(host_ticks) / (ms * 1e6); // wave 0 runs (nearly) the whole launch
    printf("{\"inst\": \"%s\", \"chains\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_inst_one_wave\": %.2f, "
           "\"cycles_per_wave_inst_per_simd\": %.2f, \"clock_ghz_from_ticks\": %.2f}\n",
           name, kDependent ? "1 dependent" : "8 independent", waves_per_simd, ms, double(host_ticks) / insts_per_wave,
           double(host_ticks) / (insts_per_wave * waves_per_simd), clock_ghz);
    (void)hipFree(out), (void)hipFree(ticks);
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.gcnArchName, cus);
#define X(ID, NAME, TEXT, T)            \
    Run<ID, false>(NAME, cus, 1);       \
    Run<ID, false>(NAME, cus, 4);       \
    Run<ID, false>(NAME, cus, 8);       \
    Run<ID, true>(NAME, cus, 1);        \
    Run<ID, true>(NAME, cus, 4);
    KINDS(X)
#undef X
    return 0;
}
