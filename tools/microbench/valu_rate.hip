// How many cycles does a SIMD of gfx950 need per 64-lane fp32 VALU instruction?  (The roofline's VALU peak.)
// Every wave runs ITER x 16 independent-chain instructions (8 chains, unrolled twice) of one kind:
//   0: v_fma_f32   1: v_add_f32   2: v_mul_f32   3: v_pk_fma_f32   4: v_cndmask_b32   5: v_max_f32
// launched with W waves per SIMD on every SIMD.  Prints wave-instructions per SIMD-cycle from the wall time of the
// launch (HIP events) and the shader clock (s_memtime ticks of one wave).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int kKind>
__global__ void __launch_bounds__(256) rate_kernel(float *out, unsigned long long *ticks, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.0f, p5 = p1 + 1.0f, p6 = p2 + 1.0f, p7 = p3 + 1.0f;
    float2v pb = {b, b}, pc = {c, c};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
    {
#define STEP(X)                                                                                        \
    if (kKind == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##X) : "v"(b), "v"(c));            \
    if (kKind == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##X) : "v"(c));                        \
    if (kKind == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##X) : "v"(b));                        \
    if (kKind == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##X) : "v"(pb), "v"(pc));       \
    if (kKind == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##X) : "v"(c));               \
    if (kKind == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##X) : "v"(c));
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *ticks = t1 - t0;
}

template <int kKind>
void Run(const char *name, int cus, int waves_per_simd)
{
    const int iters = 20000, blocks = cus * waves_per_simd; // 256 lanes = 4 waves = one per SIMD
    float *out;
    unsigned long long *ticks, host_ticks = 0;
    hipMalloc(&out, size_t(blocks) * 256 * 4);
    hipMalloc(&ticks, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<kKind>, dim3(blocks), dim3(256), 0, 0, out, ticks, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<kKind>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&host_ticks, ticks, 8, hipMemcpyDeviceToHost);
    const double insts_per_wave = double(iters) * 16;
    // s_memtime / readcyclecounter runs at a fixed 100 MHz on gfx9: use wall time and the nominal clock as well
    const double wave_insts_per_simd = insts_per_wave * waves_per_simd;
    printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_simd_per_us\": %.1f, "
           "\"cycles_per_wave_inst_at_2400MHz\": %.3f, \"counter_ticks\": %llu}\n",
           name, waves_per_simd, ms, wave_insts_per_simd / (ms * 1e3), ms * 1e-3 * 2.4e9 / wave_insts_per_simd, host_ticks);
    hipFree(out), hipFree(ticks);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.gcnArchName, cus, prop.clockRate);
    for (int w : {1, 2, 4, 8})
    {
        Run<0>("v_fma_f32", cus, w);
        Run<1>("v_add_f32", cus, w);
        Run<2>("v_mul_f32", cus, w);
        Run<3>("v_pk_fma_f32", cus, w);
        Run<4>("v_cndmask_b32", cus, w);
        Run<5>("v_max_f32", cus, w);
    }
    return 0;
}
