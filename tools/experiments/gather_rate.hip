// MEASUREMENT TOOL (not part of the library): how fast can a CU gather 128-byte records from a table that lives in L2 / the
// memory-side cache, one dependent record per lane and step — the node step of the pool walk outside LDS (csrc/pool_walk.h) —
//   scattered:    every lane reads the 7 x 16 bytes of ITS record (7 load instructions, each touching 64 different lines)
//   cooperative:  the 8 lanes of a group read the 8 x 16 bytes of one record per instruction (8 instructions, each
//                 touching 8 lines), 8 records per group — the data then sits transposed in the group
// hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip ; ./gather_rate [records] [steps] [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t v)
{
    v ^= v >> 16, v *= 0x7feb352du, v ^= v >> 15, v *= 0x846ca68bu, v ^= v >> 16;
    return v;
}

template <int kMode>
__global__ void __launch_bounds__(256) gather(const float4 *__restrict__ table, uint32_t n_records, uint32_t steps, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = mix(tid) % n_records;
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s)
    {
        float sum = 0.0f;
        if (kMode == 0)
        {
            const float4 *r = table + 8u * static_cast<size_t>(idx);
#pragma unroll
            for (int c = 0; c < 7; ++c)
            {
                const float4 v = r[c];
                sum += v.x + v.y + v.z + v.w;
            }
        }
        else
        {
            const uint32_t j = lane & 7u, g8 = lane & ~7u;
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                const uint32_t other = __shfl(idx, g8 + i);       // the record of lane 8g + i
                const float4 v = table[8u * static_cast<size_t>(other) + (j ^ (kMode == 2 ? i : 0))];
                sum += v.x + v.y + v.z + v.w;                      // (lane 8g + j now holds chunk j of 8 records)
            }
        }
        acc += sum;
        idx = mix(idx ^ __float_as_uint(sum) ^ s) % n_records;    // the next record depends on this one
    }
    if (acc == 12345.678f)
        out[0] = idx;
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? std::atoi(argv[1]) : 300000u, steps = argc > 2 ? std::atoi(argv[2]) : 2000u, waves = argc > 3 ? std::atoi(argv[3]) : 3u;
    float4 *table;
    uint32_t *out;
    hipMalloc(&table, size_t(n) * 128);
    hipMalloc(&out, 4);
    std::vector<float> host(size_t(n) * 32);
    for (size_t i = 0; i < host.size(); ++i)
        host[i] = float(i % 977) * 1e-3f;
    hipMemcpy(table, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const uint32_t blocks = prop.multiProcessorCount * waves; // 256 threads = 4 wavefronts per block: `waves` per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep)
        {
            hipEventRecord(a);
            if (mode == 0)
                hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, table, n, steps, out);
            else if (mode == 1)
                hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, table, n, steps, out);
            else
                hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, table, n, steps, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            const double records = double(blocks) * 256.0 * steps;
            std::printf("{\"mode\": \"%s\", \"records\": %u, \"table_MB\": %.1f, \"waves_per_simd\": %u, \"ms\": %.3f, \"G_records_per_s\": %.2f, \"ns_per_step\": %.1f}\n",
                        mode == 0 ? "scattered" : mode == 1 ? "cooperative" : "cooperative, swizzled chunks", n, n * 128.0 / 1e6, waves, ms, records / ms / 1e6,
                        ms * 1e6 / steps);
        }
    return 0;
}
