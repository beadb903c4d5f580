// What bounds a dependent chain of 64-byte record fetches from a table that does not fit L2 (a hierarchy walk on a
// mesh)?  Every lane follows its own pseudo-random chain through a table of 64-byte records; per step it loads
//   mode 0: the whole record with 4 x global_load_dwordx4 (what the walk does),
//   mode 1: 16 bytes of it with 1 x global_load_dwordx4,
//   mode 2: the whole record, but the 4 lanes of a quad load ONE record together (16 bytes each, one instruction per
//           record of the quad: 4 instructions for 4 records, every instruction touches 16 lines instead of 64).
// If mode 1 / 2 are much faster than mode 0 the L1 / address path (line requests per instruction) is the limit, if not
// it is the latency of the chain.   Output: M steps / s per mode and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/node_fetch.hip -o tools/microbench/node_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int kMode>
__global__ void __launch_bounds__(256) chase(const uint4 *__restrict__ table, uint32_t n_records, uint32_t steps, uint32_t *out)
{
    uint32_t cur = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % n_records;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s)
    {
        const uint4 *rec = table + 4 * static_cast<size_t>(cur);
        if (kMode == 0)
        {
            const uint4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
            acc += a.y ^ b.z ^ c.w ^ d.y;
            cur = (a.x + (b.x & 1u) + (c.x & 1u) + (d.x & 1u)) % n_records;
        }
        else if (kMode == 1)
        {
            const uint4 a = rec[0];
            acc += a.y;
            cur = a.x % n_records;
        }
        else
        {
            // quad-cooperative: lane q of a quad loads piece q of the record of quad-lane j, for j = 0..3
            const uint32_t q = threadIdx.x & 3u;
            uint32_t next = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const uint32_t cur_j = __shfl(cur, (threadIdx.x & ~3u) | j, 64);
                const uint4 piece = table[4 * static_cast<size_t>(cur_j) + q];
                // lane j needs piece 0's x (the link) and something of every piece: gather with quad shuffles
                const uint32_t link = __shfl(piece.x, (threadIdx.x & ~3u), 64);                 // piece 0 sits in quad-lane 0
                const uint32_t mix = piece.y ^ __shfl_xor(piece.y, 1, 64) ^ __shfl_xor(piece.y, 2, 64);
                if (static_cast<uint32_t>(j) == q)
                    next = link, acc += mix;
            }
            cur = next % n_records;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + cur;
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const uint32_t n_records = 1500000; // 96 MB: three times the L2
    std::vector<uint4> host(4 * size_t(n_records));
    uint32_t state = 12345;
    for (size_t i = 0; i < host.size(); ++i)
    {
        state = state * 1664525u + 1013904223u;
        host[i] = uint4{state >> 3, state, state * 3u, state * 7u};
    }
    uint4 *table;
    uint32_t *out;
    (void)hipMalloc(&table, host.size() * sizeof(uint4));
    (void)hipMemcpy(table, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice);
    const uint32_t steps = 2000;
    for (int waves : {2, 4, 8})
    {
        const uint32_t blocks = prop.multiProcessorCount * waves; // 4 waves per block = `waves` per SIMD
        (void)hipMalloc(&out, size_t(blocks) * 256 * 4);
        for (int mode = 0; mode < 3; ++mode)
        {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
            auto launch = [&](uint32_t n)
            {
                if (mode == 0)
                    hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(256), 0, 0, table, n_records, n, out);
                else if (mode == 1)
                    hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, 0, table, n_records, n, out);
                else
                    hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(256), 0, 0, table, n_records, n, out);
            };
            launch(50);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            launch(steps);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double lane_steps = double(blocks) * 256 * steps;
            printf("{\"waves_per_simd\": %d, \"mode\": %d, \"ms\": %.3f, \"G_lane_steps_per_s\": %.2f, \"ns_per_step_per_wave\": %.1f}\n", waves, mode,
                   ms, lane_steps / (ms * 1e6), ms * 1e6 / steps);
        }
        (void)hipFree(out);
    }
    return 0;
}
