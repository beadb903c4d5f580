// TEST INFRASTRUCTURE — csrc/glibc_libm.h ON THE DEVICE against the host's libm, all 2^32 arguments.
//
// tests/libm/libm_check.cpp sweeps the HOST build of the header; "device == host" was inferred from frames.  This
// program evaluates the same header's functions in a gfx950 kernel for every float bit pattern (sinf cosf tanf acosf
// atanf; the medium code's DOUBLE exp and log with three double arguments per pattern) and for 2^26 (y, x) pairs of atan2f, and compares with what the host's libm (::sinf ...) returns for the same
// arguments — not argument by argument over PCIe but chunk by chunk: both sides fold a chunk of 65 536 consecutive
// arguments into sum_i canon(bits(f(x_i))) * (2 i + 1) mod 2^64 (order independent, position sensitive; every NaN
// counts as 0x7fc00000).  A differing chunk is then resolved on the host argument by argument (device values of that
// chunk copied back).  Prints one JSON line; exit code 1 when anything differs.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I<csrc> libm_device_sweep.hip -o libm_device_sweep -lpthread
//   ./libm_device_sweep [stride_of_chunks]      (1 = all 65 536 chunks)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "glibc_libm.h"

namespace gl = mcpt::gl;

constexpr unsigned kChunkBits = 16, kChunk = 1u << kChunkBits, kChunks = 1u << (32 - kChunkBits);
constexpr int kFunctions = 8; // sinf cosf tanf acosf atanf atan2f exp log (the last two in double)

__host__ __device__ inline unsigned canon(float v)
{
    const unsigned b = gl::bits(v);
    return (b & 0x7fffffffu) > 0x7f800000u ? 0x7fc00000u : b;
}

// the second argument of atan2f for the i-th y of the sweep: bit patterns spread over signs / exponents / mantissas
__host__ __device__ inline unsigned atan2_x(unsigned u)
{
    unsigned h = u * 2654435761u + 0x9e3779b9u;
    h ^= h >> 15, h *= 0x85ebca6bu, h ^= h >> 13;
    switch (u & 3u)
    {
    case 0: return h;                                   // anything
    case 1: return (h & 0x807fffffu) | (u & 0x7f800000u); // y's exponent
    case 2: return u ^ (h & 0x3fffffu);                 // close to y
    default: return h & 0xff800000u;                    // powers of two, zeros, infinities
    }
}

__host__ __device__ inline unsigned long long canon64(double v)
{
    const unsigned long long b = gl::bits64(v);
    return (b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull ? 0x7ff8000000000000ull : b;
}

// exp / log (double): per float bit pattern u three arguments — the float promoted to double (what the renderer
// passes), the same with a hashed low mantissa half, and a hashed 64-bit pattern (tests/libm/libm_check.cpp sweeps
// the same three on the host build)
__host__ __device__ inline void double_arguments(unsigned u, double x[3])
{
    x[0] = static_cast<double>(gl::from_bits(u));
    unsigned long long h = (static_cast<unsigned long long>(u) + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
    h ^= h >> 31, h *= 0x94d049bb133111ebull, h ^= h >> 29;
    x[1] = gl::from_bits64(gl::bits64(x[0]) ^ (h & 0x1fffffffull));
    x[2] = gl::from_bits64(h);
}

// the word a function contributes for argument index u (float functions: the canonical result bits)
template <int kF>
__host__ __device__ inline unsigned long long device_side(unsigned u)
{
    const float x = gl::from_bits(u);
    switch (kF)
    {
    case 0: return canon(gl::sinf(x));
    case 1: return canon(gl::cosf(x));
    case 2: return canon(gl::tanf(x));
    case 3: return canon(gl::acosf(x));
    case 4: return canon(gl::atanf(x));
    case 5: return canon(gl::atan2f(x, gl::from_bits(atan2_x(u))));
    default:
    {
        double a[3];
        double_arguments(u, a);
        if (kF == 6)
            return canon64(gl::exp(a[0])) + 3ull * canon64(gl::exp(a[1])) + 5ull * canon64(gl::exp(a[2]));
        return canon64(gl::log(a[0])) + 3ull * canon64(gl::log(a[1])) + 5ull * canon64(gl::log(a[2]));
    }
    }
}

static unsigned long long host_libm(int f, unsigned u)
{
    const float x = gl::from_bits(u);
    switch (f)
    {
    case 0: return canon(::sinf(x));
    case 1: return canon(::cosf(x));
    case 2: return canon(::tanf(x));
    case 3: return canon(::acosf(x));
    case 4: return canon(::atanf(x));
    case 5: return canon(::atan2f(x, gl::from_bits(atan2_x(u))));
    default:
    {
        double a[3];
        double_arguments(u, a);
        if (f == 6)
            return canon64(::exp(a[0])) + 3ull * canon64(::exp(a[1])) + 5ull * canon64(::exp(a[2]));
        return canon64(::log(a[0])) + 3ull * canon64(::log(a[1])) + 5ull * canon64(::log(a[2]));
    }
    }
}

template <int kF>
__global__ void __launch_bounds__(256) sweep(unsigned long long *sums, unsigned chunk_stride, unsigned chunk_limit)
{
    // one workgroup per chunk
    const unsigned chunk = blockIdx.x * chunk_stride;
    if (chunk >= chunk_limit)
        return;
    unsigned long long acc = 0;
    for (unsigned i = threadIdx.x; i < kChunk; i += 256)
        acc += device_side<kF>(chunk * kChunk + i) * (2ull * i + 1ull);
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_xor(acc, off, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

template <int kF>
__global__ void values(unsigned long long *out, unsigned chunk)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kChunk)
        out[i] = device_side<kF>(chunk * kChunk + i);
}

#define CHECK(x)                                                                                  \
    do                                                                                            \
    {                                                                                             \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess)                                                                     \
        {                                                                                         \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                             \
        }                                                                                         \
    } while (0)

int main(int argc, char **argv)
{
    const unsigned stride = argc > 1 ? static_cast<unsigned>(strtoul(argv[1], nullptr, 10)) : 1u;
    // atan2f: 2^26 pairs (every 64th chunk of y) — the pairs libm_check.cpp sweeps on the host are 3 x 2^32 / stride
    const unsigned n_blocks = (kChunks + stride - 1) / stride;
    unsigned long long *d_sums = nullptr;
    unsigned long long *d_vals = nullptr;
    CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * n_blocks));
    CHECK(hipMalloc(&d_vals, sizeof(unsigned long long) * kChunk));
    std::vector<unsigned long long> dev(n_blocks), host(n_blocks);
    const char *names[kFunctions] = {"sinf", "cosf", "tanf", "acosf", "atanf", "atan2f", "exp", "log"};
    const unsigned n_threads = std::max(1u, std::thread::hardware_concurrency());
    std::string json = "{";
    long long total_bad = 0;
    float gpu_ms_total = 0;
    for (int f = 0; f < kFunctions; ++f)
    {
        const unsigned fstride = f == 5 ? stride * 64u : stride, blocks = (kChunks + fstride - 1) / fstride;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, nullptr));
        switch (f)
        {
        case 0: hipLaunchKernelGGL(sweep<0>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        case 1: hipLaunchKernelGGL(sweep<1>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        case 2: hipLaunchKernelGGL(sweep<2>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        case 3: hipLaunchKernelGGL(sweep<3>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        case 4: hipLaunchKernelGGL(sweep<4>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        case 5: hipLaunchKernelGGL(sweep<5>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        case 6: hipLaunchKernelGGL(sweep<6>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        default: hipLaunchKernelGGL(sweep<7>, dim3(blocks), dim3(256), 0, nullptr, d_sums, fstride, kChunks); break;
        }
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, nullptr));
        // the host's libm on the same chunks while the kernel runs
        std::atomic<unsigned> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t)
            pool.emplace_back([&]() {
                for (;;)
                {
                    const unsigned b = next.fetch_add(1);
                    if (b >= blocks)
                        break;
                    const unsigned chunk = b * fstride;
                    unsigned long long acc = 0;
                    for (unsigned i = 0; i < kChunk; ++i)
                        acc += host_libm(f, chunk * kChunk + i) * (2ull * i + 1ull);
                    host[b] = acc;
                }
            });
        for (auto &th : pool)
            th.join();
        CHECK(hipMemcpy(dev.data(), d_sums, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        gpu_ms_total += ms;
        long long bad_chunks = 0, bad_args = 0;
        unsigned first_arg = 0;
        bool have_first = false;
        for (unsigned b = 0; b < blocks; ++b)
            if (dev[b] != host[b])
            {
                ++bad_chunks;
                if (bad_chunks > 8)
                    continue; // (resolve the first few chunks only)
                const unsigned chunk = b * fstride;
                switch (f)
                {
                case 0: hipLaunchKernelGGL(values<0>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                case 1: hipLaunchKernelGGL(values<1>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                case 2: hipLaunchKernelGGL(values<2>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                case 3: hipLaunchKernelGGL(values<3>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                case 4: hipLaunchKernelGGL(values<4>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                case 5: hipLaunchKernelGGL(values<5>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                case 6: hipLaunchKernelGGL(values<6>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                default: hipLaunchKernelGGL(values<7>, dim3(kChunk / 256), dim3(256), 0, nullptr, d_vals, chunk); break;
                }
                std::vector<unsigned long long> vals(kChunk);
                CHECK(hipMemcpy(vals.data(), d_vals, sizeof(unsigned long long) * kChunk, hipMemcpyDeviceToHost));
                for (unsigned i = 0; i < kChunk; ++i)
                    if (vals[i] != host_libm(f, chunk * kChunk + i))
                    {
                        ++bad_args;
                        if (!have_first)
                            first_arg = chunk * kChunk + i, have_first = true;
                    }
            }
        total_bad += bad_chunks;
        char rec[256];
        snprintf(rec, sizeof rec, "%s\"%s\": {\"arguments\": %llu, \"chunks\": %u, \"differing_chunks\": %lld, \"differing_arguments_in_first_chunks\": %lld, \"first\": \"0x%08x\", \"gpu_ms\": %.3f}",
                 f ? ", " : "", names[f], static_cast<unsigned long long>(blocks) * kChunk, blocks, bad_chunks, bad_args, first_arg, ms);
        json += rec;
    }
    char tail[128];
    snprintf(tail, sizeof tail, ", \"gpu_ms_total\": %.3f, \"host_threads\": %u, \"differing_chunks_total\": %lld}", gpu_ms_total, n_threads, total_bad);
    json += tail;
    puts(json.c_str());
    return total_bad == 0 ? 0 : 1;
}
