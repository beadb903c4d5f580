// libmcpt_host.so (include/mcpt_host.h): the kernel body of the render path on host threads, behind
// `mcpt_cli --cpu`.  Same commit (commit.cpp) and same per-pixel functions (path_core.h) as the HIP kernel, so the
// frame is the GPU's frame bit for bit; the instantiation is picked the way the HIP launcher picks it
// (hip/render_kernel.hip).  Replaces the reference's CPU dispatch, src/renderer/renderer.cpp:97-130.
#include "mcpt_host.h"

#include <atomic>
#include <chrono>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "commit.hpp"
#include "mcsd_scene.hpp"
#include "../path_core.h"

namespace
{

using namespace mcpt;

thread_local std::string g_error;

// Which pixels a call renders and where each goes: the whole frame, or the 8x8 tiles first, first + stride, ...
// (include/mcpt.h, mcpt_tile_range) written in frame layout or packed tile after tile like the HIP kernel packs them.
struct Work
{
    uint32_t tile_first = 0, tile_stride = 1, n_tiles = 0; // n_tiles == 0: whole frame, pixel order
    bool packed = false;
};

template <uint32_t kFeatures>
void RenderAll(const DeviceScene &sc, float *frame, unsigned workers, const Work &job)
{
    using C = Config<kFeatures>;
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t n = job.n_tiles ? job.n_tiles * 64u : width * height;
    std::atomic<uint32_t> next{0};
    auto work = [&]()
    {
        for (;;)
        {
            const uint32_t begin = next.fetch_add(64); // the reference hands out 64-pixel patches too (renderer.cpp:688-699)
            if (begin >= n)
                break;
            for (uint32_t q = begin; q < std::min(begin + 64, n); ++q)
            {
                uint32_t p = q, slot = q;
                if (job.n_tiles)
                {
                    const uint32_t tile = job.tile_first + (q >> 6) * job.tile_stride, r = q & 63u;
                    const uint32_t x = (tile % tiles_x) * 8u + (r & 7u), y = (tile / tiles_x) * 8u + (r >> 3);
                    if (x >= width || y >= height)
                        continue; // padding of an edge tile: left untouched, like the HIP kernel
                    p = y * width + x;
                    slot = job.packed ? q : p;
                }
                const V3 v = render_pixel<C>(sc, p, nullptr);
                frame[3 * size_t(slot)] = v.x, frame[3 * size_t(slot) + 1] = v.y, frame[3 * size_t(slot) + 2] = v.z;
            }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < workers; ++t)
        pool.emplace_back(work);
    work();
    for (std::thread &t : pool)
        t.join();
}

} // namespace

extern "C"
{

const char *mcpt_host_last_error(void) { return g_error.c_str(); }

int mcpt_host_render(const void *mcsd_bytes, size_t size, int threads, float *frame, double *seconds)
{
    return mcpt_host_render_tiles(mcsd_bytes, size, threads, 0, 1, 0, 0, frame, seconds);
}

int mcpt_host_render_tiles(const void *mcsd_bytes, size_t size, int threads, uint32_t tile_first, uint32_t tile_stride,
                           uint32_t tile_count, int packed, float *frame, double *seconds)
{
    if (!mcsd_bytes || !frame)
    {
        g_error = "null argument";
        return 1;
    }
    try
    {
        const FlatScene flat = CommitScene(mcsd::Parse(static_cast<const uint8_t *>(mcsd_bytes), size));
        const DeviceScene sc = flat.HostView();
        const unsigned workers = threads > 0 ? static_cast<unsigned>(threads) : std::max(1u, std::thread::hardware_concurrency());
        constexpr uint32_t kAll = kFeatVolPath | kFeatEmitters | kFeatAnalytic | kFeatTextures | kFeatMicrofacet;
        constexpr uint32_t kSurface = kFeatEmitters | kFeatTextures | kFeatMicrofacet;
        constexpr uint32_t kO = kFeatOrderedWalk, kV = kFeatOrderedWalk | kFeatVoteWalk;
        const uint32_t f = flat.features;
        const bool ordered = flat.integrator.has_masks == 0; // opacity masks draw during the walk: reference order
        Work job;
        if (!(tile_first == 0 && tile_stride == 1 && tile_count == 0 && !packed))
        {
            const uint32_t tiles = ((uint32_t(sc.camera.width) + 7u) / 8u) * ((uint32_t(sc.camera.height) + 7u) / 8u);
            if (tile_stride == 0)
                throw std::runtime_error("tile_stride is 0");
            const uint32_t available = tile_first >= tiles ? 0u : (tiles - tile_first + tile_stride - 1) / tile_stride;
            job.tile_first = tile_first, job.tile_stride = tile_stride, job.packed = packed != 0;
            job.n_tiles = tile_count == 0 ? available : std::min(tile_count, available);
            if (job.n_tiles == 0)
            {
                if (seconds)
                    *seconds = 0;
                return 0;
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (!ordered)
            RenderAll<kAll>(sc, frame, workers, job);
        else if (flat.integrator.walk_sliver_reach > 0.0f)
            (f & ~kSurface) == 0 ? RenderAll<kSurface | kV | kFeatSlivers>(sc, frame, workers, job)
                                 : RenderAll<kAll | kV | kFeatSlivers>(sc, frame, workers, job);
        else if (f == 0)
            RenderAll<kO>(sc, frame, workers, job);
        else if ((f & ~kFeatEmitters) == 0)
            RenderAll<kFeatEmitters | kO>(sc, frame, workers, job);
        else if ((f & ~kSurface) == 0)
            RenderAll<kSurface | kV>(sc, frame, workers, job);
        else
            RenderAll<kAll | kV>(sc, frame, workers, job);
        if (seconds)
            *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = std::string("error when draw on the host.\n\t") + e.what();
        return 1;
    }
}

} // extern "C"
