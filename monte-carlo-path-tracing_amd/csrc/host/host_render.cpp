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

template <uint32_t kFeatures>
void RenderAll(const DeviceScene &sc, float *frame, unsigned workers)
{
    using C = Config<kFeatures>;
    const uint32_t n = static_cast<uint32_t>(sc.camera.width) * sc.camera.height;
    std::atomic<uint32_t> next{0};
    auto work = [&]()
    {
        for (;;)
        {
            const uint32_t begin = next.fetch_add(64); // the reference hands out 64-pixel patches too (renderer.cpp:688-699)
            if (begin >= n)
                break;
            for (uint32_t p = begin; p < std::min(begin + 64, n); ++p)
            {
                const V3 v = render_pixel<C>(sc, p, nullptr);
                frame[3 * p] = v.x, frame[3 * p + 1] = v.y, frame[3 * p + 2] = v.z;
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
        const auto t0 = std::chrono::steady_clock::now();
        if (!ordered)
            RenderAll<kAll>(sc, frame, workers);
        else if (flat.integrator.walk_sliver_reach > 0.0f)
            (f & ~kSurface) == 0 ? RenderAll<kSurface | kV | kFeatSlivers>(sc, frame, workers)
                                 : RenderAll<kAll | kV | kFeatSlivers>(sc, frame, workers);
        else if (f == 0)
            RenderAll<kO>(sc, frame, workers);
        else if ((f & ~kFeatEmitters) == 0)
            RenderAll<kFeatEmitters | kO>(sc, frame, workers);
        else if ((f & ~kSurface) == 0)
            RenderAll<kSurface | kV>(sc, frame, workers);
        else
            RenderAll<kAll | kV>(sc, frame, workers);
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
