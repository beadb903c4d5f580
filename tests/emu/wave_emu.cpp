// TEST INFRASTRUCTURE — the product's kernel BODIES (csrc/hip/render_kernel_impl.h render_body with csrc/pool_walk.h's cooperative
// ray query and csrc/path_core.h's uniform / merged path steps) compiled for the host behind the 64-lane lockstep shim of
// wave_shim.h, and the scheduler of that shim.  See wave_shim.h for what is modelled and why; tests/test_wave_emu.py for what is
// asserted (frames equal to the oracle's under every lane order and poison pattern; AddressSanitizer + UBSan builds of this file).
// Not part of libmcpt_hip.so, never used by the product path, bench.py or smoke(): the product has no CPU rendering path.
#include <sys/mman.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "host/commit.hpp"
#include "hip/render_kernel_impl.h"
#include "hip/sorted_body.h"

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define WAVE_EMU_ASAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__)
#define WAVE_EMU_ASAN 1
#endif
#if defined(WAVE_EMU_ASAN)
#include <sanitizer/common_interface_defs.h>
#endif
#if defined(__has_feature)
#if __has_feature(memory_sanitizer)
#define WAVE_EMU_MSAN 1
#include <sanitizer/msan_interface.h>
#endif
#endif

namespace wave_emu
{

namespace
{

constexpr uint32_t kLanes = 256, kWaves = kLanes / 64u; // (the most a workgroup has; Workgroup::n_lanes of them run)
#if defined(WAVE_EMU_ASAN)
constexpr size_t kStackBytes = 1024 * 1024;
#else
constexpr size_t kStackBytes = 256 * 1024;
#endif

// A lane's saved stack pointer; the callee-saved registers live on its stack (x86-64 System V).  ucontext's swapcontext makes a
// system call per switch (the signal mask) and, under AddressSanitizer, clears the shadow of the whole stack: hundreds of millions of
// switches per frame want neither.
extern "C" void mcpt_wave_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl mcpt_wave_switch
    .type mcpt_wave_switch, @function
mcpt_wave_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size mcpt_wave_switch, .-mcpt_wave_switch
)");

struct Lane
{
    void *sp; // saved stack pointer while the lane is not running
    char *stack;
    uint32_t tid;
    enum State
    {
        kReady,
        kParked,
        kDone
    } state;
    Kind kind;
    uint32_t site;
    uint32_t depth; // diverged regions this lane stands in (MCPT_WAVE_REGION)
    uint64_t value;
    uint32_t aux;
    void *ptr;
    uint64_t result;
};

struct Workgroup
{
    Lane lanes[kLanes];
    void *scheduler_sp = nullptr;
    uint32_t block = 0, grid = 0, n_lanes = kLanes;
    char *lds = nullptr;
    void (*body)(void *) = nullptr;
    void *body_arg = nullptr;
    // options
    int order = 0; // 0 ascending, 1 descending, 2 shuffled every round
    uint64_t shuffle_state = 0x9E3779B97F4A7C15ull;
    bool poison = false;
    uint32_t poison_word = 0;
    // statistics / errors
    uint64_t collectives = 0, rounds = 0, poisons = 0;
    std::string error;
};

thread_local Workgroup *t_group = nullptr;
thread_local Lane *t_lane = nullptr;
std::atomic<uint64_t> g_ticks{0};

void SwitchToScheduler(Lane *lane)
{
#if defined(WAVE_EMU_ASAN)
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(lane->state == Lane::kDone ? nullptr : &fake, nullptr, 0); // (the scheduler's stack: the thread's own)
#endif
    mcpt_wave_switch(&lane->sp, t_group->scheduler_sp);
#if defined(WAVE_EMU_ASAN)
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

void LaneEntry()
{
#if defined(WAVE_EMU_ASAN)
    __sanitizer_finish_switch_fiber(nullptr, nullptr, nullptr);
#endif
    Workgroup *g = t_group;
    g->body(g->body_arg);
    t_lane->state = Lane::kDone;
    SwitchToScheduler(t_lane);
    std::abort(); // (a finished lane is never resumed)
}

void Resume(Workgroup &g, Lane &lane)
{
    t_lane = &lane;
#if defined(WAVE_EMU_ASAN)
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, lane.stack, kStackBytes);
#endif
    mcpt_wave_switch(&g.scheduler_sp, lane.sp);
#if defined(WAVE_EMU_ASAN)
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
    t_lane = nullptr;
}

// Completes the collective of `members` (lanes of one wavefront parked at the same site).
void Complete(Workgroup &g, const std::vector<Lane *> &members)
{
    const Lane &first = *members.front();
    ++g.collectives;
    for (const Lane *m : members)
        if (m->kind != first.kind)
        {
            g.error = "lanes of one wavefront stand at the same source position with different operations";
            return;
        }
    switch (first.kind)
    {
    case kBallot:
    {
        uint64_t mask = 0;
        for (const Lane *m : members)
            mask |= m->value ? (1ull << (m->tid & 63u)) : 0ull;
        for (Lane *m : members)
            m->result = mask;
        break;
    }
    case kReadFirst:
    {
        const Lane *lowest = members.front();
        for (const Lane *m : members)
            if (m->tid < lowest->tid)
                lowest = m;
        for (Lane *m : members)
            m->result = lowest->value;
        break;
    }
    case kReadLane:
    case kShuffleXor:
        for (Lane *m : members)
        {
            const uint32_t from = first.kind == kReadLane ? m->aux : ((m->tid & 63u) ^ m->aux);
            const Lane *source = nullptr;
            for (const Lane *s : members)
                if ((s->tid & 63u) == from)
                    source = s;
            if (!source && first.kind == kReadLane)
            {
                g.error = "readlane of a lane that is not active";
                return;
            }
            m->result = source ? source->value : m->value;
        }
        break;
    case kPoison:
        ++g.poisons;
        if (g.poison)
        {
            uint32_t *area = static_cast<uint32_t *>(first.ptr);
            for (uint64_t i = 0; i < first.value; ++i)
                area[i] = g.poison_word;
#if defined(WAVE_EMU_MSAN)
            __msan_poison(area, first.value * sizeof(uint32_t)); // (MemorySanitizer build: whatever the area holds is uninitialised to the query)
#endif
        }
        break;
    default:
        break;
    }
    for (Lane *m : members)
        m->state = Lane::kReady;
}

void RunWorkgroup(Workgroup &g)
{
    t_group = &g;
    for (uint32_t t = g.n_lanes; t < kLanes; ++t)
        g.lanes[t].state = Lane::kDone, g.lanes[t].tid = t;
    for (uint32_t t = 0; t < g.n_lanes; ++t)
    {
        Lane &lane = g.lanes[t];
        lane.tid = t, lane.state = Lane::kReady, lane.depth = 0;
        // the frame mcpt_wave_switch pops: control words, r15 ... rbp, then `ret` into LaneEntry (with the stack aligned as a call leaves it)
        uint64_t *top = reinterpret_cast<uint64_t *>(lane.stack + kStackBytes);
        *--top = 0;                                             // (LaneEntry's "return address": never used)
        *--top = reinterpret_cast<uint64_t>(&LaneEntry);
        for (int r = 0; r < 6; ++r)
            *--top = 0;
        uint32_t control[2] = {0x1F80u, 0x037Fu};               // MXCSR and the x87 control word at their defaults
        *--top = uint64_t(control[0]) | uint64_t(control[1]) << 32;
        lane.sp = top;
    }
    uint32_t order[kLanes];
    for (;;)
    {
        ++g.rounds;
        for (uint32_t t = 0; t < kLanes; ++t)
            order[t] = g.order == 1 ? kLanes - 1u - t : t;
        if (g.order == 2)
            for (uint32_t t = kLanes - 1u; t > 0; --t)
            {
                g.shuffle_state ^= g.shuffle_state << 13, g.shuffle_state ^= g.shuffle_state >> 7, g.shuffle_state ^= g.shuffle_state << 17;
                const uint32_t k = static_cast<uint32_t>(g.shuffle_state % (t + 1u));
                const uint32_t s = order[t];
                order[t] = order[k], order[k] = s;
            }
        for (uint32_t i = 0; i < kLanes; ++i)
            if (g.lanes[order[i]].state == Lane::kReady)
                Resume(g, g.lanes[order[i]]);
        if (!g.error.empty())
            break;
        // every lane is parked or done.  Per wavefront: the collective of the lanes deepest inside diverged regions; the top of
        // the persistent loop (and the sleep of an emptied wavefront) only when nothing else waits.
        bool all_done = true, progressed = false;
        uint32_t waves_at_barrier = 0, waves_alive = 0;
        std::vector<Lane *> barrier_members[kWaves];
        for (uint32_t w = 0; w < kWaves && g.error.empty(); ++w)
        {
            bool alive = false, strong = false;
            uint32_t deepest = 0;
            for (uint32_t l = 0; l < 64u; ++l)
            {
                const Lane &lane = g.lanes[64u * w + l];
                if (lane.state != Lane::kParked)
                    continue;
                alive = true;
                if (lane.kind != kConverge && lane.kind != kSleep)
                    deepest = strong ? std::max(deepest, lane.depth) : lane.depth, strong = true;
            }
            if (!alive)
                continue;
            all_done = false, ++waves_alive;
            std::vector<Lane *> members;
            for (uint32_t l = 0; l < 64u; ++l)
            {
                Lane &lane = g.lanes[64u * w + l];
                if (lane.state != Lane::kParked)
                    continue;
                const bool weak = lane.kind == kConverge || lane.kind == kSleep;
                if (strong ? (!weak && lane.depth == deepest) : true)
                    members.push_back(&lane);
            }
            if (!strong)
            {
                // (a `continue`d lane and a sleeping one can meet here: each kind among its own)
                std::vector<Lane *> converging, sleeping;
                for (Lane *m : members)
                    (m->kind == kConverge ? converging : sleeping).push_back(m);
                if (!converging.empty())
                    Complete(g, converging);
                if (!sleeping.empty())
                    Complete(g, sleeping);
                progressed = true;
                continue;
            }
            for (const Lane *m : members)
                if (m->site != members.front()->site)
                {
                    g.error = "lanes of one wavefront stand in the same region at different cross-lane operations: a diverged block without MCPT_WAVE_REGION";
                    break;
                }
            if (!g.error.empty())
                break;
            if (members.front()->kind == kSyncThreads)
            {
                ++waves_at_barrier;
                barrier_members[w] = members;
                continue;
            }
            Complete(g, members);
            progressed = true;
        }
        if (all_done || !g.error.empty())
            break;
        if (!progressed)
        {
            // (s_barrier: the wavefronts that still run meet; a wavefront that has left the kernel is not waited for)
            if (waves_at_barrier != waves_alive)
            {
                g.error = "deadlock: no collective can complete";
                break;
            }
            for (uint32_t w = 0; w < kWaves; ++w)
                if (!barrier_members[w].empty())
                    Complete(g, barrier_members[w]);
        }
    }
    if (!g.error.empty())
    {
        char where[256] = "";
        for (uint32_t t = 0; t < kLanes; ++t)
            if (g.lanes[t].state == Lane::kParked)
            {
                snprintf(where, sizeof where, " (block %u: lane %u parked at site %u:%u, operation %u)", g.block, t, g.lanes[t].site >> 20, g.lanes[t].site & 0xFFFFFu,
                         static_cast<uint32_t>(g.lanes[t].kind));
                break;
            }
        g.error += where;
    }
    t_group = nullptr;
}

} // namespace

uint32_t thread_index() { return t_lane ? t_lane->tid : 0u; }
uint32_t block_index() { return t_group ? t_group->block : 0u; }
uint32_t grid_blocks() { return t_group ? t_group->grid : 1u; }
uint32_t block_size() { return t_group ? t_group->n_lanes : 1u; }
void *dynamic_lds_base() { return t_group ? t_group->lds : nullptr; }
uint64_t tick() { return g_ticks.fetch_add(1, std::memory_order_relaxed); }
void region_enter()
{
    if (t_lane)
        ++t_lane->depth;
}
void region_leave()
{
    if (t_lane)
        --t_lane->depth;
}

uint32_t site_rank(const char *file, int line)
{
    // (an identity for the consistency check of the scheduler and for error messages: file hash, line)
    const char *base = strrchr(file, '/');
    base = base ? base + 1 : file;
    uint32_t h = 0;
    for (const char *c = base; *c; ++c)
        h = h * 31u + static_cast<uint8_t>(*c);
    return (h & 0xFFFu) << 20 | static_cast<uint32_t>(line);
}

uint64_t collective(Kind kind, uint32_t site, uint64_t value, uint32_t aux, void *ptr)
{
    Lane *lane = t_lane;
    if (!lane)
        return kind == kBallot ? (value ? 1u : 0u) : (kind == kReadFirst || kind == kReadLane || kind == kShuffleXor) ? value : 0u; // a "wavefront" of one lane
    lane->kind = kind, lane->site = site, lane->value = value, lane->aux = aux, lane->ptr = ptr, lane->state = Lane::kParked;
    SwitchToScheduler(lane);
    return lane->result;
}

} // namespace wave_emu

namespace
{

using namespace mcpt;

thread_local std::string g_error;

struct Options
{
    uint32_t order;       // lane order between collectives: 0 ascending, 1 descending, 2 shuffled
    uint32_t seed;        // ... of the shuffle
    uint32_t poison;      // 1: fill a wavefront's pool area with `poison_word` before every ray query
    uint32_t poison_word;
    uint32_t max_blocks;  // "CUs" of the launch (0: 4)
    uint32_t per_cu;      // resident workgroups per CU the launch is shaped for (0: 1)
    uint32_t lane_spread; // RenderJob::lane_spread (0: the launcher's rule)
    uint32_t compact;     // RenderJob::compact
    uint32_t scatter;     // RenderJob::scatter (0xFFFFFFFF: the launcher's rule)
    uint32_t threads;     // host threads, one workgroup each at a time (0: all cores)
    uint32_t xcd_bands;     // RenderJob::xcd_bands: the hand-out in eight bands, one counter each
    uint32_t lds_shortfall; // bytes the LDS array is SHORTER than the launch asks for (the sanitizer build's self-test: it must notice)
};

struct Report
{
    uint64_t collectives, rounds, queries;
    uint32_t blocks, lds_bytes, lane_spread, scatter;
};

template <uint32_t kFeatures, bool kCount, bool kLdsGeometry>
struct BodyArgs
{
    const DeviceScene *sc;
    const RenderJob *job;
    float *out;
    TraceCounters *counters;
    static void Run(void *p)
    {
        const BodyArgs *a = static_cast<const BodyArgs *>(p);
        render_body<kFeatures, kCount, kLdsGeometry>(*a->sc, *a->job, a->out, a->counters);
    }
};

// (the path market's words, RenderJob::market: one array per render call of the calling thread)
thread_local std::vector<uint32_t> t_market;

RenderJob FilmJob(const DeviceScene &sc, const Options &opt, uint32_t *work_counter)
{
    for (uint32_t i = 0; i < kBands * kBandStride; ++i)
        work_counter[i] = 0;
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height);
    const uint32_t tiles_x = (width + 7u) / 8u, tiles_y = (height + 7u) / 8u;
    RenderJob job{};
    job.n_items = tiles_x * tiles_y * 64u, job.tile_first = 0, job.tile_stride = 1, job.tiles_x = tiles_x;
    job.work_counter = work_counter;
    const uint32_t events = opt.compact & 3u; // (bit 2 of the option: the camera-ray pre-pass, WithPrepass below)
    job.lane_spread = opt.lane_spread, job.compact = events, job.tail_spread = events /* (one switch for both kinds of event) */, job.scatter = opt.xcd_bands ? 0u : opt.scatter, job.pool_walk = 1;
    job.xcd_bands = opt.xcd_bands;
    if (events >= 2) // (2: the tail spread with the path market between workgroups)
    {
        t_market.assign(kMarketWords, 0u);
        job.market = t_market.data();
    }
    return job;
}

void RunGrid(void (*body)(void *), void *body_arg, uint32_t n_lanes, uint64_t blocks, size_t lds_bytes, const RenderJob &shaped, const Options &opt, Report *report);

// The camera-ray PRE-PASS on the host (hip/primary_kernel.hip's table, DeviceScene::prehit): the closest hit of every (pixel, sample)'s
// camera ray by the per-lane ordered walk — so that the lockstep build runs what the kernels do with it: samples that start at their
// first vertex, and the next sample in the same step (path_core.h regenerate_in_step).
thread_local std::vector<uint32_t> t_prehit;
DeviceScene WithPrepass(const DeviceScene &sc_in)
{
    DeviceScene sc = sc_in;
    const uint32_t width = static_cast<uint32_t>(sc.camera.width), height = static_cast<uint32_t>(sc.camera.height), spp = sc.camera.spp;
    const uint32_t tiles_x = (width + 7u) / 8u, tiles_y = (height + 7u) / 8u;
    t_prehit.assign(size_t(2) * tiles_x * tiles_y * 64u * spp, 0u);
    for (uint32_t item = 0; item < tiles_x * tiles_y * 64u; ++item)
    {
        const uint32_t tile = item >> 6, r = item & 63u, x = (tile % tiles_x) * 8u + (r & 7u), y = (tile / tiles_x) * 8u + (r >> 3);
        if (x >= width || y >= height)
            continue;
        for (uint32_t s = 0; s < spp; ++s)
        {
            PathState st{};
            st.pixel = y * width + x, st.sample = s;
            start_sample(sc, st);
            Ray ray = make_ray(st.origin, st.dir);
            HitRaw hit;
            hit.inst = hit.prim = 0, hit.a = hit.b = hit.c = 0.0f, hit.inside = false;
            TraceStats ts{0, 0, 0, 0};
            uint32_t stack[kWalkStackMax];
            const bool found = walk_ordered<false, true, false, true, 1u>(sc, stack, ray, hit, ts);
            uint32_t *rec = t_prehit.data() + 2 * (size_t(item) * spp + s);
            rec[0] = found ? hit.prim : kNone, rec[1] = found ? hit.inst : 0u;
        }
    }
    sc.prehit = t_prehit.data(), sc.prehit_step = 1, sc.prehit_tile_first = 0, sc.prehit_tile_stride = 1, sc.prehit_tiles_x = tiles_x;
    return sc;
}

template <uint32_t kFeatures, bool kCount, bool kLdsGeometry>
void RenderGrid(const DeviceScene &sc_in, const Options &opt, float *frame, TraceCounters *counters, Report *report)
{
    const DeviceScene sc = (opt.compact & 4u) ? WithPrepass(sc_in) : sc_in;
    uint32_t work_counter[kBands * kBandStride];
    const RenderJob job = FilmJob(sc, opt, work_counter);
    RenderJob shaped;
    const uint64_t blocks = ShapeLaunch<kFeatures, kLdsGeometry>(job, opt.per_cu ? static_cast<int>(opt.per_cu) : 1, opt.max_blocks ? opt.max_blocks : 4u, shaped);
    BodyArgs<kFeatures, kCount, kLdsGeometry> args{&sc, &shaped, frame, counters};
    RunGrid(&BodyArgs<kFeatures, kCount, kLdsGeometry>::Run, &args, 256u, blocks, LaunchLdsBytes<kFeatures, kCount, kLdsGeometry>(sc), shaped, opt, report);
}

// the class-sorted kernel (hip/sorted_body.h): workgroups of kSortLanes lanes, every lane a path, shaped like LaunchSorted does
template <uint32_t kFeatures, bool kLdsGeometry>
struct SortedArgs
{
    const DeviceScene *sc;
    const RenderJob *job;
    float *out;
    static void Run(void *p)
    {
        const SortedArgs *a = static_cast<const SortedArgs *>(p);
        sorted_body<kFeatures, kLdsGeometry>(*a->sc, *a->job, a->out);
    }
};

template <uint32_t kFeatures, bool kLdsGeometry>
void RenderGridSorted(const DeviceScene &sc, const Options &opt, float *frame, Report *report)
{
    uint32_t work_counter[kBands * kBandStride];
    RenderJob shaped = FilmJob(sc, opt, work_counter);
    shaped.lane_spread = 1, shaped.scatter = 0, shaped.sort_classes = 1;
    const uint64_t resident = uint64_t(opt.max_blocks ? opt.max_blocks : 4u) * (opt.per_cu ? opt.per_cu : 1u);
    uint64_t blocks = (uint64_t(shaped.n_items) + kSortLanes - 1) / kSortLanes;
    blocks = blocks > resident ? resident : blocks;
    SortedArgs<kFeatures, kLdsGeometry> args{&sc, &shaped, frame};
    RunGrid(&SortedArgs<kFeatures, kLdsGeometry>::Run, &args, kSortLanes, blocks, SortedLdsBytes<kFeatures, kLdsGeometry>(sc), shaped, opt, report);
}

void RunGrid(void (*body)(void *), void *body_arg, uint32_t n_lanes, uint64_t blocks, size_t lds_bytes, const RenderJob &shaped, const Options &opt, Report *report)
{
    if (report)
        report->blocks = static_cast<uint32_t>(blocks), report->lds_bytes = static_cast<uint32_t>(lds_bytes), report->lane_spread = shaped.lane_spread, report->scatter = shaped.scatter;
    if (lds_bytes > 160u * 1024u)
        throw std::runtime_error("the launch asks for more LDS than a CU has");
    std::atomic<uint32_t> next{0};
    std::atomic<uint64_t> collectives{0}, rounds{0}, queries{0};
    std::string error;
    std::mutex error_lock;
    const unsigned n_threads = std::max(1u, std::min<unsigned>(opt.threads ? opt.threads : std::thread::hardware_concurrency(), static_cast<unsigned>(blocks)));
    auto work = [&]()
    {
        wave_emu::Workgroup *g = new wave_emu::Workgroup();
        char *stacks = static_cast<char *>(mmap(nullptr, wave_emu::kStackBytes * wave_emu::kLanes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (stacks == MAP_FAILED)
            throw std::runtime_error("no memory for the lanes' stacks");
        for (uint32_t t = 0; t < wave_emu::kLanes; ++t)
            g->lanes[t].stack = stacks + wave_emu::kStackBytes * t;
        // (a heap array of exactly the launch's size: AddressSanitizer sees what lies beyond)
        if (opt.lds_shortfall >= lds_bytes)
            throw std::runtime_error("lds_shortfall");
        const size_t lds_given = lds_bytes - opt.lds_shortfall;
        // (16-byte alignment by hand: the array must END where the launch's LDS ends)
        std::vector<char> lds_store(lds_given + 16);
        char *lds_base = lds_store.data() + lds_store.size() - lds_given;
        lds_base -= reinterpret_cast<uintptr_t>(lds_base) & 15u;
        if (lds_base < lds_store.data())
            throw std::runtime_error("lds alignment");
        // (the slack behind an aligned start is given back: resize so that the vector ends with the area)
        lds_store.resize(static_cast<size_t>(lds_base - lds_store.data()) + lds_given);
        for (;;)
        {
            const uint32_t b = next.fetch_add(1);
            if (b >= blocks)
                break;
            g->block = b, g->grid = static_cast<uint32_t>(blocks), g->n_lanes = n_lanes;
            g->lds = lds_base;
            memset(g->lds, 0xA5, lds_given);
#if defined(WAVE_EMU_MSAN)
            __msan_poison(g->lds, lds_given); // (a workgroup's LDS starts uninitialised)
#endif
            g->body = body, g->body_arg = body_arg;
            g->order = static_cast<int>(opt.order), g->shuffle_state = 0x9E3779B97F4A7C15ull ^ (uint64_t(opt.seed) << 32 | b);
            g->poison = opt.poison != 0, g->poison_word = opt.poison_word;
            g->collectives = g->rounds = g->poisons = 0;
            g->error.clear();
            wave_emu::RunWorkgroup(*g);
            collectives += g->collectives, rounds += g->rounds, queries += g->poisons;
            if (!g->error.empty())
            {
                std::lock_guard<std::mutex> hold(error_lock);
                if (error.empty())
                    error = g->error;
                next = static_cast<uint32_t>(blocks);
            }
        }
        munmap(stacks, wave_emu::kStackBytes * wave_emu::kLanes);
        delete g;
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t)
        pool.emplace_back(work);
    work();
    for (std::thread &t : pool)
        t.join();
    if (report)
        report->collectives = collectives, report->rounds = rounds, report->queries = queries;
    if (!error.empty())
        throw std::runtime_error(error);
}

} // namespace

// ---- the scheduler's own checks: a body with a diverged block, with and without its mark ----
struct SelfTest
{
    bool marked;
    std::atomic<uint32_t> failures{0};
    static void Run(void *p)
    {
        SelfTest *t = static_cast<SelfTest *>(p);
        const uint32_t lane = __lane_id();
        unsigned long long inside = 0;
        if (lane % 3u == 0u)
        {
            if (t->marked)
            {
                MCPT_WAVE_REGION();
                inside = __ballot((lane & 1u) != 0u);
            }
            else
                inside = __ballot((lane & 1u) != 0u);
        }
        const unsigned long long all = __ballot(true);
        const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lane + 100u)));
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(all >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(all), 0u));
        unsigned long long expect_inside = 0;
        for (uint32_t l = 0; l < 64u; l += 3u)
            expect_inside |= (l & 1u) ? 1ull << l : 0ull;
        if (all != ~0ull || first != 100u || rank != lane || (lane % 3u == 0u && inside != expect_inside))
            t->failures.fetch_add(1);
        __syncthreads();
    }
};

extern "C"
{

const char *mcpt_wave_emu_last_error(void) { return g_error.c_str(); }

// 0: a diverged block WITH its MCPT_WAVE_REGION mark — ballots, readfirstlane and mbcnt ranks must come out as on a wavefront
// (returns the number of lanes that saw something else).  1: the same block WITHOUT the mark — the scheduler must refuse it
// (returns 0 when it did, the error text is in mcpt_wave_emu_last_error).
int mcpt_wave_emu_selftest(int unmarked)
{
    SelfTest test;
    test.marked = unmarked == 0;
    wave_emu::Workgroup *g = new wave_emu::Workgroup();
    std::vector<char> stacks(wave_emu::kStackBytes * wave_emu::kLanes + 64);
    char *base = stacks.data() + (64u - (reinterpret_cast<uintptr_t>(stacks.data()) & 63u));
    for (uint32_t t = 0; t < wave_emu::kLanes; ++t)
        g->lanes[t].stack = base + wave_emu::kStackBytes * t;
    g->block = 0, g->grid = 1, g->lds = nullptr, g->body = &SelfTest::Run, g->body_arg = &test, g->order = 2;
    wave_emu::RunWorkgroup(*g);
    g_error = g->error;
    const bool refused = !g->error.empty();
    delete g;
    if (unmarked)
        return refused ? 0 : 1;
    return refused ? -1 : static_cast<int>(test.failures.load());
}

int mcpt_wave_emu_film(const char *mcsd_path, uint32_t *width, uint32_t *height)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        *width = static_cast<uint32_t>(flat.camera.width), *height = static_cast<uint32_t>(flat.camera.height);
        return 0;
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// The kernel body of the instantiation <features, counted, LDS-resident> on the scene of an .mcsd file, 256 lanes per workgroup in
// lockstep.  frame: width x height x 3 floats.  counters: 8 x u64 (TraceCounters) or NULL for an uncounted instantiation.
int mcpt_wave_emu_render(const char *mcsd_path, uint32_t features, int lds_geometry, const Options *options, float *frame, unsigned long long *counters, Report *report)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        const Options opt = options ? *options : Options{0, 0, 0, 0, 4, 1, 0, 0, 0xFFFFFFFFu, 0, 0, 0};
        if (report)
            *report = Report{};
        constexpr uint32_t kScene = kAll | kFeatSlivers; // what a scene can ask of an instantiation
        const uint32_t needs = flat.features | (flat.integrator.walk_sliver_reach > 0.0f ? uint32_t(kFeatSlivers) : 0u);
        if ((needs & kScene & ~features) != 0)
            throw std::runtime_error("the instantiation does not cover the scene's features");
        const bool lds_fits = StagedBytes(sc, true, (features & kFeatPoolWalk) != 0) <= kLdsGeometryBytes;
        if (lds_geometry && !lds_fits)
            throw std::runtime_error("the scene's traversal data does not fit LDS");
        TraceCounters *tc = reinterpret_cast<TraceCounters *>(counters);
#define MCPT_EMU_CASE(F, COUNT, LDS)                                                                 \
    if (features == (F) && (counters != nullptr) == (COUNT) && (lds_geometry != 0) == (LDS))          \
    {                                                                                                \
        RenderGrid<(F), COUNT, LDS>(sc, opt, frame, tc, report);                                      \
        return 0;                                                                                    \
    }
        MCPT_EMU_CASE(kO, false, true)                       // the per-lane walk in the same loop
        MCPT_EMU_CASE(kP, false, true)                       // cornell: the production kernel
        MCPT_EMU_CASE(kPM, false, true)                      // ... with merged queries
        MCPT_EMU_CASE(kFeatEmitters | kP, false, true)       // LDS-resident scenes with emitters
        MCPT_EMU_CASE(kFeatEmitters | kPM, false, true)
        MCPT_EMU_CASE(kFeatEmitters | kPB, false, false)     // dragon's class: outside LDS, merged
        MCPT_EMU_CASE(kFeatEmitters | kPBU, false, false)    // ... two queries per vertex
        MCPT_EMU_CASE(kSurface | kPB | kS, false, false)     // surface materials outside LDS, sliver rules
        MCPT_EMU_CASE(kAll | kP, true, true)                 // the counted LDS form
#undef MCPT_EMU_CASE
        throw std::runtime_error("this instantiation is not part of the lockstep build");
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

// The class-sorted kernel's body (hip/sorted_body.h) of the instantiation <features, LDS-resident>: workgroups of 128 lanes.
int mcpt_wave_emu_render_sorted(const char *mcsd_path, uint32_t features, int lds_geometry, const Options *options, float *frame, Report *report)
{
    try
    {
        const FlatScene flat = CommitScene(mcsd::Load(mcsd_path));
        const DeviceScene sc = flat.HostView();
        const Options opt = options ? *options : Options{0, 0, 0, 0, 4, 1, 0, 0, 0xFFFFFFFFu, 0, 0, 0};
        if (report)
            *report = Report{};
        if ((flat.features & kAll & ~features) != 0 || flat.integrator.walk_sliver_reach > 0.0f || flat.integrator.has_masks)
            throw std::runtime_error("the instantiation does not cover the scene's features");
        if (lds_geometry && StagedBytes(sc, true, (features & kFeatPoolWalk) != 0) > kLdsGeometryBytes)
            throw std::runtime_error("the scene's traversal data does not fit LDS");
        constexpr uint32_t kG = kFeatGroup128;
#define MCPT_EMU_CASE(F, LDS)                                       \
    if (features == (F) && (lds_geometry != 0) == (LDS))            \
    {                                                               \
        RenderGridSorted<(F), LDS>(sc, opt, frame, report);         \
        return 0;                                                   \
    }
        MCPT_EMU_CASE(kVolumeLean | kO | kG, true)                        // volumetric-caustic: the production kernel
        MCPT_EMU_CASE(kAll | kO | kG, true)
        MCPT_EMU_CASE(kVolumeLean | kP | kG, true)                        // ... with the pool walk inside
        MCPT_EMU_CASE(kSurface | kPBU | kG, false)                        // surface-material meshes outside LDS (round 6)
        MCPT_EMU_CASE(kSurface | kPBU | kFeatConductorOnly | kG, false)
        MCPT_EMU_CASE(kSurface | kPBU | kFeatDielectricOnly | kG, false)
#undef MCPT_EMU_CASE
        throw std::runtime_error("this instantiation is not part of the lockstep build");
    }
    catch (const std::exception &e)
    {
        g_error = e.what();
        return 1;
    }
}

} // extern "C"
