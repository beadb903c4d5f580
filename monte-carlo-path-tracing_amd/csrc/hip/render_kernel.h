// Launch interface between the C-ABI layer (capi.cpp) and the HIP kernel.
#ifndef MCPT_RENDER_KERNEL_H
#define MCPT_RENDER_KERNEL_H

#include "../wave_target.h"
#if defined(MCPT_WAVE_EMU)
typedef int hipError_t; // (the lockstep host build of the kernel bodies, tests/emu: declarations only)
typedef void *hipStream_t;
#else
#include <hip/hip_runtime_api.h>
#endif

#include "../device_scene.h"

namespace mcpt
{

constexpr int kBlockSize = 256;
constexpr uint32_t kHitCounters = 32;
constexpr uint32_t kScatterAuto = 0xFFFFFFFFu;
constexpr uint32_t kMarketSlots = 16384, kMarketRecord = 48, kMarketReadyAt = 128, kMarketRecordsAt = kMarketReadyAt + kMarketSlots;
constexpr size_t kMarketWords = kMarketRecordsAt + size_t(kMarketSlots) * kMarketRecord; // RenderJob::market (3.2 MB); the launch's caller zeroes the first kMarketRecordsAt
constexpr uint32_t kBands = 8, kBandStride = 32; // RenderJob::xcd_bands: one counter per XCD, 128 bytes apart
constexpr size_t kWaveClockWords = 4u * 8u * 4u, kPhaseSumWords = 64; // RenderJob::wave_clock: words per CU (at most 8 workgroups of 4 wavefronts, four words each); phase_sums
bool LastLaunchTransposed(); // what the calling thread's last LaunchRender chose (for the kernel description)
// lane_spread from the job's EXPENSIVE pixels (camera ray hits something): the largest power of two with
// spread <= kSpreadNum / kSpreadDen * launched lanes / expensive pixels.  Fitted to rank-share measurements
// (profiles/r02_experiments/lane_spread_rank_shares.json): the best spread leaves 2-3 expensive pixels per active lane.
#ifndef MCPT_SPREAD_NUM
#define MCPT_SPREAD_NUM 3
#endif
constexpr uint32_t kSpreadNum = MCPT_SPREAD_NUM, kSpreadDen = 1, kMaxStreamSpread = 16;
// Traversal data (both hierarchies + triangle positions) up to this size is staged
// in LDS, next to the traversal stacks (walk_depth x 256 x 4 B per workgroup):
// 4 workgroups per CU x (24 KiB + stacks) leaves room in the 160 KiB of a CU.
constexpr size_t kLdsGeometryBytes = 24 * 1024;

// Work description of one launch: pixels are enumerated tile by tile
// (8x8 pixel tiles, row-major tile order); item q -> local tile q / 64, pixel
// q % 64 inside it; global tile = tile_first + local_tile * tile_stride.
struct RenderJob
{
    uint32_t n_items;     // 64 * number of tiles handled by this launch
    uint32_t tile_first;  // first global tile
    uint32_t tile_stride; // distance between consecutive tiles of this launch
    uint32_t tiles_x;     // tiles per image row
    uint32_t packed;      // 0: write frame layout, 1: write packed tile layout
    uint32_t reference_walk; // 1: force the reference-order walk (validation); masks force it anyway
    // independent-sample RNG mode (mcpt_renderer_set_rng; lane-owns-a-path kernel only).  sample_split = K >= 1:
    // work item q = k * n_items + item renders samples k, k + K, ... of the item's pixel; with K > 1 `out` holds K
    // planes of `plane_stride` pixels of UNNORMALISED sums that ReduceSamplePlanes folds into the frame.
    uint32_t independent_samples, rng_seed, sample_split, plane_stride;
    // Dynamic work distribution (null: every lane walks its fixed list q, q + stride, ...): a zeroed counter in HBM;
    // a lane that has finished a pixel takes the next item not yet handed out (one atomic per wavefront and fetch).
    // Pixels cost very different amounts (a camera ray that leaves the scene against a 17-bounce path on glass), and
    // with fixed lists the frame lasts as long as the unluckiest wavefront's list.  Items are handed out in image (tile)
    // order: lanes that fetch at about the same time work on neighbouring pixels (a permuted order — tiles far apart —
    // was measured: matpreview 400 against 436 Msamples/s, the locality is worth more than the spread).
    // Lanes per item slot.  A job with fewer items than the GPU holds lanes (a rank's share of a strong-scaling run) is
    // spread: only every lane_spread-th lane takes items, so that every wavefront slot of the GPU is used and each
    // wavefront carries fewer paths — fewer diverged instructions per wavefront, and a pixel's chain of samples (the
    // reference's one RNG stream per pixel makes it sequential) gets shorter.  0: the launcher's choice; 1: dense.
    uint32_t lane_spread;
    // Pixel order of the lane-owns-a-path kernel.  0: item q (the 64 lanes of a wavefront render one 8x8 tile).
    // 1: TRANSPOSED — lane l of the w-th wavefront's worth of items takes item l * (items / 64) + w: 64 different tiles.
    // kScatterAuto: the launcher's choice — transposed when the traversal data sits in LDS (no locality to lose) and the
    // job gives every lane at most one pixel (nothing left for the work counter to balance): a frame then lasts as long
    // as its slowest wavefront, and a tile of expensive pixels makes a slow one; transposed, every wavefront holds the
    // same mix and thins out as its cheap pixels finish (cornell 512x512: 65.3 -> 61.0 ms; volumetric-caustic, 3.5
    // pixels per lane: 140 -> 149 ms, so not there).
    uint32_t scatter;
    // kHitCounters words (may be null): the pre-pass adds the number of camera rays that hit something — what the
    // expensive part of the job is.  The stream kernel sizes lane_spread from it (lane_spread 0 only).
    uint32_t *hit_counters;
    uint32_t *work_counter;
    // XCD BANDS (round 6; image order only: no tile_order, no scatter, no sample split).  1: the hand-out positions are cut into eight
    // contiguous bands (horizontal stripes of the film), band b has its own counter (work_counter[kBandStride * b], all zeroed), and a
    // workgroup takes from the band of the XCD it runs on (HW_REG_XCC_ID) while that has items, from the next bands after that.  Each
    // of MI355X's eight XCDs has its own 4 MB L2: the wavefronts of one XCD then work on one region of the film — camera rays, first
    // bounces and shadow rays that touch one region of the scene — instead of all eight L2s each caching all of it.
    uint32_t xcd_bands;
    // Hand-out order of the tiles (null: image order).  The work counter hands out items 0, 1, 2, ...; with a table,
    // hand-out position p stands for local tile (uint32_t)tile_order[p >> 6] — the tiles sorted MOST EXPENSIVE FIRST
    // (hip/tile_order.hip: a cost from the pre-pass's camera-ray hits).  The reference's one random stream per pixel makes
    // a pixel a sequential chain, so a frame ends with the chains that started last: those should be the short ones.
    // Only with the work counter; packed output keeps the image-order layout.
    const unsigned long long *tile_order;
    // Lane-owns-a-path kernel on LDS-resident scenes: 1 = once the items are handed out, the paths still in flight are
    // dealt out over the four wavefronts of their workgroup (rounds 3-5: packed into its first ones) whenever another 64 of its
    // lanes have retired (render_kernel_impl.h, "EVENTS OF A THINNING WORKGROUP").  The image does not depend on it.
    uint32_t compact;
    // Pool-walk kernels OUTSIDE LDS with the work counter (round 6): 1 = once the counter is dry, the paths a workgroup still holds are
    // dealt out evenly over its four wavefronts whenever few enough of them are left (render_kernel_impl.h, "TAIL SPREAD"): a frame
    // of dragon/scene.xml ends on wavefronts that hold 32 long pixel chains each while their three neighbours have nothing left, and
    // a chain runs faster the more helper lanes its wavefront has.  The image does not depend on it.
    uint32_t tail_spread;
    // ... and BETWEEN workgroups (null: off): the path market, kMarketWords words of device memory, header and ready words zeroed
    // before the launch.  A wavefront whose workgroup is done does not leave: it takes a ticket and waits for a path; a wavefront
    // that still holds two or more paths after the counter ran dry gives half of them (at most one per waiting ticket) away — state
    // and pending shadow ray through device memory — so that the frame's last paths run one per wavefront, with 63 helper lanes each,
    // on ALL wavefronts of the GPU instead of 32 per wavefront on a few dozen (render_kernel_impl.h, "PATH MARKET").  Words: [0] tickets
    // taken, [32] records given, [64] items of the job finished (a waiting wavefront leaves when that reaches the job's items), [96] workgroups of the launch that
    // have started (a wavefront waits only when that is all of them: nobody waits for a workgroup that is not resident),
    // [128 + s] generation of slot s, records of 48 words behind them.  The image does not depend on it.
    uint32_t *market;
    // Device memory for the records of a launch (the kernels that read the scene and the job through a pointer: LaunchRecords below,
    // render_kernel_impl.h records_behind_pointer); the renderer's own.  Host-side meaning only.
    struct LaunchRecords *launch_records;
    // Full-feature scenes with the traversal data in LDS: 1 = the class-sorted kernel (hip/sorted_kernel.hip: the paths of a
    // workgroup are regrouped by what their ray found, between the ray query and the shading).  The image does not depend on it.
    uint32_t sort_classes;
    // Cost probe of the lane-owns-a-path kernel (may be null): n_items / 64 words; a lane adds the steps its pixel took to
    // its tile's word when the pixel is finished.  A low-spp draw with this set measures what each tile costs.
    uint32_t *tile_steps;
    // LDS-resident scenes without slivers: the ray queries as the wavefront-cooperative pool walk (pool_walk.h).  1 = in the
    // lean instantiations (where it is the measured choice), 2 = wherever an instantiation with it exists (also the class-sorted
    // full-feature kernels).  The image does not depend on it.
    uint32_t pool_walk;
    // LANES PER PATH BY TILE COST (pool-walk kernels with a probed tile_order; all 0: off).  The tiles are handed out most expensive
    // first, and a pixel is one sequential chain: a tile whose 64 chains would fill a wavefront for most of the frame is the frame's
    // critical path.  Positions below level_until[0] / [1] / [2] are therefore handed out to every 8th / 4th / 2nd lane of a wavefront
    // only — the lanes in between are helpers of the wavefront's ray queries (pool_walk.h), so those chains run at 0.5 - 0.7 of the
    // dense time — and a wavefront stays that sparse while it holds a pixel of such a position (render_kernel_impl.h).  Computed by
    // the host from the probe's tile costs (capi.cpp, LevelThresholds); the second set is for launches at 4 wavefronts per SIMD.
    uint32_t level_until[3], level_until_4[3];
    // DIAGNOSTIC (may be null; MCPT_WAVE_CLOCK=<file> sets it, capi.cpp): four words per wavefront of the launch — the constant
    // 100 MHz clock (s_memrealtime) when the wavefront starts, when it leaves the kernel and when it last took a pixel, and the
    // number of pixels it took.  What a frame's tail looks like.
    unsigned long long *wave_clock;
    // ... and, in builds with -DMCPT_PHASE_CLOCK=1, 64 words of their own behind them for the per-phase sums of all wavefronts
    // (phase_clock.h; round 5's advisor: they used to sit at a fixed offset that assumed 256 CUs)
    unsigned long long *phase_sums;
};

// ---- the records of one launch, behind a pointer (round 6) -----------------------------------------
// The render kernels used to take the scene's and the job's records by value: ~1.3 KB of kernel arguments, which the compiler loads
// once, keeps in scalar registers across the persistent loop and — 106 SGPRs — spills into lanes of vector registers (116-210 spilled
// scalars per kernel, v_readlane / v_writelane around their uses).  The reference's kernel takes two pointers (renderer.cpp:88).  The
// kernels of the instantiations named by records_behind_pointer() (render_kernel_impl.h) take ONE pointer to {scene, job} in device
// memory (RenderJob::launch_records) and read the fields through the constant address space: scalar loads that the compiler may repeat at
// a use instead of keeping the value — 67-131 spilled scalars, cornell -1.6 %, volumetric-caustic -1.8 %, matpreview -0.6 ... -0.9 %
// (EXPERIMENTS R6-8).
struct LaunchRecords
{
    DeviceScene sc;
    RenderJob job;
};
// Writes the records to `job.launch_records` (device memory of sizeof(LaunchRecords) bytes that belongs to the caller — a renderer owns
// one, like its work counter: its draws are ordered on one stream at a time) in stream order: a one-wavefront kernel that takes the two
// records by value and copies its own argument segment.  No host buffer outlives the call, nothing waits.  The kernel that reads the
// records is launched behind it on the same stream.
hipError_t StageLaunchRecords(const DeviceScene &sc, const RenderJob &job, hipStream_t stream);
// Whether LaunchRender (no counters) runs an instantiation with the tail spread and the path market for this job: the diffuse
// (+ emitters, slivers) pool-walk kernels outside LDS.
bool TailSpreadRuns(const DeviceScene &sc, const RenderJob &job);

// ---- stream kernel (stream_core.h, stream_kernel_impl.h) ------------------------------------------
// Launch parameters of the stream kernel.  `slots` and `refill_at` are inputs of PlanRenderStream (0 = the
// built-in choice for the scene), the rest is filled in by it.
struct StreamLaunch
{
    uint32_t slots_in_memory; // input: 0 = one slot per lane, path state in registers; 1 = `slots` slots per workgroup in memory
    uint32_t lane_spread; // output: RenderJob::lane_spread as resolved by the plan (one slot per lane only)
    uint32_t waves;      // input: wavefronts per SIMD the instantiation is compiled for (2, 3; 0 / 4: the default budget)
    uint32_t wave_local; // input (one slot per lane only): 1 = every wavefront runs its rounds alone, no workgroup barrier
    uint32_t slots;      // path slots per workgroup (a multiple of 256)
    uint32_t refill_at;  // a wavefront fetches new rays when this many of its lanes are free
    uint32_t blocks, blocks_per_cu, lds_bytes;
    uint32_t scratch_words_per_block; // the workgroup's region of the scratch buffer, in 32-bit words
    uint32_t variant;    // which instantiation (index into the dispatcher's table)
};

// Can this job run on the stream kernel?  (No opacity masks — they draw random numbers during a walk, so the
// visiting order is part of the image —, not the reference-order validation walk, at most kStreamMaxShadow shadow
// rays per vertex, a non-empty scene.)
bool StreamSupports(const DeviceScene &sc, const RenderJob &job);
// Scene class for which the lane-owns-a-path kernel is the faster one: traversal data small enough for LDS.
bool StreamPrefersLanes(const DeviceScene &sc);
// Scene class (outside LDS) that the lane-owns-a-path kernel can run with the wavefront-cooperative pool walk (pool_walk.h).
bool PoolBigSupports(const DeviceScene &sc);
// Whether the formulations the rule no longer chooses (stream kernel, queued renderer, mode 3, the trace-rate experiment) are part
// of this build (`make EXPERIMENTAL=1`; hip/formulations_not_built.hip otherwise: their `...Supports` say no)
bool FormulationsBuilt();
// Chooses the instantiation and the launch shape; `name` receives a description.  The caller provides a scratch
// buffer of at least blocks * scratch_words_per_block words and then calls LaunchRenderStream with the same cfg.
hipError_t PlanRenderStream(const DeviceScene &sc, const RenderJob &job, bool counted, uint32_t n_cus, StreamLaunch *cfg,
                            const char **name);
hipError_t LaunchRenderStream(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters,
                              hipStream_t stream, uint32_t *scratch, const StreamLaunch &cfg);

// Throughput mode 2 (job.independent_samples == 2): the instantiations whose random draws are Owen-scrambled Sobol points
// (hip/render_variants_lowdisc.hip).
hipError_t LaunchRenderLowDiscrepancy(const DeviceScene &sc, const RenderJob &job, float *out, hipStream_t stream, uint32_t n_cus,
                                      const char **variant);

hipError_t LaunchRender(const DeviceScene &sc, const RenderJob &job, float *out, TraceCounters *counters,
                        hipStream_t stream, uint32_t n_cus, const char **variant);
// The class-sorted form of the lane-owns-a-path kernel (hip/sorted_kernel.hip); hipErrorNotSupported when the scene is
// not one of its classes (the caller takes LaunchRender).
// keys[t] = (0xFFFFFFFF - steps[t]) << 32 | t, sorted ascending into `sorted`: the tiles most expensive first.
hipError_t LaunchTileOrderFromSteps(const uint32_t *steps, uint32_t n_tiles, unsigned long long *keys, unsigned long long *sorted,
                                    void *temp, size_t temp_bytes, hipStream_t stream);
hipError_t LaunchRenderSorted(const DeviceScene &sc, const RenderJob &job, float *out, hipStream_t stream, uint32_t n_cus,
                              const char **variant);

// Primary-visibility pre-pass (hip/primary_kernel.hip): the closest hit of every camera ray of the job's pixels into
// `prehit` (2 words per (pixel, sample) of the WHOLE frame: width * height * spp * 2 words); the render kernels use it
// when DeviceScene::prehit points to it.  Not for scenes with opacity masks or the reference-order validation walk.
// (With split samples DeviceScene::prehit_step is the launch's sample step.)
bool PrimaryPrepassSupports(const DeviceScene &sc, const RenderJob &job);
hipError_t LaunchPrimaryPrepass(const DeviceScene &sc, const RenderJob &job, uint32_t *prehit, TraceCounters *counters,
                                hipStream_t stream, uint32_t n_cus);

// Tile hand-out order from the pre-pass (hip/tile_order.hip).  keys / sorted: n_tiles 64-bit words each; `temp`: scratch of
// TileOrderTempBytes(n_tiles) bytes.  After the call (asynchronous on `stream`) the low words of `sorted` are the job's
// local tiles, most expensive first (equal cost: image order).
size_t TileOrderTempBytes(uint32_t n_tiles);
hipError_t LaunchTileOrder(const DeviceScene &sc, const RenderJob &job, const uint32_t *prehit, unsigned long long *keys,
                           unsigned long long *sorted, void *temp, size_t temp_bytes, hipStream_t stream);

// frame[p] = (planes[0][p] + ... + planes[K-1][p]) / spp for p < n_pixels (3 floats each), planes K x plane_stride pixels.
hipError_t LaunchReduceSamplePlanes(const float *planes, float *frame, uint32_t n_pixels, uint32_t split, uint32_t plane_stride,
                                    float spp_inv, hipStream_t stream);

// Multi-GPU gather: scatters one rank's packed tiles (device memory) into a device frame.
hipError_t LaunchUnpackTiles(const float *packed, float *frame, uint32_t tile_first, uint32_t tile_stride, uint32_t n_tiles,
                             uint32_t tiles_x, uint32_t width, uint32_t height, hipStream_t stream);

// ---- multi-kernel wavefront formulation (hip/wavefront_kernels.hip) --------------------------------------
// One slot per item of the job, path state in HBM; a frame = rounds of (shade launch, trace launch) until a round lists
// no ray.  counters: WavefrontCounterWords() words, zero before the first round.  Buffers sized by WavefrontSizes (32-bit words).
bool WavefrontSupports(const DeviceScene &sc, const RenderJob &job);
void WavefrontSizes(const DeviceScene &sc, uint32_t n_slots, size_t *cold_words, size_t *hot_words, size_t *id_words);
uint32_t WavefrontCounterWords(); // two round parities x two ray kinds x the queues; the last round's half sums to 0 when the frame is done
hipError_t LaunchWavefrontRound(const DeviceScene &sc, const RenderJob &job, float *out, uint32_t *cold, uint32_t *hot, uint32_t *ids,
                                uint32_t *counters, uint32_t n_slots, bool first_round, uint32_t parity, hipStream_t stream);

// ---- queued renderer (queue_core.h, hip/queued_kernels.*) ------------------------------------------------
// Path slots (a pool: a finished slot takes the next unassigned pixel), ray queues and per-material shade queues in
// HBM; a frame = round 0 (every slot starts) + rounds of (trace launch, one shade launch per material group) until a
// round queues nothing.  The caller owns ONE buffer of QueuedSizes::total_words() words, zeroes the counter block
// (QueuedCounters) and the job's work counter before round 0, and reads the counter block to find out when the frame
// is finished: the frame is done when the block of parity (last round + 1) & 1 sums to zero.
struct QueuedSizes
{
    uint32_t cap, n_slots, n_present;
    size_t slot_words, ext_words, shadow_words, entry_words, counter_words, spill_words;
    size_t total_words() const { return slot_words + ext_words + shadow_words + entry_words + counter_words + spill_words; }
};
bool QueuedSupports(const DeviceScene &sc, const RenderJob &job);
uint32_t QueuedGroups(const BsdfRec *bsdfs, size_t n_bsdfs, bool any_instance_without_bsdf); // bit g: the scene needs group g's launch
void QueuedLayout(uint32_t n_slots_wanted, uint32_t groups, uint32_t walk_depth, uint32_t n_cus, QueuedSizes *sizes);
uint32_t QueuedTraceBlocks(uint32_t n_cus);
uint32_t *QueuedCounters(uint32_t *base, const QueuedSizes &sizes);
hipError_t LaunchQueuedRound(const DeviceScene &sc, const RenderJob &job, float *out, uint32_t *base, const QueuedSizes &sizes, uint32_t groups,
                             uint32_t round, uint32_t n_cus, hipStream_t stream);

// Experiment (hip/trace_rate_kernel.hip): closest-hit rate of a lean trace-only kernel on a batch of rays in HBM.
hipError_t RunTraceRate(const DeviceScene &sc, uint32_t n, const float *rays_dev, int mode, int waves, uint32_t refill_at, uint32_t n_cus,
                        uint32_t *found_dev, float *milliseconds, hipStream_t stream);

// Unit kernels for diagnostics and parity tests (one query per lane).
// pool_walk: through the wavefront-cooperative pool walk (32-bit items, quadrics, sliver rules) instead of walk_ordered.
hipError_t LaunchIntersect(const DeviceScene &sc, uint32_t n, const float *rays, const uint32_t *seeds, float *out,
                           uint32_t *seeds_out, bool reference_walk, hipStream_t stream, bool pool_walk = false);
hipError_t LaunchTracePixel(const DeviceScene &sc, uint32_t pixel, uint32_t capacity, float *out, uint32_t *n_steps,
                            hipStream_t stream);
hipError_t LaunchBsdf(const DeviceScene &sc, uint32_t n, uint32_t id_bsdf, int mode, const float *recs,
                      const uint32_t *seeds, float *out, uint32_t *seeds_out, hipStream_t stream);

} // namespace mcpt

#endif // MCPT_RENDER_KERNEL_H
