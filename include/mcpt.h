/* mcpt — C ABI of the MI355X path tracer (libmcpt_hip.so).
 *
 * Drop-in boundary for the render path of zhiwei-c/Monte-Carlo-Path-Tracing.
 * The reference has no FFI; its seam is the C++ class `csrt::Renderer`
 * (reference include/csrt/renderer/renderer.hpp:30-81) constructed from a
 * `csrt::RendererConfig` (renderer.hpp:18-28) and driven by
 * `csrt::RayTracer` (src/ray_tracer.cpp:124-159) from `main`
 * (apps/main.cpp:25-95).  Each entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no exceptions cross
 * this boundary: every call returns 0 on success or a non-zero status, and
 * `mcpt_last_error()` returns the message (the reference throws
 * csrt::MyException with chained text, include/csrt/utils/misc.hpp:52-62).
 *
 * Threading: calls on one renderer must be serialised by the caller (the
 * reference's Renderer is not reentrant either, renderer.cpp:17-22); different
 * renderers may be used from different threads.
 */
#ifndef MCPT_H
#define MCPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- renderer configuration (replaces csrt::RendererConfig) -------------- */

typedef struct mcpt_config mcpt_config;

/* Replaces csrt::LoadConfig(filename) (reference src/parser/parser.cpp:94-179)
 * for scenes already converted to MCSD (include/mcsd_format.h). */
int mcpt_config_load_mcsd(const char *path, mcpt_config **out);
int mcpt_config_from_mcsd_bytes(const void *bytes, size_t size, mcpt_config **out);

/* Replaces csrt::LoadConfig for Mitsuba-style XML scene files
 * (parser.cpp:94-1617 + model_loader.cpp).  Supported subset: see DESIGN.md. */
int mcpt_config_load_xml(const char *path, mcpt_config **out);

/* The same, for a scene whose file names mesh files that are not on disk (the reference repository ships
 * resources/scene/dragon/scene.xml without four of its OBJ files, /root/reference/.MISSING_LARGE_BLOBS).
 * `standins` is a text table, one line per missing file: the file name as written in the XML followed by a
 * procedural mesh description ("blob ..." / "sheet ...", see csrc/host/standin_mesh.cpp).  A file that IS on
 * disk is always read; a missing file without a line is the reference's error (model_loader.cpp:440-447). */
int mcpt_config_load_xml_with_standins(const char *path, const char *standins, mcpt_config **out);
/* Replaces the mesh of triangle-mesh instance `instance` (scene order) by the stand-in one table line
 * describes; BSDF, media and to_world stay.  For configurations stored as MCSD with small placeholders where
 * the large stand-ins go (tests/golden/dragon_real_meshes.mcsd): the result equals loading the XML with the
 * same table. */
int mcpt_config_set_instance_standin(mcpt_config *cfg, uint32_t instance, const char *standin_line);

/* Built-in scenes ("cornell-box"): the same records the XML front end produces
 * for the reference's example scenes, available without any scene file. */
int mcpt_config_builtin(const char *name, mcpt_config **out);

/* Replaces the CLI overrides of apps/main.cpp:46-52 (`-w -h -s`): values <= 0
 * keep the scene's own.  As in the reference, fov_x is NOT recomputed. */
int mcpt_config_set_film(mcpt_config *cfg, int width, int height, int spp);
int mcpt_config_get_film(const mcpt_config *cfg, int *width, int *height, int *spp);

int mcpt_config_save_mcsd(const mcpt_config *cfg, const char *path);
/* The same bytes into the caller's buffer; buffer == NULL only reports the size.  (How a configuration travels to
 * libmcpt_host.so, include/mcpt_host.h.) */
int mcpt_config_serialize(const mcpt_config *cfg, void *buffer, size_t capacity, size_t *size);
void mcpt_config_destroy(mcpt_config *cfg);

/* ---- renderer (replaces csrt::Renderer) ---------------------------------- */

typedef struct mcpt_renderer mcpt_renderer;

/* Which pixels a draw call produces: 8x8 pixel tiles in row-major tile order,
 * tile index t = tile_first + k * tile_stride for k in [0, tile_count).
 * tile_count == 0 means "every tile from tile_first in steps of tile_stride".
 * {0, 1, 0} is the whole frame; rank r of N GPUs uses {r, N, 0}. */
typedef struct mcpt_tile_range
{
    uint32_t tile_first, tile_stride, tile_count;
} mcpt_tile_range;

typedef struct mcpt_stats
{
    double render_seconds;      /* wall time of the draw call (kernel + copies) */
    double kernel_milliseconds; /* HIP-event time of the render kernel on its stream */
    uint64_t samples;           /* pixels * spp produced by this call */
    /* filled only by mcpt_renderer_draw_counted(): */
    uint64_t closest_rays, shadow_rays, node_tests, prim_tests, shaded_hits;
    /* ordered walk: wavefront-level steps of the node and primitive phases (each step
     * runs for all 64 lanes whether or not they have work): lane utilisation of the
     * node phase = (node_tests / 2) / (64 * wave_node_steps) */
    uint64_t wave_node_steps, wave_prim_steps;
    /* stream kernel, counting mode: shader-clock ticks summed over wavefronts — in the shade phase, in the
     * trace phase, waiting at the workgroup barriers between them — and the number of rounds (summed over
     * workgroups).  0 for the lane-owns-a-path kernel. */
    uint64_t ticks_shade, ticks_trace, ticks_wait, rounds;
} mcpt_stats;

/* Replaces Renderer::Renderer(const RendererConfig&) (reference
 * src/renderer/renderer.cpp:259-348 incl. Scene::Scene, scene.cpp:118-141):
 * commits the configuration into flat tables, builds the two-level LBVH and
 * uploads everything to HBM of `device` (HIP device ordinal). */
int mcpt_renderer_create(const mcpt_config *cfg, int device, mcpt_renderer **out);

/* Replaces Renderer::Draw(float *frame) (renderer.cpp:678-721): blocking; fills
 * the caller's HOST buffer of width*height*3 float32 (row 0 = top, linear RGB,
 * mean over spp of per-sample-clamped radiance).  stats may be NULL. */
int mcpt_renderer_draw(mcpt_renderer *r, float *frame, mcpt_stats *stats);

/* Device-resident variant for multi-GPU tiling and benchmarking: renders the
 * tiles of `range` into a DEVICE buffer on `stream` (a hipStream_t, NULL = the
 * default stream) and returns after enqueueing when `blocking` is 0.
 *   packed == 0: out is a full frame (width*height*3 floats), only this
 *                range's pixels are written;
 *   packed != 0: out holds the range's tiles back to back, 64 pixels * 3
 *                floats per tile (pixels outside the image are left untouched).
 * The FIRST draw of a tile range waits on `stream` once or twice even with blocking == 0: it reads the range's statistics
 * (camera-ray hits of the pre-pass, the 2-spp cost probe: mcpt_renderer_set_tile_order, _set_stream_waves) to the host, lays
 * the tiles out and keeps the result; later draws of the same range only enqueue.
 * The draws of ONE renderer are ordered — enqueue them on one stream at a time: its work counter, the records of the launch in
 * flight and its path market belong to that launch.  Different renderers may draw at the same time, also on one device. */
int mcpt_renderer_draw_device(mcpt_renderer *r, float *out_device, const mcpt_tile_range *range,
                              int packed, void *stream, int blocking, mcpt_stats *stats);

/* Same as mcpt_renderer_draw but with the in-kernel counting mode on: ray,
 * node-test and primitive-test totals for the roofline bookkeeping
 * (SURVEY.md §8d).  Slower; the image is identical. */
int mcpt_renderer_draw_counted(mcpt_renderer *r, float *frame, mcpt_stats *stats);

/* Number of 8x8 tiles of the frame, and how many of them `range` selects. */
int mcpt_renderer_tile_count(const mcpt_renderer *r, uint32_t *tiles_total);
uint32_t mcpt_tile_range_size(uint32_t tiles_total, const mcpt_tile_range *range);

/* Scatter packed tiles (host memory) back into a host frame. */
int mcpt_unpack_tiles(const float *packed, const mcpt_tile_range *range, int width, int height, float *frame);

/* Committed-table inspection (tests, tools).  `what`: "nodes" (float, 8 per
 * node), "node_area", "walk_nodes" (16 per node), "walk_prims" (12 per slot),
 * "tri_pos" (float, 12 per primitive), "tri_attr" (36 per primitive),
 * "lut_brdf", "lut_albedo", "light_cdf", "env_tables", "camera" (12).  Returns a
 * pointer into renderer-owned HOST memory, valid until the renderer dies.
 * "market": 3 floats read back from the device after the last draw has finished (the call waits for it) — tickets taken by
 * waiting wavefronts, path records given away, items of the job finished: what the kernels outside LDS did with the frame's last
 * paths (csrc/hip/render_kernel_impl.h, "PATH MARKET"); zeros for a renderer whose kernels have none.  Thread-local storage. */
int mcpt_renderer_table(const mcpt_renderer *r, const char *what, const void **data, size_t *count);

/* Scene statistics (9 x uint64): nodes and TLAS nodes of the reference-topology
 * trees, primitives, instances, feature bits, bytes of geometry in HBM, nodes and
 * depth of the ordered-walk hierarchy, whether opacity masks are present. */
int mcpt_renderer_info(const mcpt_renderer *r, uint64_t info[9]);

/* Which ray query later draws use.  0 (default): near-first walk of the SAH
 * hierarchy — same image as the reference's traversal (reference
 * src/rtcore/accel/tlas.cpp:22-41, blas.cpp:26-43), fewer node visits.
 * 1: the reference's own trees in the reference's visiting order (validation;
 * always used when a BSDF has an opacity map, whose test draws random numbers
 * during the walk). */
int mcpt_renderer_set_walk(mcpt_renderer *r, int reference_order);
/* Which one later draws use — 1 also when mcpt_renderer_create's self-check found a pixel on which the two walks differ and
 * fell back to the reference's order (a line on stderr says so): the production walk returns the reference's answers as long
 * as near-coincident hits lie within its tie radius, an engineering bound; every new renderer renders a sample of its film
 * (every k-th 8 x 8 tile, at most 65 536 pixels, 1 spp) with both walks and compares.  Environment MCPT_CHECK_WALKS: 0 = no
 * check, N > 0 = the whole film at N spp instead of the sample. */
int mcpt_renderer_get_walk(const mcpt_renderer *r, int *reference_order);

/* Self-check of the production ray query on THIS scene and film: renders the frame twice — ordered walk of the
 * SAH hierarchy (what draws use) and the reference's trees in the reference's order — and compares them bit for
 * bit.  The ordered walk's tie / sliver radii (csrc/host/commit.cpp) are bounds on the rounding error of the
 * watertight triangle test; a scene outside them would show up here as n_differing > 0 instead of silently
 * leaving the reference image.  first_pixel (y * width + x, 0xFFFFFFFF if none) and max_abs_diff may be NULL.
 * Costs two draws (the reference-order one is several times slower).  No reference counterpart. */
int mcpt_renderer_check_walks(mcpt_renderer *r, uint64_t *n_differing, uint32_t *first_pixel, float *max_abs_diff);

/* Tuning knob of the vote-scheduled ordered walk (scenes of >= 2048 primitives; the image does not
 * depend on it): a wavefront leaves its box-test phase for the primitive tests as soon as fewer than
 * `leave_below` lanes are still searching, or as soon as `leave_at` lanes hold a primitive (0 = never
 * for that reason).  The commit's defaults are 8 / 12 (0 / 0 for small scenes).  No reference
 * counterpart. */
int mcpt_renderer_set_walk_schedule(mcpt_renderer *r, uint32_t leave_below, uint32_t leave_at);

/* Which kernel formulation later draws use; the image does not depend on it (tests assert the frames are
 * bit-identical).  Replaces the reference's megakernel dispatch (src/renderer/renderer.cpp:88-95).
 *   mode -1 (default): the lane-owns-a-path kernel for the few-KB scenes whose traversal data sits in LDS
 *          (cornell-box, volumetric-caustic: it is VALU-bound there and faster); for every other scene the stream kernel
 *          in wavefront rounds with the work counter (mode 4) — the built-in rule, no measurement inside a draw — UNLESS a
 *          calibrated choice for this scene, film and device is known: mcpt_renderer_calibrate (or MCPT_CALIBRATE=1 in
 *          the environment, for the first draw) times four configurations — lanes kernel with fixed lists / with the work
 *          counter, stream kernel in workgroup rounds / in wavefront rounds — on a sample of the frame's tiles at a few
 *          spp, keeps the fastest (they are within +-25 % of each other: DESIGN.md section 3) and STORES it, in the
 *          process and in the file MCPT_CALIBRATION_FILE (default $HOME/.cache/mcpt/calibration.txt, "off" = none), so
 *          that later renderers and later processes start with it.  mcpt_renderer_last_kernel reports the choice and,
 *          when it was measured, the four timings.
 *   mode 1: the STREAM kernel (csrc/stream_core.h) — a workgroup owns `slots` path slots (0 = built-in
 *          choice, otherwise a multiple of 256) whose rays go through a workgroup-local pool: emitted rays are
 *          compacted by wavefront ballot / prefix count, a lane that finishes a ray fetches the next one
 *          (`refill_at`: when that many lanes of a wavefront are free; 0 = built-in).  Scenes it does not cover
 *          (opacity masks, more than two shadow rays per vertex, the reference-order validation walk) fall back
 *          to mode 0.
 *   mode 2: the stream kernel with `slots` (> 256) slots per workgroup whose path state lives in memory instead of
 *          the lanes' registers (experiments; instantiated for few scene classes, otherwise falls back to mode 0).
 *   mode 3: the MULTI-KERNEL wavefront formulation (csrc/hip/wavefront_kernels.hip): one path slot per pixel of the draw in
 *          HBM, a frame = rounds of two launches — a shade kernel (one lane per slot; emitted rays compacted by wavefront
 *          ballot / prefix count into ray queues) and a lean trace kernel (one lane per queued ray, walk state only,
 *          2-3x the ray rate of the walk inside the single-kernel formulations) — until a round queues no ray.  Same
 *          frame.  Blocking (the host decides when to stop).  Instantiated for surface materials with one shadow ray per
 *          vertex (otherwise falls back); measured slower than mode 1 on the BASELINE scenes, so
 *          the library never picks it by itself.
 *   mode 4: mode 1 (one slot per lane) with WAVEFRONT rounds: each wavefront of a workgroup alternates its shade and
 *          trace phases alone — own ray list, no workgroup barrier — so no wavefront waits for the longest ray of the
 *          other three; its shadow rays fill only its own free lanes.  Faster than mode 1 on matpreview (+10 %) and the
 *          interior scenes (+9 .. +21 %), slower on dragon/scene.xml (-14 %): the fourth candidate of mode -1's
 *          calibration.  Same frame.
 *   mode 0: the lane-owns-a-path state machine (csrc/path_core.h, round 1's kernel).
 *   mode 5: the QUEUED renderer (csrc/queue_core.h, csrc/hip/queued_kernels.*): path slots as 96-byte records in a
 *          pool in HBM (`slots` = pool size in units of 4096 slots, 0 = one per pixel of the draw), a frame = rounds of
 *          one lean trace launch — it files every answered extension ray, with its hit, under the MATERIAL GROUP it hit
 *          (wavefront ballot + prefix count, one atomic per wavefront and group); an unoccluded shadow ray adds the
 *          direct light it carries to its slot — and one shade launch per BSDF kind present in the scene, each compiled
 *          with that kind's model only (no register spills) and fed 64 slots of that kind per wavefront.  Same frame.
 *          Surface paths on triangle meshes with one light sample per vertex (otherwise falls back).  Blocking.
 *          Measured slower than mode 4 on the BASELINE scenes (every round waits for the slowest ray of the whole GPU,
 *          and the per-pixel random stream makes a frame a chain of rounds: DESIGN.md section 3e), so the library never
 *          picks it by itself. */
int mcpt_renderer_set_kernel(mcpt_renderer *r, int mode, uint32_t slots, uint32_t refill_at);
/* Whether modes 1 - 5 above (and mcpt_debug_trace_rate) are part of this build.  Since round 5 the default build of the library
 * holds the lane-owns-a-path kernels only — the one formulation its rule chooses on every scene measured, by 1.4 - 4 x
 * (csrc/hip/formulations_not_built.hip) — and a request for another mode renders with them, like on any scene a mode does not
 * cover; `make EXPERIMENTAL=1` builds the others in.  No reference counterpart. */
int mcpt_build_has_formulations(void);
/* Times the kernel configurations on this renderer's scene and film now (blocking, eight sample launches) and stores
 * the winner for mode -1 — see mcpt_renderer_set_kernel.  No reference counterpart. */
int mcpt_renderer_calibrate(mcpt_renderer *r);
/* What the last draw actually ran, as arguments for mcpt_renderer_set_kernel / _set_work_distribution / _set_prepass
 * (so that a second renderer, a profiler run, ... can repeat the calibrated choice without calibrating). */
int mcpt_renderer_last_choice(const mcpt_renderer *r, int *kernel, int *work_distribution, int *prepass);

/* How the pixels of a draw reach the lanes; the image does not depend on it.  0: every lane walks a fixed list
 * (item q, q + launched lanes, ...).  1: a work counter in HBM — a lane that has finished a pixel takes the next
 * item nobody has taken yet (one atomic per wavefront and fetch), in image (tile) order.  Pixels cost very
 * different amounts (a camera ray that leaves the scene against a 17-bounce path through glass), and with fixed
 * lists a frame lasts as long as the unluckiest wavefront's list: dragon/scene.xml 698 -> 969, matpreview 363 ->
 * 491 / 231 -> 338, volumetric-caustic 767 -> 855 Msamples/s at full size.  -1 (default): 1 for the scenes whose
 * hierarchy sits in LDS; for the others part of the first draw's calibration (see mcpt_renderer_set_kernel) —
 * the reference's dining-room is 15 % faster with fixed lists.  No reference counterpart (its CPU back end hands
 * out 64-pixel patches dynamically, renderer.cpp:688-699; its CUDA back end launches one thread per pixel). */
int mcpt_renderer_set_work_distribution(mcpt_renderer *r, int mode);

/* Order in which the work counter hands out the tiles of a draw; the image does not depend on it.  A pixel is a
 * sequential chain of rounds (one random stream through all its samples, renderer.cpp:62-81), so a frame ends with the
 * chains that started last.  1: tiles most expensive first.  Their cost is MEASURED: the first draw of a tile range runs a
 * 2-spp probe in which the lanes kernel counts the steps of every tile (jobs where at least half of the camera rays hit
 * something; others keep image order); without the probe (MCPT_COST_ORDER=0) an estimate from what the pre-pass's camera
 * rays hit, weighted by BSDF kind (csrc/hip/tile_order.hip).  0: image order.
 * -1 (default): 1 whenever a draw runs the pre-pass and the work counter.
 * Scenes whose traversal data sits in LDS (no pre-pass there), draws that give every resident lane at most one pixel and
 * fill at least half of the lanes (cornell-box 512 x 512 on one MI355X): with -1 / 1 the first draw of a tile range
 * measures the tiles with a 2-spp probe (steps per tile), a wavefront then renders one tile, and the tiles are laid over the
 * grid so that every SIMD holds a wavefront of each cost quarter and all SIMDs the same sum (csrc/capi.cpp,
 * CostOrderedTable): 60.8 -> 56.2 ms; an explicit mcpt_renderer_set_pixel_order wins.  With more pixels than lanes
 * (volumetric-caustic) the same probe orders the work counter's hand-out, most expensive first (+1 %).  No reference counterpart.
 * 2 (round 6): image order in EIGHT BANDS, one per XCD of an MI355X — the film's hand-out positions are cut into eight contiguous
 * ranges with a counter each, and a workgroup takes from the band of the XCD it runs on while that has items, from the next bands
 * after that: each XCD's private L2 then serves one region of the film (pool-walk kernels outside LDS; EXPERIMENTS.md R6-5). */
int mcpt_renderer_set_tile_order(mcpt_renderer *r, int mode);

/* Class sort of the lane-owns-a-path kernel (csrc/hip/sorted_kernel.hip); the image does not depend on it.  Scenes whose
 * traversal data sits in LDS and that need more than the diffuse model (a participating medium, quadrics, microfacet
 * BSDFs): between resolve / roulette and connect / scatter, the paths of a 128-lane workgroup are counting-sorted by the
 * class of their vertex (10 classes: medium scattering event, surface without BSDF, the six BSDF kinds, finished sample,
 * exhausted lane) and their state (36 words) moves through LDS to its place in that order, so that a wavefront shades one
 * or two kinds of vertex instead of all of them.  The reference runs whatever each thread's path hit, diverged: the per-hit `switch` of
 * src/renderer/bsdfs/bsdf.cpp:188-211 inside the megakernel src/renderer/renderer.cpp:88-95.
 * mode -1 (default): on for those scenes; 0: off (the unsorted kernel); 1: same as -1. */
int mcpt_renderer_set_class_sort(mcpt_renderer *r, int mode);

/* Ray queries of the lane-owns-a-path kernel on LDS-resident scenes (csrc/pool_walk.h); the image does not depend on it.
 * 0: one walk per lane (walk_ordered) — a wavefront's query lasts as long as its slowest lane's walk, the shape of the
 * reference's per-thread nested stack walk (src/rtcore/accel/tlas.cpp:13-76, blas.cpp:18-77).  1: the wavefront-cooperative
 * pool walk — the rays of a wavefront's query are records in LDS and every lane takes the next (ray, node) or (ray,
 * primitive) item of a shared list, whichever ray it belongs to; closest hits are decided at the end among the candidates
 * within the tie radius of the nearest, in the reference's visiting order.  Scenes without slivers / opacity masks whose
 * hierarchy has at most 1024 nodes, on a 4-wide exact form of that hierarchy (DeviceScene::pool_nodes).  -1 (default): the
 * library's choice — on in the lean instantiations (cornell-box 512 x 512 spp 256: 55.2 -> 41.6 ms; a rank's 1/8 share of it
 * 42.0 -> 27.2 ms: lanes without a path of their own work on their wavefront's rays), off in the class-sorted full-feature
 * ones (volumetric-caustic: 231 -> 241 ms); environment MCPT_POOL_WALK = 0 / 1 / 2 overrides the default.
 * 2: like 1, and the lean LDS-resident kernels (cornell-box's class) run with MERGED QUERIES as the kernels outside LDS do
 * (csrc/path_core.h, path_step_merged: a vertex's last shadow ray travels with the next segment's closest query) — same frame;
 * a full 512 x 512 frame is 27 % slower with it (three workgroups per CU instead of four), see EXPERIMENTS.md R6-1. */
int mcpt_renderer_set_pool_walk(mcpt_renderer *r, int mode);

/* Register budget of the stream kernel's instantiation on scenes outside LDS (surface materials): compiled for 4, 3 or 2
 * wavefronts per SIMD (128 / 168 / 256 VGPRs: 288 / 210 / 4 spilled registers).  The image does not depend on it.
 * -1 (default): the library's rule — scenes without a dielectric or thin dielectric run the instantiations compiled
 * without those models at 3 (matpreview rough conductor 990 -> 831 ms, dragon/scene.xml 185 -> 172 ms against the full set
 * at 4); scenes with one keep the full set, at 4 when throughput-bound (rough dielectric: 1268 ms at 4, 1287 at 3) and at
 * 2 when chain-bound (fewer pixels whose camera ray hits something than the GPU holds lanes; decided in the first draw of
 * a tile range, when the pre-pass's hit count is known).  No reference counterpart (its CUDA back end compiles one
 * megakernel at whatever the compiler allots). */
int mcpt_renderer_set_stream_waves(mcpt_renderer *r, int waves);

/* Primary-visibility pre-pass (csrc/hip/primary_kernel.hip): the camera ray of sample s of pixel p is a function of
 * (p, s) alone — stratified in x, van der Corput in y, no random number (reference src/renderer/renderer.cpp:68-76) —
 * so the closest hits of ALL camera rays of a draw are computed first by a lean kernel (one lane per (pixel, sample),
 * coherent wavefronts), and the render kernels start every sample at its first vertex.  Same image, bit for bit; the
 * first of the closest rays of every sample leaves the sequential per-pixel chain.  Costs 8 bytes of HBM per sample of
 * the frame.  mode -1 (default): on for scenes whose hierarchy does not sit in LDS, where the scene allows it (no
 * opacity masks, not the reference-order validation walk) — dragon/scene.xml 1.4x, the reference's
 * dining-room 1.07x, matpreview +-0; the LDS-resident scenes lose 4-5 % to it; 0: off; 1: on wherever allowed.  Replaces the camera-ray part of the first
 * Scene::Intersect of ShadePath (src/renderer/integrators/path.cpp:18-21). */
int mcpt_renderer_set_prepass(mcpt_renderer *r, int mode);

/* Which random streams later draws use.  No reference counterpart for modes 1 and 2.
 *   mode 0 (default): the reference's — Tea-seeded LCG per pixel, threaded through ALL samples of the pixel
 *          (reference src/renderer/renderer.cpp:62-81, include/csrt/utils/math.hpp:43-63).  Frames are the CPU
 *          reference's bit for bit; the samples of a pixel are inherently sequential, so a pixel is one lane's work.
 *   mode 1: THROUGHPUT mode, not per-pixel comparable with the reference: every (pixel, sample) starts its own
 *          stream from a PCG hash (RXS-M-XS) of (seed, pixel, sample index) — same pixel jitter, same generator
 *          step inside the sample, same estimator — so the samples of a pixel are independent and are spread over
 *          `sample_split` lanes (samples k, k + K, ...; 0 = 16 (pixel, sample-subset) items per resident lane, in
 *          powers of two up to 32); a second kernel adds the lanes' sums in lane order (deterministic).  What it
 *          buys: small films and small per-GPU tile shares still fill the GPU (strong scaling beyond one pixel per
 *          lane), and short items handed out by the work counter balance what whole-pixel chains cannot (whole
 *          frames on one GPU: cornell-box 64 -> 45 ms, dragon/scene.xml 200 -> 92 ms).  Graded by mean-square error
 *          against a converged image, not per pixel.  Both kernel formulations, with the camera-ray pre-pass; whole
 *          frames or packed tile ranges.
 *   mode 2: mode 1 with LOW-DISCREPANCY points: draw d of sample s of a pixel is an Owen-scrambled Sobol point — draws come
 *          in pairs (2k, 2k + 1), every pair is the first two Sobol dimensions (a (0,2)-sequence) over an Owen-shuffled
 *          sample index, each dimension scrambled with its own hash of (seed, pixel, k) ("padded" sequences; hash-based
 *          nested uniform scrambling), so every pair of consecutive draws of a pixel — a point on a light, a scattered
 *          direction — is stratified over any power-of-two run of its samples (csrc/vecmath.h ld_next).  Same pixel
 *          jitter (the reference's own radical inverse, include/csrt/utils/math.hpp:29-41), same estimator; RMSE at 256
 *          spp: 0.57 (dragon/scene.xml) ... 0.70 (cornell-box) x mode 1's, falling like N^-0.55 ... N^-0.62 instead of N^-0.5.  At most 8192 samples per
 *          pixel; runs in the lane-owns-a-path kernel (csrc/hip/render_variants_lowdisc.hip, full feature set), with
 *          `sample_split` like mode 1; no counting mode.  Pinned bit for bit against the same kernel body compiled for
 *          the host (tests/emu). */
int mcpt_renderer_set_rng(mcpt_renderer *r, int mode, uint32_t seed, uint32_t sample_split);

/* Which pixels the 64 lanes of a wavefront of the lane-owns-a-path kernel render.  0: one 8x8 tile (neighbouring camera
 * rays).  1: transposed — 64 different tiles, one pixel each.  -1 (default): transposed when the scene's traversal data
 * sits in LDS and the draw gives every lane at most one pixel — then nothing can be re-balanced while the frame runs, it
 * lasts as long as its slowest wavefront, and a wavefront that holds an average mix of pixels (and thins out as its cheap
 * ones finish) is faster than one that holds a tile of expensive ones: cornell-box 512x512 spp 256 65.3 -> 61.0 ms.
 * Same frame. */
int mcpt_renderer_set_pixel_order(mcpt_renderer *r, int mode);

/* Small jobs (fewer pixels than the GPU holds lanes: a rank's share of a strong-scaling run, a thumbnail).  The
 * reference's one RNG stream per pixel makes a pixel's samples a sequential chain, so such a job lasts as long as one
 * chain however many lanes idle.  The launcher therefore SPREADS it: only every `lanes_per_path`-th lane of a wavefront
 * takes pixels, all wavefront slots of the GPU are used, and each wavefront executes the (fewer) diverged instructions
 * of fewer paths — the chain gets shorter.  0 (default): the launcher's choice (the sparsest power of two up to 8 that
 * still fits the resident lanes; 1 for jobs that fill the GPU); 1: dense; 2 .. 64: as given.  Same frame. */
int mcpt_renderer_set_lane_spread(mcpt_renderer *r, uint32_t lanes_per_path);

/* Name of the kernel instantiation the last draw launched ("" before the first draw); renderer-owned string. */
const char *mcpt_renderer_last_kernel(const mcpt_renderer *r);

/* ---- one frame over the GPUs of a node (replaces csrt::Renderer for N devices) ---- */

/* The reference's Draw is one blocking call that returns the whole frame (renderer.cpp:678-721); this is the
 * same call over `n_devices` HIP devices of one process (`devices` = their ordinals, NULL = 0 .. n-1): the
 * configuration is committed ONCE and uploaded to every device (one host thread per GPU), rank k of N renders
 * the 8x8 tiles k, k + N, ... into a packed block in its own HBM, ONE grouped ncclSend / ncclRecv gather brings
 * the blocks to devices[0] over xGMI (librccl.so.1 is opened at first use; not needed for single-GPU renderers),
 * devices[0] scatters them into the frame and copies it to the caller's host buffer.  The frame is bit-identical
 * for every N (pixels are independent: one RNG stream per pixel, renderer.cpp:62-66).
 * flags: MCPT_TILED_ALWAYS_GATHER takes the RCCL route even with one device (tests on a 1-GPU box). */
typedef struct mcpt_tiled_renderer mcpt_tiled_renderer;
#define MCPT_TILED_ALWAYS_GATHER 1u
/* Test switch: `devices` may list one device several times — N logical ranks (a renderer, a stream and a packed
 * block each) on fewer GPUs.  Real RCCL refuses such a communicator; the environment variable MCPT_RCCL_LIBRARY
 * names a library to bind in its place (tests/rccl_shim: the seven entry points over hipMemcpyAsync). */
#define MCPT_TILED_LOGICAL_RANKS 2u
int mcpt_tiled_renderer_create(const mcpt_config *cfg, int n_devices, const int *devices, unsigned flags,
                               mcpt_tiled_renderer **out);
/* Blocking; `frame` is a HOST buffer of width*height*3 floats.  stats (may be NULL): wall time of the call, the
 * slowest GPU's kernel time, samples of the whole frame. */
int mcpt_tiled_renderer_draw(mcpt_tiled_renderer *r, float *frame, mcpt_stats *stats);
/* mcpt_renderer_set_kernel for every rank. */
int mcpt_tiled_renderer_set_kernel(mcpt_tiled_renderer *r, int mode, uint32_t slots, uint32_t refill_at);
void mcpt_tiled_renderer_destroy(mcpt_tiled_renderer *r);
/* create + draw + destroy. */
int mcpt_render_tiled(const mcpt_config *cfg, int n_devices, const int *devices, float *frame, mcpt_stats *stats);
/* HIP devices visible to this process (0 without a GPU; never fails for that reason). */
int mcpt_device_count(int *n_devices);

/* The reference-topology LBVH (reference src/rtcore/accel/bvh_builder.cpp:74-207) of n
 * boxes (6 floats each: lo.xyz, hi.xyz) with areas, built by the host builder
 * (on_device == 0, no GPU needed) or by the HIP builder.
 * Output, 2n-1 nodes in pre-order: links[2k] = skip, links[2k+1] = object (box index
 * or 0xFFFFFFFF), geom[7k] = area, geom[7k+1..3] = lo, geom[7k+4..6] = hi.
 * seconds (may be NULL): build time; for the device builder it includes the copies
 * to and from the GPU. */
/* Diagnostics / tests: the hand-out table the renderer lays out from probed tile costs (csrc/capi.cpp, CostOrderedTable):
 * table[g] = the tile wavefront slot g of a one-pixel-per-lane launch renders.  steps: n_tiles words; layout 0 = cost order
 * (most expensive first), 1 = the production layout (odd quarters reversed), 3 = greedy on the SIMDs' sums.  Host only. */
int mcpt_debug_cost_table(const uint32_t *steps, uint32_t n_tiles, uint32_t n_cus, int layout, uint32_t *table);

int mcpt_debug_lbvh_build(uint32_t n, const float *boxes, const float *areas, int on_device, uint32_t *links,
                          float *geom, double *seconds);

/* Unit-level GPU queries for diagnostics and parity tests (host pointers in and
 * out; no reference counterpart — the reference has no tests).
 *   intersect: rays = origin[3] dir[3] per query; out = 19 floats per query:
 *     valid, inside, instance, primitive, t, uv[2], position[3], normal[3],
 *     tangent[3], bitangent[3] (closest hit incl. bump / back-face handling).
 *   bsdf: records = wo[3] wi[3] normal[3] tangent[3] bitangent[3] uv[2] inside
 *     (18 floats); mode 0 = evaluate, 1 = sample; out = valid, pdf,
 *     attenuation[3], wi[3].  seeds / seeds_out: the LCG state before / after. */
int mcpt_debug_intersect(mcpt_renderer *r, uint32_t n, const float *rays, const uint32_t *seeds, float *out,
                         uint32_t *seeds_out);
int mcpt_debug_bsdf(mcpt_renderer *r, uint32_t id_bsdf, int mode, uint32_t n, const float *records,
                    const uint32_t *seeds, float *out, uint32_t *seeds_out);

/* Experiment: closest-hit rate of a LEAN trace-only kernel (walk state only; csrc/hip/trace_rate_kernel.hip) on n rays
 * (origin[3] dir[3], host memory).  mode 0: one ray per lane; mode 1: persistent wavefronts that re-fill free lanes
 * from a global queue (refill_at lanes free, 0 = 16).  waves_per_simd: 4 or 8 (launch bound).  found[i] = primitive
 * hit (0xFFFFFFFF: none); milliseconds = the trace kernel alone.  Triangle scenes without opacity masks.  It bounds
 * what a multi-kernel wavefront formulation could gain; no render path uses it. */
int mcpt_debug_trace_rate(mcpt_renderer *r, uint32_t n, const float *rays, int mode, int waves_per_simd, uint32_t refill_at,
                          uint32_t *found, float *milliseconds);

/* Diagnostics: the steps of one pixel (index y * width + x) on the device, one lane, all of the
 * pixel's samples: 16 floats per step = ray origin[3], direction[3], largest throughput
 * component after the step, primitive hit (-1: none), distance, shadow queries, last shadow
 * result, LCG state after the step (bit pattern), radiance accumulated so far [3], depth.
 * n_steps[0] = steps written (<= capacity).  No reference counterpart. */
int mcpt_debug_trace_pixel(mcpt_renderer *r, uint32_t pixel, uint32_t capacity, float *out, uint32_t *n_steps);

/* Replaces Renderer::~Renderer / ReleaseData (renderer.cpp:350-369). */
void mcpt_renderer_destroy(mcpt_renderer *r);

/* ---- image output (replaces image_io::Write, src/utils/image_io.cpp:25-53) -- */

/* Writes `frame` (width*height*3 float32) by file suffix: ".png" (sRGB 8-bit,
 * the reference's transfer curve), ".exr" (linear float32 scanline OpenEXR),
 * ".pfm", ".f32" (raw). */
int mcpt_write_image(const char *path, const float *frame, int width, int height);

/* Thread-local message of the last failed call on this thread. */
const char *mcpt_last_error(void);

/* Library / device identification string, e.g. "mcpt 0.1 hip gfx950". */
const char *mcpt_version(void);

#ifdef __cplusplus
}
#endif

#endif /* MCPT_H */
