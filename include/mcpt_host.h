/* mcpt_host — the render path's kernel body compiled for the host (libmcpt_host.so, optional).
 *
 * Replaces the reference's CPU back end behind `-c / --cpu` (reference apps/main.cpp:130-137,
 * src/renderer/renderer.cpp:678-721 with BackendType::kCpu): the SAME functions the HIP kernel runs
 * (csrc/path_core.h and below, csrc/host/commit.cpp), one pixel per task on a host thread pool.  It exists for
 * BASELINE config 1 ("cornell-box 512x512 spp=16 on the CPU path, plumbing") and for boxes without a GPU; it is
 * a separate shared object — libmcpt_hip.so itself has no CPU path and never loads this one — and it is not the
 * test oracle (oracle/ is an independent restatement used only by tests).
 */
#ifndef MCPT_HOST_H
#define MCPT_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Renders the configuration serialised in `mcsd_bytes` (mcpt_config_serialize, include/mcpt.h; or the contents
 * of a .mcsd file) into `frame` (width*height*3 float32, row 0 = top) on `threads` host threads (<= 0: all).
 * seconds (may be NULL): wall time of the pixel loop, commit excluded.  Returns 0 or non-zero with a message in
 * mcpt_host_last_error(). */
int mcpt_host_render(const void *mcsd_bytes, size_t size, int threads, float *frame, double *seconds);
/* The same for the 8x8 tiles tile_first, tile_first + tile_stride, ... (tile_count of them, 0 = all that exist;
 * mcpt_tile_range of include/mcpt.h): packed == 0 writes their pixels into a full frame, packed != 0 writes the
 * tiles back to back, 64 pixels x 3 floats each, pixels outside the image untouched — byte for byte what
 * mcpt_renderer_draw_device produces on the GPU for the same range. */
int mcpt_host_render_tiles(const void *mcsd_bytes, size_t size, int threads, uint32_t tile_first, uint32_t tile_stride,
                           uint32_t tile_count, int packed, float *frame, double *seconds);
const char *mcpt_host_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* MCPT_HOST_H */
