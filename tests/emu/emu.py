"""TEST INFRASTRUCTURE — ctypes loader of tests/emu/libmcpt_emu.so (host build
of the kernel body; see emulator.cpp)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmcpt_emu.so")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


SO_LD = os.path.join(HERE, "libmcpt_emu_ld.so")


class Emulator:
    def __init__(self, low_discrepancy=False):
        """low_discrepancy: the same source compiled with MCPT_LOW_DISCREPANCY (a draw of the random stream returns an
        Owen-scrambled Sobol point: csrc/vecmath.h) — the host twin of mcpt_renderer_set_rng mode 2."""
        subprocess.run(["make", "-s", "-C", HERE], check=True)
        lib = ctypes.CDLL(SO_LD if low_discrepancy else SO)
        assert bool(lib.mcpt_emu_low_discrepancy()) == bool(low_discrepancy)
        lib.mcpt_emu_last_error.restype = ctypes.c_char_p
        lib.mcpt_emu_render.restype = ctypes.c_int
        lib.mcpt_emu_render.argtypes = [ctypes.c_char_p, _f32p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.POINTER(ctypes.c_uint32)]
        lib.mcpt_emu_nodes.restype = ctypes.c_int
        lib.mcpt_emu_nodes.argtypes = [ctypes.c_char_p, _u32p, _f32p, ctypes.c_uint32]
        lib.mcpt_emu_wave_model.restype = ctypes.c_int
        lib.mcpt_emu_wave_model.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        lib.mcpt_emu_pool_model.restype = ctypes.c_int
        lib.mcpt_emu_pool_model.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        lib.mcpt_emu_walk.restype = ctypes.c_int
        lib.mcpt_emu_walk.argtypes = [ctypes.c_char_p, _f32p, _f32p, ctypes.c_uint32, ctypes.c_uint32, _u32p]
        self.lib = lib

    def set_rng(self, independent, seed=0):
        """mcpt_emu_render with one independent stream per (pixel, sample), like mcpt_renderer_set_rng modes 1 / 2."""
        self.lib.mcpt_emu_set_rng(int(bool(independent)), ctypes.c_uint32(seed))
        return self

    def draws(self, word, n):
        """n successive draws of this build's generator from the state `word`."""
        self.lib.mcpt_emu_draws.argtypes = [ctypes.c_uint32, ctypes.c_uint32, _f32p]
        out = np.zeros(n, dtype=np.float32)
        self.lib.mcpt_emu_draws(word, n, out)
        return out

    def ld_pack(self, sample, seed, pixel):
        self.lib.mcpt_emu_ld_pack.restype = ctypes.c_uint32
        self.lib.mcpt_emu_ld_pack.argtypes = [ctypes.c_uint32] * 3
        return int(self.lib.mcpt_emu_ld_pack(sample, seed, pixel))

    def render(self, mcsd_path, width, height, variant=-1, counted=False):
        frame = np.zeros((height, width, 3), dtype=np.float32)
        counters = np.zeros(6, dtype=np.uint32)
        feat = ctypes.c_uint32()
        rc = self.lib.mcpt_emu_render(str(mcsd_path).encode(), frame, variant,
                                      counters.ctypes.data if counted else None, ctypes.byref(feat))
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        info = {"features": feat.value}
        if counted:
            info.update(zip(("closest_rays", "shadow_rays", "node_tests", "prim_tests", "shaded_hits", "samples"),
                            (int(c) for c in counters)))
        return frame, info

    def render_stream(self, mcsd_path, width, height, slots_per_block=0, counted=False):
        """The stream formulation (csrc/stream_core.h) on the host."""
        self.lib.mcpt_emu_render_stream.restype = ctypes.c_int
        self.lib.mcpt_emu_render_stream.argtypes = [ctypes.c_char_p, _f32p, ctypes.c_uint32, ctypes.c_void_p]
        frame = np.zeros((height, width, 3), dtype=np.float32)
        counters = np.zeros(6, dtype=np.uint32)
        rc = self.lib.mcpt_emu_render_stream(str(mcsd_path).encode(), frame, slots_per_block,
                                             counters.ctypes.data if counted else None)
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        info = dict(zip(("closest_rays", "shadow_rays", "node_tests", "prim_tests", "shaded_hits", "samples"),
                        (int(c) for c in counters))) if counted else {}
        return frame, info

    def render_queued(self, mcsd_path, width, height, n_slots=0):
        """The queued formulation (csrc/queue_core.h) on the host: (frame, rounds)."""
        self.lib.mcpt_emu_render_queued.restype = ctypes.c_int
        self.lib.mcpt_emu_render_queued.argtypes = [ctypes.c_char_p, _f32p, ctypes.c_uint32, ctypes.c_void_p]
        frame = np.zeros((height, width, 3), dtype=np.float32)
        rounds = ctypes.c_uint32()
        if self.lib.mcpt_emu_render_queued(str(mcsd_path).encode(), frame, n_slots, ctypes.byref(rounds)) != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        return frame, rounds.value

    ORDERED = 32      # feature bit of the ordered walk in a forced `variant`
    WIDE = 256        # ... of the 4-wide quantised hierarchy (with ORDERED)
    REFERENCE = -2    # `variant`: the launcher's pick, but with the reference-order walk

    def trace_pixel(self, mcsd_path, x, y, width, ordered=True, capacity=4096):
        """Per-step records of one pixel (steps[n, 16], lcg[n]): the CPU twin of
        capi.Renderer.trace_pixel / mcpt_debug_trace_pixel."""
        self.lib.mcpt_emu_debug_pixel.restype = ctypes.c_int
        self.lib.mcpt_emu_debug_pixel.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p,
                                                   ctypes.c_uint32]
        out = np.zeros((capacity, 16), np.float32)
        n = self.lib.mcpt_emu_debug_pixel(str(mcsd_path).encode(), y * width + x, 1 if ordered else 0, out.ctypes.data,
                                          capacity)
        if n < 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        steps = out[:n]
        return steps, steps[:, 11].copy().view(np.uint32)

    def closest(self, mcsd_path, rays, ordered=True):
        """Closest-hit queries (rays[n, 6]) with the ordered or the reference-order walk:
        (primitive[n] int64 with -1 = miss, distance[n] float32)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 2), np.float32)
        self.lib.mcpt_emu_closest.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        # ordered: False / 0 = the reference's trees in the reference's order, True / 1 = the ordered walk of the binary
        # hierarchy, 2 = the 4-wide quantised hierarchy with the short stack (the production walk of scenes outside LDS)
        if self.lib.mcpt_emu_closest(str(mcsd_path).encode(), rays.ctypes.data, len(rays), int(ordered),
                                     out.ctypes.data) != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        return out[:, 0].astype(np.int64), out[:, 1].copy()

    def set_walk_tree(self, strategy: int):
        """Split rule of the ordered-walk hierarchy for later renders: 0 production,
        1 exact sweep, 2 median, 3 children swapped."""
        self.lib.mcpt_emu_set_walk_tree(int(strategy))

    def wave_model(self, mcsd_path):
        """Lock-step 64-lane model of the walks (see emulator.cpp): what a wavefront pays
        when the shadow walk runs on its own vs paired with the next closest walk."""
        out = (ctypes.c_double * 6)()
        if self.lib.mcpt_emu_wave_model(str(mcsd_path).encode(), out) != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        keys = ("lane_node_steps", "wave_node_steps_separate", "wave_node_steps_paired",
                "lane_prim_tests", "wave_prim_phases_separate", "wave_prim_phases_paired")
        return dict(zip(keys, out))

    def pool_model(self, mcsd_path):
        """Per-lane walk vs a wavefront-shared pool of (ray, node) items (see emulator.cpp)."""
        out = (ctypes.c_double * 24)()
        if self.lib.mcpt_emu_pool_model(str(mcsd_path).encode(), out) != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        keys = ["lane_wave_node_steps", "lane_node_visits", "lane_wave_prim_phases", "lane_prim_tests",
                "pool_node_steps", "pool_node_visits", "pool_prim_steps", "pool_prim_tests"]
        for spec in range(4):   # lock-step while-while with `spec` speculatively collected primitives
            keys += [f"spec{spec}_node_steps", f"spec{spec}_node_visits", f"spec{spec}_prim_steps", f"spec{spec}_prim_tests"]
        return dict(zip(keys, out))

    def walk(self, mcsd_path, capacity=1 << 21):
        """The ordered-walk hierarchy: (nodes[n, 4, 4] float32 with bit-pattern links,
        prims[m, 3, 4], info dict)."""
        nodes = np.zeros((capacity, 4, 4), dtype=np.float32)
        prims = np.zeros((capacity, 3, 4), dtype=np.float32)
        counts = np.zeros(8, dtype=np.uint32)
        rc = self.lib.mcpt_emu_walk(str(mcsd_path).encode(), nodes.reshape(-1), prims.reshape(-1), capacity, capacity,
                                    counts)
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode() if rc == -1 else "capacity too small")
        return nodes[:counts[0]], prims[:counts[1]], {"depth": int(counts[2]), "has_masks": bool(counts[3]),
                                                        "commit_ms": {"geometry+lbvh": int(counts[4]), "walk_tree": int(counts[5]),
                                                                      "total": int(counts[6])}}

    def pool_nodes(self, mcsd_path, capacity=1 << 21):
        """The 4-wide exact hierarchy of the pool walk: (planes[n, 6, 4] float32 = lo.xyz / hi.xyz of the four children,
        refs[n, 4] uint32, depth)."""
        self.lib.mcpt_emu_pool_nodes.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        nodes = np.zeros((capacity, 8, 4), dtype=np.float32)
        counts = np.zeros(2, dtype=np.uint32)
        rc = self.lib.mcpt_emu_pool_nodes(str(mcsd_path).encode(), nodes.ctypes.data, capacity, counts.ctypes.data)
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode() if rc == -1 else "capacity too small")
        nodes = nodes[:counts[0]]
        return nodes[:, :6, :].copy(), nodes[:, 6, :].copy().view(np.uint32), int(counts[1])

    def nodes(self, mcsd_path, capacity=1 << 22):
        links = np.zeros((capacity, 2), dtype=np.uint32)
        geom = np.zeros((capacity, 7), dtype=np.float32)
        n = self.lib.mcpt_emu_nodes(str(mcsd_path).encode(), links, geom, capacity)
        if n < 0:
            raise RuntimeError(self.lib.mcpt_emu_last_error().decode())
        return links[:n], geom[:n]
