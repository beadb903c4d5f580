"""ctypes binding of the C ABI in include/mcpt.h (libmcpt_hip.so).

Mirrors the reference's host-side interface for the render path:
`Config` stands for csrt::RendererConfig (+ csrt::LoadConfig), `Renderer` for
csrt::Renderer with its `Draw(float *frame)`; errors surface as
`McptError` carrying the library's message (the reference throws
csrt::MyException).  The library renders on the GPU only: if the shared object
or a HIP device is missing, construction fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCPT_LIB") or os.path.join(HERE, "libmcpt_hip.so")   # MCPT_LIB: experiment builds
CSRC = os.path.join(HERE, "csrc")


class McptError(RuntimeError):
    pass


class TileRange(ctypes.Structure):
    """8x8-pixel tiles t = first + k * stride, k < count (count 0 = all)."""
    _fields_ = [("tile_first", ctypes.c_uint32), ("tile_stride", ctypes.c_uint32),
                ("tile_count", ctypes.c_uint32)]


class Stats(ctypes.Structure):
    _fields_ = [("render_seconds", ctypes.c_double), ("kernel_milliseconds", ctypes.c_double),
                ("samples", ctypes.c_uint64), ("closest_rays", ctypes.c_uint64),
                ("shadow_rays", ctypes.c_uint64), ("node_tests", ctypes.c_uint64),
                ("prim_tests", ctypes.c_uint64), ("shaded_hits", ctypes.c_uint64),
                ("wave_node_steps", ctypes.c_uint64), ("wave_prim_steps", ctypes.c_uint64),
                ("ticks_shade", ctypes.c_uint64), ("ticks_trace", ctypes.c_uint64),
                ("ticks_wait", ctypes.c_uint64), ("rounds", ctypes.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(quiet: bool = True) -> str:
    """Compile libmcpt_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-s", "-j8", "-C", CSRC], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


_lib = None


def has_formulations() -> bool:
    """Whether the stream kernel, the queued renderer, mode 3 and the trace-rate experiment are part of the library (make EXPERIMENTAL=1)."""
    return bool(lib().mcpt_build_has_formulations())


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McptError(f"{LIB_PATH} is not built (run __graft_entry__.build())")
    # PyTorch wheels bundle their own libamdhip64.so.7; a process must hold ONE
    # HIP runtime, otherwise the second one sees no devices and streams /
    # tensors cannot be shared.  Importing torch first lets the loader satisfy
    # this library's libamdhip64.so.7 dependency with the runtime torch loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, cp, i32, u32 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32
    L.mcpt_last_error.restype = cp
    L.mcpt_version.restype = cp
    for name in ("mcpt_config_load_mcsd", "mcpt_config_load_xml", "mcpt_config_builtin"):
        getattr(L, name).argtypes = [cp, ctypes.POINTER(vp)]
    L.mcpt_config_load_xml_with_standins.argtypes = [cp, cp, ctypes.POINTER(vp)]
    L.mcpt_config_set_instance_standin.argtypes = [vp, u32, cp]
    L.mcpt_config_from_mcsd_bytes.argtypes = [cp, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.mcpt_config_set_film.argtypes = [vp, i32, i32, i32]
    L.mcpt_config_get_film.argtypes = [vp] + [ctypes.POINTER(i32)] * 3
    L.mcpt_config_save_mcsd.argtypes = [vp, cp]
    L.mcpt_config_destroy.argtypes = [vp]
    L.mcpt_config_destroy.restype = None
    L.mcpt_renderer_create.argtypes = [vp, i32, ctypes.POINTER(vp)]
    L.mcpt_renderer_draw.argtypes = [vp, vp, ctypes.POINTER(Stats)]
    L.mcpt_renderer_draw_counted.argtypes = [vp, vp, ctypes.POINTER(Stats)]
    L.mcpt_renderer_draw_device.argtypes = [vp, vp, ctypes.POINTER(TileRange), i32, vp, i32,
                                            ctypes.POINTER(Stats)]
    L.mcpt_renderer_tile_count.argtypes = [vp, ctypes.POINTER(u32)]
    L.mcpt_tile_range_size.argtypes = [u32, ctypes.POINTER(TileRange)]
    L.mcpt_tile_range_size.restype = u32
    L.mcpt_unpack_tiles.argtypes = [vp, ctypes.POINTER(TileRange), i32, i32, vp]
    L.mcpt_renderer_table.argtypes = [vp, cp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    L.mcpt_renderer_info.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.mcpt_renderer_set_walk.argtypes = [vp, ctypes.c_int]
    L.mcpt_renderer_check_walks.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(u32),
                                            ctypes.POINTER(ctypes.c_float)]
    L.mcpt_renderer_set_walk_schedule.argtypes = [vp, u32, u32]
    L.mcpt_renderer_set_kernel.argtypes = [vp, i32, u32, u32]
    L.mcpt_renderer_set_rng.argtypes = [vp, i32, u32, u32]
    L.mcpt_renderer_set_prepass.argtypes = [vp, i32]
    L.mcpt_renderer_set_class_sort.argtypes = [vp, i32]
    L.mcpt_renderer_set_pool_walk.argtypes = [vp, i32]
    L.mcpt_renderer_get_walk.argtypes = [vp, ctypes.POINTER(i32)]
    L.mcpt_renderer_set_stream_waves.argtypes = [vp, i32]
    L.mcpt_renderer_set_lane_spread.argtypes = [vp, u32]
    L.mcpt_renderer_set_pixel_order.argtypes = [vp, i32]
    L.mcpt_renderer_set_work_distribution.argtypes = [vp, i32]
    L.mcpt_renderer_last_choice.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.mcpt_renderer_last_kernel.argtypes = [vp]
    L.mcpt_renderer_last_kernel.restype = cp
    L.mcpt_debug_cost_table.argtypes = [vp, u32, u32, i32, vp]
    L.mcpt_debug_lbvh_build.argtypes = [u32, vp, vp, i32, vp, vp, ctypes.POINTER(ctypes.c_double)]
    L.mcpt_debug_intersect.argtypes = [vp, u32, vp, vp, vp, vp]
    L.mcpt_debug_bsdf.argtypes = [vp, u32, i32, u32, vp, vp, vp, vp]
    L.mcpt_debug_trace_pixel.argtypes = [vp, u32, u32, vp, vp]
    L.mcpt_debug_trace_rate.argtypes = [vp, u32, vp, i32, i32, u32, vp, ctypes.POINTER(ctypes.c_float)]
    L.mcpt_renderer_destroy.argtypes = [vp]
    L.mcpt_renderer_destroy.restype = None
    L.mcpt_write_image.argtypes = [cp, vp, i32, i32]
    L.mcpt_config_serialize.argtypes = [vp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.mcpt_tiled_renderer_create.argtypes = [vp, i32, ctypes.POINTER(i32), ctypes.c_uint, ctypes.POINTER(vp)]
    L.mcpt_tiled_renderer_draw.argtypes = [vp, vp, ctypes.POINTER(Stats)]
    L.mcpt_tiled_renderer_set_kernel.argtypes = [vp, i32, u32, u32]
    L.mcpt_tiled_renderer_destroy.argtypes = [vp]
    L.mcpt_tiled_renderer_destroy.restype = None
    L.mcpt_render_tiled.argtypes = [vp, i32, ctypes.POINTER(i32), vp, ctypes.POINTER(Stats)]
    L.mcpt_device_count.argtypes = [ctypes.POINTER(i32)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise McptError(lib().mcpt_last_error().decode(errors="replace"))


EXPORTED_SYMBOLS = [
    "mcpt_config_load_mcsd", "mcpt_config_from_mcsd_bytes", "mcpt_config_load_xml",
    "mcpt_config_load_xml_with_standins", "mcpt_config_set_instance_standin", "mcpt_config_builtin", "mcpt_config_set_film", "mcpt_config_get_film",
    "mcpt_config_save_mcsd", "mcpt_config_destroy", "mcpt_renderer_create",
    "mcpt_renderer_draw", "mcpt_renderer_draw_device", "mcpt_renderer_draw_counted",
    "mcpt_renderer_tile_count", "mcpt_tile_range_size", "mcpt_unpack_tiles",
    "mcpt_renderer_table", "mcpt_renderer_info", "mcpt_renderer_set_walk", "mcpt_renderer_set_walk_schedule",
    "mcpt_renderer_set_kernel", "mcpt_renderer_last_kernel", "mcpt_renderer_check_walks", "mcpt_renderer_set_rng", "mcpt_renderer_set_prepass", "mcpt_renderer_set_lane_spread", "mcpt_renderer_set_pixel_order", "mcpt_renderer_set_work_distribution", "mcpt_renderer_last_choice",
    "mcpt_renderer_destroy",
    "mcpt_renderer_calibrate", "mcpt_renderer_set_tile_order", "mcpt_renderer_set_class_sort", "mcpt_renderer_set_pool_walk", "mcpt_renderer_get_walk", "mcpt_renderer_set_stream_waves", "mcpt_debug_intersect", "mcpt_debug_bsdf", "mcpt_debug_lbvh_build", "mcpt_debug_cost_table", "mcpt_debug_trace_pixel", "mcpt_debug_trace_rate", "mcpt_build_has_formulations",
    "mcpt_write_image", "mcpt_last_error", "mcpt_version",
    "mcpt_config_serialize", "mcpt_tiled_renderer_create", "mcpt_tiled_renderer_draw",
    "mcpt_tiled_renderer_set_kernel", "mcpt_tiled_renderer_destroy", "mcpt_render_tiled", "mcpt_device_count",
]


def debug_cost_table(steps, n_cus: int, layout: int = 1):
    """The wavefront-slot -> tile table laid out from probed tile costs (csrc/capi.cpp, CostOrderedTable)."""
    steps = np.ascontiguousarray(steps, np.uint32)
    table = np.zeros(len(steps), np.uint32)
    _check(lib().mcpt_debug_cost_table(steps.ctypes.data, len(steps), n_cus, layout, table.ctypes.data))
    return table


def lbvh_build(boxes, areas, on_device=False):
    """Reference-topology LBVH of boxes[n, 6] / areas[n] by the host builder or the HIP
    builder: (links[2n-1, 2] uint32 = (skip, object), geom[2n-1, 7] = (area, lo, hi), seconds)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    areas = np.ascontiguousarray(areas, dtype=np.float32)
    n = len(boxes)
    n_nodes = max(2 * n - 1, 0)
    links = np.zeros((max(n_nodes, 1), 2), dtype=np.uint32)
    geom = np.zeros((max(n_nodes, 1), 7), dtype=np.float32)
    sec = ctypes.c_double()
    _check(lib().mcpt_debug_lbvh_build(n, boxes.ctypes.data, areas.ctypes.data, 1 if on_device else 0,
                                       links.ctypes.data, geom.ctypes.data, ctypes.byref(sec)))
    return links[:n_nodes], geom[:n_nodes], sec.value


class Config:
    """A renderer configuration (csrt::RendererConfig)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def _make(fn, *args):
        h = ctypes.c_void_p()
        _check(fn(*args, ctypes.byref(h)))
        return Config(h)

    @classmethod
    def load_mcsd(cls, path):
        return cls._make(lib().mcpt_config_load_mcsd, str(path).encode())

    @classmethod
    def from_mcsd_bytes(cls, raw: bytes):
        return cls._make(lib().mcpt_config_from_mcsd_bytes, raw, len(raw))

    @classmethod
    def from_scene(cls, scene):
        """From a mcsd.Scene built in Python."""
        from . import mcsd
        return cls.from_mcsd_bytes(mcsd.dumps(scene))

    @classmethod
    def load_xml(cls, path, standins=None):
        """`standins`: text table of procedural stand-ins for mesh files that are not on disk
        (include/mcpt.h, mcpt_config_load_xml_with_standins)."""
        if standins is None:
            return cls._make(lib().mcpt_config_load_xml, str(path).encode())
        return cls._make(lib().mcpt_config_load_xml_with_standins, str(path).encode(), standins.encode())

    @classmethod
    def builtin(cls, name):
        return cls._make(lib().mcpt_config_builtin, name.encode())

    def set_instance_standin(self, instance, line):
        _check(lib().mcpt_config_set_instance_standin(self._h, instance, line.encode()))
        return self

    def set_film(self, width=0, height=0, spp=0):
        _check(lib().mcpt_config_set_film(self._h, width, height, spp))
        return self

    def film(self):
        w, h, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib().mcpt_config_get_film(self._h, ctypes.byref(w), ctypes.byref(h), ctypes.byref(s)))
        return w.value, h.value, s.value

    def save_mcsd(self, path):
        _check(lib().mcpt_config_save_mcsd(self._h, str(path).encode()))

    def serialize(self) -> bytes:
        n = ctypes.c_size_t()
        _check(lib().mcpt_config_serialize(self._h, None, 0, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        _check(lib().mcpt_config_serialize(self._h, buf, n.value, ctypes.byref(n)))
        return buf.raw[:n.value]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mcpt_config_destroy(self._h)
            self._h = None


class Renderer:
    """csrt::Renderer: commit once, draw many times."""

    def __init__(self, config: Config, device: int = 0):
        self.width, self.height, self.spp = config.film()
        h = ctypes.c_void_p()
        _check(lib().mcpt_renderer_create(config._h, device, ctypes.byref(h)))
        self._h = h
        self.device = device
        n = ctypes.c_uint32()
        _check(lib().mcpt_renderer_tile_count(self._h, ctypes.byref(n)))
        self.tiles_total = n.value

    def draw(self, counted=False):
        """Renderer::Draw(float*): returns (frame[h, w, 3] float32, stats dict)."""
        frame = np.empty((self.height, self.width, 3), dtype=np.float32)
        st = Stats()
        fn = lib().mcpt_renderer_draw_counted if counted else lib().mcpt_renderer_draw
        _check(fn(self._h, frame.ctypes.data, ctypes.byref(st)))
        return frame, st.as_dict()

    def draw_into(self, frame: np.ndarray, counted=False):
        """Renderer::Draw(float*) into the caller's host buffer (h x w x 3 float32, C order): blocking, the
        device-to-host copy of the frame included.  Returns the stats dict."""
        assert frame.dtype == np.float32 and frame.flags.c_contiguous and frame.size == self.height * self.width * 3
        st = Stats()
        fn = lib().mcpt_renderer_draw_counted if counted else lib().mcpt_renderer_draw
        _check(fn(self._h, frame.ctypes.data, ctypes.byref(st)))
        return st.as_dict()

    def tiles_in(self, rng: TileRange) -> int:
        return lib().mcpt_tile_range_size(self.tiles_total, ctypes.byref(rng))

    def draw_device(self, out_ptr: int, rng: TileRange = None, packed=False, stream=None,
                    blocking=True):
        """Render into device memory (e.g. a torch tensor's data_ptr())."""
        rng = rng or TileRange(0, 1, 0)
        st = Stats()
        _check(lib().mcpt_renderer_draw_device(self._h, ctypes.c_void_p(out_ptr), ctypes.byref(rng),
                                               int(packed), ctypes.c_void_p(stream or 0),
                                               int(blocking), ctypes.byref(st)))
        return st.as_dict()

    def debug_intersect(self, origins, dirs, seeds=None):
        """Closest hit of n rays on the GPU: (out[n, 19], seeds_after[n])."""
        rays = np.ascontiguousarray(np.concatenate([origins, dirs], axis=1), dtype=np.float32)
        n = len(rays)
        seeds = np.ones(n, dtype=np.uint32) if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
        out = np.zeros((n, 19), dtype=np.float32)
        after = np.zeros(n, dtype=np.uint32)
        _check(lib().mcpt_debug_intersect(self._h, n, rays.ctypes.data, seeds.ctypes.data,
                                          out.ctypes.data, after.ctypes.data))
        return out, after

    def debug_bsdf(self, id_bsdf, mode, records, seeds):
        records = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 18)
        n = len(records)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        out = np.zeros((n, 8), dtype=np.float32)
        after = np.zeros(n, dtype=np.uint32)
        _check(lib().mcpt_debug_bsdf(self._h, id_bsdf, mode, n, records.ctypes.data, seeds.ctypes.data,
                                     out.ctypes.data, after.ctypes.data))
        return out, after

    def trace_pixel(self, x, y, capacity=4096):
        """Per-step records of one pixel on the device (see mcpt.h): (steps[n, 16] float32,
        lcg[n] uint32 = column 11 reinterpreted)."""
        out = np.zeros((capacity, 16), dtype=np.float32)
        n = np.zeros(1, dtype=np.uint32)
        _check(lib().mcpt_debug_trace_pixel(self._h, y * self.width + x, capacity, out.ctypes.data, n.ctypes.data))
        steps = out[:int(n[0])]
        return steps, steps[:, 11].copy().view(np.uint32)

    def trace_rate(self, rays, mode, waves_per_simd=4, refill_at=0):
        """Experiment kernel (mcpt.h): (found[n] uint32 primitive or 0xFFFFFFFF, milliseconds of the lean trace kernel)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        found = np.zeros(len(rays), dtype=np.uint32)
        ms = ctypes.c_float()
        _check(lib().mcpt_debug_trace_rate(self._h, len(rays), rays.ctypes.data, mode, waves_per_simd, refill_at, found.ctypes.data,
                                           ctypes.byref(ms)))
        return found, ms.value

    def table(self, what: str) -> np.ndarray:
        data, count = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().mcpt_renderer_table(self._h, what.encode(), ctypes.byref(data), ctypes.byref(count)))
        buf = (ctypes.c_float * count.value).from_address(data.value)
        return np.frombuffer(buf, dtype=np.float32).copy()

    def info(self):
        arr = (ctypes.c_uint64 * 9)()
        _check(lib().mcpt_renderer_info(self._h, arr))
        keys = ("nodes", "tlas_nodes", "primitives", "instances", "features", "geometry_bytes",
                "walk_nodes", "walk_depth", "has_masks")
        return dict(zip(keys, (int(v) for v in arr)))

    def set_walk(self, reference_order: bool):
        """False (default): ordered walk of the SAH hierarchy; True: the reference's
        trees in the reference's order (validation mode)."""
        _check(lib().mcpt_renderer_set_walk(self._h, 1 if reference_order else 0))
        return self

    def check_walks(self):
        """Both ray queries on this film, compared bit for bit: (pixels that differ, first such pixel or None,
        largest absolute difference)."""
        n, first, worst = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_float()
        _check(lib().mcpt_renderer_check_walks(self._h, ctypes.byref(n), ctypes.byref(first), ctypes.byref(worst)))
        return n.value, (None if first.value == 0xFFFFFFFF else first.value), worst.value

    def set_kernel(self, stream, slots: int = 0, refill_at: int = 0):
        """-1 (default): by scene class; True / 1: the stream kernel (workgroup-local ray pool) wherever the scene
        allows it; False / 0: the lane-owns-a-path kernel; 2: stream with `slots` slots per workgroup in memory.
        The image does not depend on it."""
        _check(lib().mcpt_renderer_set_kernel(self._h, int(stream), slots, refill_at))   # 2: slots in memory
        return self

    def set_rng(self, mode: int, seed: int = 0, sample_split: int = 0):
        """0: the reference's random stream (default; frames comparable per pixel).  1: throughput mode — an
        independent PCG-hashed stream per (pixel, sample), samples of a pixel spread over `sample_split` lanes.  2: the same with
        Owen-scrambled Sobol points instead of pseudo-random numbers (at most 8192 spp; smaller error at equal spp)."""
        _check(lib().mcpt_renderer_set_rng(self._h, mode, seed, sample_split))
        return self

    def calibrate(self):
        """mcpt_renderer_calibrate: time the kernel configurations on this scene now and store the winner for mode -1."""
        _check(lib().mcpt_renderer_calibrate(self._h))
        return self

    def last_choice(self):
        """(kernel, work distribution, pre-pass) the last draw ran — arguments for the three setters."""
        k, w, p = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib().mcpt_renderer_last_choice(self._h, ctypes.byref(k), ctypes.byref(w), ctypes.byref(p)))
        return k.value, w.value, p.value

    def set_work_distribution(self, mode: int):
        """-1 library's choice (default), 0 fixed per-lane pixel lists, 1 work counter (dynamic).  Image unchanged."""
        _check(lib().mcpt_renderer_set_work_distribution(self._h, mode))
        return self

    def set_prepass(self, mode: int):
        """Primary-visibility pre-pass (all camera rays ahead of the sample chains): -1 library's choice (default),
        0 off, 1 on where the scene allows it.  The image does not depend on it."""
        _check(lib().mcpt_renderer_set_prepass(self._h, mode))
        return self

    def set_class_sort(self, mode: int):
        """Class-sorted form of the lanes kernel (full-feature scenes in LDS): -1 library's choice (on), 0 off, 1 on.  Same frame."""
        _check(lib().mcpt_renderer_set_class_sort(self._h, mode))
        return self

    def walk(self) -> int:
        """1: draws use the reference-order walk (set_walk(1), or the self-check of the creation fell back to it); 0: the ordered walk."""
        v = ctypes.c_int()
        _check(lib().mcpt_renderer_get_walk(self._h, ctypes.byref(v)))
        return int(v.value)

    def set_pool_walk(self, mode: int):
        """Ray queries of the lanes kernel on LDS-resident scenes: -1 library's choice, 0 one walk per lane, 1 the
        wavefront-cooperative pool walk (csrc/pool_walk.h), 2 the same with merged queries in the lean LDS kernels too.  Same frame."""
        _check(lib().mcpt_renderer_set_pool_walk(self._h, mode))
        return self

    def set_stream_waves(self, waves: int):
        """Register budget of the stream kernel on meshes: -1 library's rule, 2 / 3 / 4 wavefronts per SIMD.  Same frame."""
        _check(lib().mcpt_renderer_set_stream_waves(self._h, waves))
        return self

    def set_tile_order(self, mode: int):
        """Hand-out order of the tiles: -1 library's choice, 0 image order, 1 most expensive first (from the pre-pass), 2 image order in
        eight bands, one per XCD.  Same frame."""
        _check(lib().mcpt_renderer_set_tile_order(self._h, mode))
        return self

    def set_pixel_order(self, mode: int):
        """Lanes kernel: -1 library's choice, 0 a wavefront renders one 8x8 tile, 1 transposed (64 tiles).  Same frame."""
        _check(lib().mcpt_renderer_set_pixel_order(self._h, mode))
        return self

    def set_lane_spread(self, lanes_per_path: int):
        """Small jobs: only every n-th lane of a wavefront takes pixels (0 = library's choice, 1 = dense).  Same frame."""
        _check(lib().mcpt_renderer_set_lane_spread(self._h, lanes_per_path))
        return self

    def last_kernel(self) -> str:
        return lib().mcpt_renderer_last_kernel(self._h).decode()

    def set_walk_schedule(self, leave_below: int, leave_at: int):
        """Vote thresholds of the ordered walk on large scenes (see mcpt.h); the image does not change."""
        _check(lib().mcpt_renderer_set_walk_schedule(self._h, leave_below, leave_at))
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().mcpt_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def device_count() -> int:
    n = ctypes.c_int()
    _check(lib().mcpt_device_count(ctypes.byref(n)))
    return n.value


TILED_ALWAYS_GATHER = 1
TILED_LOGICAL_RANKS = 2   # test switch: one device listed several times (needs MCPT_RCCL_LIBRARY = tests/rccl_shim)


class TiledRenderer:
    """One frame over several GPUs of this process (mcpt_tiled_renderer_*): one commit, one renderer per
    device, one RCCL gather of finished tiles to devices[0]; draw() is blocking and returns the whole frame
    like csrt::Renderer::Draw."""

    def __init__(self, config: Config, devices=(0,), flags: int = 0):
        self.width, self.height, self.spp = config.film()
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        _check(lib().mcpt_tiled_renderer_create(config._h, len(devices), devs, flags, ctypes.byref(h)))
        self._h = h

    def draw(self):
        frame = np.empty((self.height, self.width, 3), dtype=np.float32)
        st = Stats()
        _check(lib().mcpt_tiled_renderer_draw(self._h, frame.ctypes.data, ctypes.byref(st)))
        return frame, st.as_dict()

    def set_kernel(self, mode, slots: int = 0, refill_at: int = 0):
        _check(lib().mcpt_tiled_renderer_set_kernel(self._h, int(mode), slots, refill_at))
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().mcpt_tiled_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


HOST_LIB_PATH = os.path.join(HERE, "libmcpt_host.so")


def host_render(config: Config, threads: int = 0, rng: TileRange = None, packed: bool = False, out: np.ndarray = None):
    """The optional host build of the kernel body (include/mcpt_host.h; what `mcpt_cli --cpu` runs):
    (frame, seconds).  A separate shared object — libmcpt_hip.so has no CPU path.  With `rng` only that tile
    range is rendered — into `out` / a zeroed full frame, or with packed=True into a packed tile buffer
    (tiles x 64 x 3), exactly as mcpt_renderer_draw_device lays it out on the GPU."""
    if not os.path.exists(HOST_LIB_PATH):
        raise McptError(f"{HOST_LIB_PATH} is not built")
    H = ctypes.CDLL(HOST_LIB_PATH)
    H.mcpt_host_render_tiles.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    H.mcpt_host_last_error.restype = ctypes.c_char_p
    w, h, _ = config.film()
    raw = config.serialize()
    rng = rng or TileRange(0, 1, 0)
    if out is None:
        if packed:
            tiles = ((w + 7) // 8) * ((h + 7) // 8)
            out = np.zeros((lib().mcpt_tile_range_size(tiles, ctypes.byref(rng)), 64, 3), dtype=np.float32)
        else:
            out = np.zeros((h, w, 3), dtype=np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous
    sec = ctypes.c_double()
    if H.mcpt_host_render_tiles(raw, len(raw), threads, rng.tile_first, rng.tile_stride, rng.tile_count, int(packed),
                                out.ctypes.data, ctypes.byref(sec)) != 0:
        raise McptError(H.mcpt_host_last_error().decode(errors="replace"))
    return out, sec.value


def unpack_tiles(packed: np.ndarray, rng: TileRange, width: int, height: int, frame: np.ndarray):
    packed = np.ascontiguousarray(packed, dtype=np.float32)
    assert frame.dtype == np.float32 and frame.flags.c_contiguous
    _check(lib().mcpt_unpack_tiles(packed.ctypes.data, ctypes.byref(rng), width, height, frame.ctypes.data))


def write_image(path, frame: np.ndarray):
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    _check(lib().mcpt_write_image(str(path).encode(), frame.ctypes.data, frame.shape[1], frame.shape[0]))
