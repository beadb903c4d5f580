"""TEST INFRASTRUCTURE — ctypes loader of tests/emu/libmcpt_wave_emu.so: the product's kernel bodies (render_body, the
cooperative pool walk, the uniform / merged path steps) on the host behind the 64-lane lockstep shim (wave_shim.h,
wave_emu.cpp)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmcpt_wave_emu.so")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

# feature bits (csrc/device_scene.h) and the launcher's names for their combinations (csrc/hip/render_kernel_impl.h)
VOLPATH, EMITTERS, ANALYTIC, TEXTURES, MICROFACET = 1, 2, 4, 8, 16
ORDERED, SLIVERS, POOL, POOL_BIG, POOL_MERGE = 1 << 5, 1 << 7, 1 << 14, 1 << 15, 1 << 17
ALL = VOLPATH | EMITTERS | ANALYTIC | TEXTURES | MICROFACET
SURFACE = EMITTERS | TEXTURES | MICROFACET
P = ORDERED | POOL
PM = P | POOL_MERGE
PBU = P | POOL_BIG
PB = PBU | POOL_MERGE
GROUP128 = 1 << 18
NO_TRANSMISSION, DIELECTRIC_ONLY, CONDUCTOR_ONLY = 1 << 11, 1 << 12, 1 << 13
VOLUME_LEAN = VOLPATH | ANALYTIC | MICROFACET


class Options(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("order", "seed", "poison", "poison_word", "max_blocks", "per_cu", "lane_spread",
                                               "compact", "scatter", "threads", "xcd_bands", "lds_shortfall")]


class Report(ctypes.Structure):
    _fields_ = [("collectives", ctypes.c_uint64), ("rounds", ctypes.c_uint64), ("queries", ctypes.c_uint64),
                ("blocks", ctypes.c_uint32), ("lds_bytes", ctypes.c_uint32), ("lane_spread", ctypes.c_uint32),
                ("scatter", ctypes.c_uint32)]


class WaveEmulator:
    def __init__(self):
        subprocess.run(["make", "-s", "-C", HERE, "wave"], check=True)
        lib = ctypes.CDLL(SO)
        lib.mcpt_wave_emu_last_error.restype = ctypes.c_char_p
        lib.mcpt_wave_emu_render.restype = ctypes.c_int
        lib.mcpt_wave_emu_render.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(Options), _f32p,
                                             ctypes.c_void_p, ctypes.POINTER(Report)]
        lib.mcpt_wave_emu_render_sorted.restype = ctypes.c_int
        lib.mcpt_wave_emu_render_sorted.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(Options), _f32p, ctypes.POINTER(Report)]
        self.lib = lib

    def render_sorted(self, mcsd_path, width, height, features, lds, order=0, seed=0, poison=None, max_blocks=4, per_cu=1, threads=0):
        """The class-sorted kernel's body (csrc/hip/sorted_body.h), workgroups of 128 lanes.  -> (frame, report dict)"""
        opt = Options(order, seed, 0 if poison is None else 1, poison or 0, max_blocks, per_cu, 1, 0, 0, threads, 0, 0)
        frame = np.zeros((height, width, 3), dtype=np.float32)
        rep = Report()
        rc = self.lib.mcpt_wave_emu_render_sorted(str(mcsd_path).encode(), features, 1 if lds else 0, ctypes.byref(opt), frame, ctypes.byref(rep))
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_wave_emu_last_error().decode())
        return frame, {n: getattr(rep, n) for n, _ in Report._fields_}

    def render(self, mcsd_path, width, height, features, lds, order=0, seed=0, poison=None, max_blocks=4, per_cu=1,
               lane_spread=0, compact=0, scatter=0xFFFFFFFF, threads=0, counted=False, xcd_bands=0):
        """-> (frame, report dict).  order: 0 ascending / 1 descending / 2 shuffled lanes between collectives; poison: None or
        the 32-bit word a wavefront's pool area is filled with before every ray query."""
        opt = Options(order, seed, 0 if poison is None else 1, poison or 0, max_blocks, per_cu, lane_spread, compact, scatter, threads, xcd_bands, 0)
        frame = np.zeros((height, width, 3), dtype=np.float32)
        counters = np.zeros(8, dtype=np.uint64)
        rep = Report()
        rc = self.lib.mcpt_wave_emu_render(str(mcsd_path).encode(), features, 1 if lds else 0, ctypes.byref(opt), frame,
                                           counters.ctypes.data if counted else None, ctypes.byref(rep))
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_wave_emu_last_error().decode())
        info = {n: getattr(rep, n) for n, _ in Report._fields_}
        if counted:
            info["counters"] = counters
        return frame, info
