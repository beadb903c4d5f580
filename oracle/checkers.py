"""TEST INFRASTRUCTURE — ctypes loaders for the two CPU checkers.

  Oracle     oracle/libmcpt_oracle.so  (our restatement, built by `make oracle`)
  Reference  oracle/_ref/libmcpt_ref.so (the real reference compiled from its
             own sources, built by `make ref` where /root/reference exists)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libmcpt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmcpt_ref.so")
REFERENCE_DIR = os.environ.get("MCPT_REFERENCE_DIR", "/root/reference")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(ref: bool = True, quiet: bool = True) -> None:
    """Compile the checkers (the reference build only where its sources exist)."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True, stdout=out)
    if ref and os.path.isdir(os.path.join(REFERENCE_DIR, "src")):
        subprocess.run(["make", "-s", "-j8", "-C", HERE, "ref",
                        f"REFERENCE={REFERENCE_DIR}"], check=True, stdout=out)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        lib = ctypes.CDLL(ORACLE_SO)
        lib.mcpt_oracle_last_error.restype = ctypes.c_char_p
        lib.mcpt_oracle_load.restype = ctypes.c_void_p
        lib.mcpt_oracle_load.argtypes = [ctypes.c_char_p]
        lib.mcpt_oracle_free.argtypes = [ctypes.c_void_p]
        lib.mcpt_oracle_dims.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 3
        lib.mcpt_oracle_render.restype = ctypes.c_int
        lib.mcpt_oracle_render.argtypes = [
            ctypes.c_void_p, _f32p, ctypes.c_uint32, ctypes.c_uint32,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.mcpt_oracle_trace_pixel.argtypes = [ctypes.c_void_p, ctypes.c_uint32,
                                                ctypes.c_uint32, _f32p, _u32p]
        lib.mcpt_oracle_bsdf.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int,
                                         _f32p, ctypes.POINTER(ctypes.c_uint32), _f32p]
        lib.mcpt_oracle_intersect.argtypes = [ctypes.c_void_p, _f32p, _f32p,
                                              ctypes.POINTER(ctypes.c_uint32), _f32p]
        lib.mcpt_oracle_node_count.restype = ctypes.c_uint32
        lib.mcpt_oracle_node_count.argtypes = [ctypes.c_void_p]
        lib.mcpt_oracle_nodes.argtypes = [ctypes.c_void_p, _u32p, _f32p]
        lib.mcpt_oracle_kulla_conty.argtypes = [_f32p, _f32p]
        lib.mcpt_oracle_bvh_build.restype = ctypes.c_int
        lib.mcpt_oracle_bvh_build.argtypes = [ctypes.c_uint32, _f32p, _f32p] + \
            [_u32p] * 4 + [_f32p] * 2
        lib.mcpt_oracle_tea4.restype = ctypes.c_uint32
        lib.mcpt_oracle_tea4.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        lib.mcpt_oracle_random_float.restype = ctypes.c_float
        lib.mcpt_oracle_random_float.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
        lib.mcpt_oracle_vdc2.restype = ctypes.c_float
        lib.mcpt_oracle_vdc2.argtypes = [ctypes.c_uint32]
        lib.mcpt_oracle_vdc3.restype = ctypes.c_float
        lib.mcpt_oracle_vdc3.argtypes = [ctypes.c_uint32]
        self.lib = lib

    def _err(self):
        return self.lib.mcpt_oracle_last_error().decode()

    def render(self, mcsd_path, threads=0, first_pixel=0, n_pixel=None,
               with_stats=False):
        """Returns (frame[h,w,3] float32, info dict)."""
        h = self.lib.mcpt_oracle_load(str(mcsd_path).encode())
        if not h:
            raise RuntimeError(self._err())
        try:
            w_, h_, spp = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint32()
            self.lib.mcpt_oracle_dims(h, ctypes.byref(w_), ctypes.byref(h_),
                                      ctypes.byref(spp))
            frame = np.zeros((h_.value, w_.value, 3), dtype=np.float32)
            total = w_.value * h_.value
            n = total - first_pixel if n_pixel is None else n_pixel
            sec = ctypes.c_double()
            stats = (ctypes.c_uint64 * 4)()
            rc = self.lib.mcpt_oracle_render(
                h, frame, first_pixel, n, threads, ctypes.byref(sec),
                ctypes.cast(stats, ctypes.c_void_p) if with_stats else None)
            if rc != 0:
                raise RuntimeError(self._err())
            info = {"seconds": sec.value, "width": w_.value,
                    "height": h_.value, "spp": spp.value, "pixels": n}
            if with_stats:
                info.update(closest_rays=stats[0], shadow_rays=stats[1],
                            node_tests=stats[2], prim_tests=stats[3])
            return frame, info
        finally:
            self.lib.mcpt_oracle_free(h)

    def trace_pixel(self, mcsd_path, i, j):
        """Per-sample (radiance[spp,3], lcg_state_after[spp]) of pixel (i, j)."""
        h = self.lib.mcpt_oracle_load(str(mcsd_path).encode())
        if not h:
            raise RuntimeError(self._err())
        try:
            w_, h_, spp = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint32()
            self.lib.mcpt_oracle_dims(h, ctypes.byref(w_), ctypes.byref(h_),
                                      ctypes.byref(spp))
            rad = np.zeros((spp.value, 3), dtype=np.float32)
            st = np.zeros(spp.value, dtype=np.uint32)
            self.lib.mcpt_oracle_trace_pixel(h, i, j, rad, st)
            return rad, st
        finally:
            self.lib.mcpt_oracle_free(h)

    def open(self, mcsd_path):
        h = self.lib.mcpt_oracle_load(str(mcsd_path).encode())
        if not h:
            raise RuntimeError(self._err())
        return _Session(self.lib.mcpt_oracle_bsdf, self.lib.mcpt_oracle_intersect,
                        self.lib.mcpt_oracle_free, h)

    def nodes(self, mcsd_path):
        h = self.lib.mcpt_oracle_load(str(mcsd_path).encode())
        if not h:
            raise RuntimeError(self._err())
        try:
            n = self.lib.mcpt_oracle_node_count(h)
            links = np.zeros((n, 4), dtype=np.uint32)
            geom = np.zeros((n, 7), dtype=np.float32)
            self.lib.mcpt_oracle_nodes(h, links, geom)
            return links, geom
        finally:
            self.lib.mcpt_oracle_free(h)

    def kulla_conty(self):
        brdf = np.zeros(128 * 128, dtype=np.float32)
        albedo = np.zeros(128, dtype=np.float32)
        self.lib.mcpt_oracle_kulla_conty(brdf, albedo)
        return brdf, albedo

    def bvh_build(self, aabbs, areas):
        return _bvh_build(self.lib.mcpt_oracle_bvh_build, aabbs, areas)

    def lcg(self, seed, n):
        s = ctypes.c_uint32(seed)
        vals = [self.lib.mcpt_oracle_random_float(ctypes.byref(s)) for _ in range(n)]
        return vals, s.value


class _Session:
    """Unit-level calls against one committed scene (either checker)."""

    def __init__(self, bsdf_fn, intersect_fn, close_fn, handle):
        self._bsdf, self._intersect, self._close, self._h = bsdf_fn, intersect_fn, close_fn, handle

    def close(self):
        if self._h:
            self._close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def bsdf(self, id_bsdf, mode, rec, seed):
        """rec: 18 floats (wo, wi, n, t, b, uv, inside).  Returns (out[8], seed)."""
        rec = np.ascontiguousarray(rec, dtype=np.float32)
        out = np.zeros(8, dtype=np.float32)
        s = ctypes.c_uint32(seed)
        self._bsdf(self._h, id_bsdf, mode, rec, ctypes.byref(s), out)
        return out, s.value

    def intersect(self, origin, direction, seed=1):
        out = np.zeros(19, dtype=np.float32)
        s = ctypes.c_uint32(seed)
        self._intersect(self._h, np.ascontiguousarray(origin, dtype=np.float32),
                        np.ascontiguousarray(direction, dtype=np.float32),
                        ctypes.byref(s), out)
        return out, s.value


def _bvh_build(fn, aabbs, areas):
    aabbs = np.ascontiguousarray(aabbs, dtype=np.float32).reshape(-1, 6)
    areas = np.ascontiguousarray(areas, dtype=np.float32)
    n = len(areas)
    m = max(2 * n - 1, 1)
    leaf, left, right, obj = (np.zeros(m, dtype=np.uint32) for _ in range(4))
    area = np.zeros(m, dtype=np.float32)
    box = np.zeros((m, 6), dtype=np.float32)
    count = fn(n, aabbs, areas, leaf, left, right, obj, area, box)
    if count < 0:
        raise RuntimeError("bvh build failed")
    return dict(leaf=leaf[:count], left=left[:count], right=right[:count],
                object=obj[:count], area=area[:count], box=box[:count])


def reference_available() -> bool:
    return os.path.exists(REF_SO)


STB_SO = os.path.join(HERE, "_ref", "libstb_ref.so")


def stb_available() -> bool:
    return os.path.exists(STB_SO)


class Stb:
    """The reference's vendored image decoder / resizer (extern/stb), compiled by
    oracle/Makefile from the headers where they lie (oracle/stb_driver.c)."""

    def __init__(self):
        if not os.path.exists(STB_SO):
            raise RuntimeError("oracle/_ref/libstb_ref.so is not built")
        self.lib = ctypes.CDLL(STB_SO)
        ip = ctypes.POINTER(ctypes.c_int)
        self.lib.mcpt_stb_load8.argtypes = [ctypes.c_char_p, ip, ip, ip, ctypes.c_void_p, ctypes.c_size_t]
        self.lib.mcpt_stb_loadf.argtypes = [ctypes.c_char_p, ip, ip, ip, ctypes.c_void_p, ctypes.c_size_t]
        self.lib.mcpt_stb_resize.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int]

    def _load(self, fn, dtype, path, capacity=1 << 26):
        w, h, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        buf = np.zeros(capacity, dtype=dtype)
        rc = fn(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), buf.ctypes.data, capacity)
        if rc != 0:
            raise RuntimeError(f"stb_image could not read {path} (rc {rc})")
        return buf[:w.value * h.value * c.value].reshape(h.value, w.value, c.value).copy()

    def load8(self, path):
        return self._load(self.lib.mcpt_stb_load8, np.uint8, path)

    def loadf(self, path):
        return self._load(self.lib.mcpt_stb_loadf, np.float32, path)

    def resize(self, image, out_w, out_h):
        image = np.ascontiguousarray(image, dtype=np.float32)
        h, w, c = image.shape
        out = np.zeros((out_h, out_w, c), dtype=np.float32)
        if self.lib.mcpt_stb_resize(image.ctypes.data, w, h, out.ctypes.data, out_w, out_h, c) != 0:
            raise RuntimeError("stbir_resize_float_linear failed")
        return out


class Reference:
    """The compiled reference (oracle/_ref).  Not reentrant (file-scope globals
    in renderer.cpp:17-22): one render at a time per process."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise RuntimeError("oracle/_ref/libmcpt_ref.so is not built")
        lib = ctypes.CDLL(REF_SO)
        lib.mcpt_ref_last_error.restype = ctypes.c_char_p
        lib.mcpt_ref_render.restype = ctypes.c_int
        lib.mcpt_ref_render.argtypes = [ctypes.c_char_p, _f32p, ctypes.c_void_p]
        lib.mcpt_ref_trace_pixel.restype = ctypes.c_int
        lib.mcpt_ref_trace_pixel.argtypes = [ctypes.c_char_p, ctypes.c_uint32,
                                             ctypes.c_uint32, _f32p, _u32p]
        lib.mcpt_ref_open.restype = ctypes.c_void_p
        lib.mcpt_ref_open.argtypes = [ctypes.c_char_p]
        lib.mcpt_ref_close.argtypes = [ctypes.c_void_p]
        lib.mcpt_ref_bsdf.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int,
                                      _f32p, ctypes.POINTER(ctypes.c_uint32), _f32p]
        lib.mcpt_ref_intersect.argtypes = [ctypes.c_void_p, _f32p, _f32p,
                                           ctypes.POINTER(ctypes.c_uint32), _f32p]
        lib.mcpt_ref_tea4.restype = ctypes.c_uint32
        lib.mcpt_ref_tea4.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        lib.mcpt_ref_random_float.restype = ctypes.c_float
        lib.mcpt_ref_random_float.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
        lib.mcpt_ref_vdc2.restype = ctypes.c_float
        lib.mcpt_ref_vdc2.argtypes = [ctypes.c_uint32]
        lib.mcpt_ref_vdc3.restype = ctypes.c_float
        lib.mcpt_ref_vdc3.argtypes = [ctypes.c_uint32]
        lib.mcpt_ref_kulla_conty.argtypes = [_f32p, _f32p]
        lib.mcpt_ref_binding_round_trip.restype = ctypes.c_int
        lib.mcpt_ref_binding_round_trip.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.mcpt_ref_bvh_build.restype = ctypes.c_int
        lib.mcpt_ref_bvh_build.argtypes = [ctypes.c_uint32, _f32p, _f32p] + \
            [_u32p] * 4 + [_f32p] * 2
        self.lib = lib

    def render(self, mcsd_path, width, height):
        """Runs Renderer::Draw on all host threads; progress goes to stderr."""
        frame = np.zeros((height, width, 3), dtype=np.float32)
        sec = ctypes.c_double()
        # the reference prints a progress line per 64-pixel patch to stderr
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(2)
        os.dup2(devnull, 2)
        try:
            rc = self.lib.mcpt_ref_render(str(mcsd_path).encode(), frame,
                                          ctypes.byref(sec))
        finally:
            os.dup2(saved, 2)
            os.close(saved)
            os.close(devnull)
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_ref_last_error().decode())
        return frame, {"seconds": sec.value}

    def binding_round_trip(self, mcsd_path, out_path):
        """MCSD -> csrt::RendererConfig -> MCSD through integration/mcpt_backend.hpp."""
        if self.lib.mcpt_ref_binding_round_trip(str(mcsd_path).encode(), str(out_path).encode()) != 0:
            raise RuntimeError(self.lib.mcpt_ref_last_error().decode())

    def open(self, mcsd_path):
        h = self.lib.mcpt_ref_open(str(mcsd_path).encode())
        if not h:
            raise RuntimeError(self.lib.mcpt_ref_last_error().decode())
        return _Session(self.lib.mcpt_ref_bsdf, self.lib.mcpt_ref_intersect,
                        self.lib.mcpt_ref_close, h)

    def trace_pixel(self, mcsd_path, i, j, spp):
        rad = np.zeros((spp, 3), dtype=np.float32)
        st = np.zeros(spp, dtype=np.uint32)
        rc = self.lib.mcpt_ref_trace_pixel(str(mcsd_path).encode(), i, j, rad, st)
        if rc != 0:
            raise RuntimeError(self.lib.mcpt_ref_last_error().decode())
        return rad, st

    def kulla_conty(self):
        brdf = np.zeros(128 * 128, dtype=np.float32)
        albedo = np.zeros(128, dtype=np.float32)
        self.lib.mcpt_ref_kulla_conty(brdf, albedo)
        return brdf, albedo

    def bvh_build(self, aabbs, areas):
        return _bvh_build(self.lib.mcpt_ref_bvh_build, aabbs, areas)

    def lcg(self, seed, n):
        s = ctypes.c_uint32(seed)
        vals = [self.lib.mcpt_ref_random_float(ctypes.byref(s)) for _ in range(n)]
        return vals, s.value
