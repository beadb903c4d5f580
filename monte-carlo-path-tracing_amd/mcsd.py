"""MCSD ("Monte-Carlo scene description") reader / writer.

MCSD is the on-disk form of the renderer configuration that crosses the
drop-in boundary: one record per field of the reference's
``csrt::RendererConfig`` (reference: include/csrt/renderer/renderer.hpp:18-28;
camera include/csrt/renderer/camera.hpp:13-22, integrator
include/csrt/renderer/integrators/integrator.hpp:17-29, textures
include/csrt/renderer/textures/texture.hpp:21-27, BSDFs
include/csrt/renderer/bsdfs/bsdf.hpp:40-58, media
include/csrt/renderer/medium/medium.hpp:40-45, instances
include/csrt/rtcore/instance.hpp:30-51, emitters
include/csrt/renderer/emitters/emitter.hpp:30-47).

The byte layout is specified in include/mcsd_format.h.  Three independent
readers consume it: the product host library (csrc/host/mcsd_io.cpp), the
oracle restatement (oracle/mcpt_oracle.cpp) and the driver that feeds the
compiled reference (oracle/ref_driver.cpp).  Everything is little-endian
32-bit words.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

MAGIC = b"MCSD"
VERSION = 1
INVALID = 0xFFFFFFFF

# integrator types (integrator.hpp:11-15)
INTEGRATOR_PATH, INTEGRATOR_VOLPATH = 0, 1
# texture types (texture.hpp:13-19)
TEX_CONSTANT, TEX_CHECKERBOARD, TEX_BITMAP = 1, 2, 3
# bsdf types (bsdf.hpp:17-27)
(BSDF_AREA_LIGHT, BSDF_DIFFUSE, BSDF_ROUGH_DIFFUSE, BSDF_CONDUCTOR,
 BSDF_DIELECTRIC, BSDF_THIN_DIELECTRIC, BSDF_PLASTIC) = 1, 2, 3, 4, 5, 6, 7
# phase functions (medium.hpp:15-19)
PHASE_ISOTROPIC, PHASE_HG = 0, 1
# instance types (instance.hpp:15-24)
(INST_CUBE, INST_RECTANGLE, INST_MESHES, INST_SPHERE, INST_DISK,
 INST_CYLINDER) = 1, 2, 3, 4, 5, 6
# emitter types (emitter.hpp:19-28)
(EMIT_POINT, EMIT_SPOT, EMIT_DIRECTIONAL, EMIT_SUN, EMIT_ENVMAP,
 EMIT_CONSTANT) = 1, 2, 3, 4, 5, 6

BSDF_PAYLOAD_WORDS = 12
EMITTER_PAYLOAD_WORDS = 24

IDENTITY = np.eye(4, dtype=np.float32)


def _f32(x):
    return np.asarray(x, dtype=np.float32)


@dataclass
class Camera:
    spp: int = 64
    width: int = 1024
    height: int = 1024
    fov_x: float = 19.5
    eye: tuple = (0.0, 1.0, 6.8)
    look_at: tuple = (0.0, 1.0, 0.0)
    up: tuple = (0.0, 1.0, 0.0)


@dataclass
class Integrator:
    type: int = INTEGRATOR_PATH
    hide_emitters: bool = False
    pdf_rr: float = 0.95
    depth_rr: int = 5
    depth_max: int = INVALID


@dataclass
class Texture:
    type: int = TEX_CONSTANT
    color: tuple = (0.5, 0.5, 0.5)          # constant
    color0: tuple = (0.4, 0.4, 0.4)         # checkerboard
    color1: tuple = (0.2, 0.2, 0.2)
    to_uv: np.ndarray = field(default_factory=lambda: IDENTITY.copy())
    width: int = 0                          # bitmap
    height: int = 0
    channel: int = 0
    data: Optional[np.ndarray] = None


@dataclass
class Bsdf:
    type: int = BSDF_DIFFUSE
    twosided: bool = False
    id_opacity: int = INVALID
    id_bump_map: int = INVALID
    # area light
    weight: float = 1.0
    id_radiance: int = INVALID
    # diffuse / rough diffuse / plastic
    id_diffuse_reflectance: int = INVALID
    id_roughness: int = INVALID
    use_fast_approx: bool = True
    # conductor / dielectric
    id_roughness_u: int = INVALID
    id_roughness_v: int = INVALID
    id_specular_reflectance: int = INVALID
    id_specular_transmittance: int = INVALID
    reflectivity: tuple = (0.0, 0.0, 0.0)
    edgetint: tuple = (0.0, 0.0, 0.0)
    eta: float = 1.0


@dataclass
class Medium:
    sigma_a: tuple = (0.0, 0.0, 0.0)
    sigma_s: tuple = (0.0, 0.0, 0.0)
    phase_type: int = PHASE_ISOTROPIC
    g: tuple = (0.0, 0.0, 0.0)


@dataclass
class Instance:
    type: int = INST_MESHES
    id_bsdf: int = INVALID
    id_medium_int: int = INVALID
    id_medium_ext: int = INVALID
    flip_normals: bool = False
    to_world: np.ndarray = field(default_factory=lambda: IDENTITY.copy())
    sphere_radius: float = 1.0
    sphere_center: tuple = (0.0, 0.0, 0.0)
    cyl_radius: float = 1.0
    cyl_p0: tuple = (0.0, 0.0, 0.0)
    cyl_p1: tuple = (0.0, 0.0, 0.0)
    texcoords: Optional[np.ndarray] = None   # (n,2) f32
    positions: Optional[np.ndarray] = None   # (n,3) f32
    normals: Optional[np.ndarray] = None     # (n,3) f32
    tangents: Optional[np.ndarray] = None    # (n,3) f32
    bitangents: Optional[np.ndarray] = None  # (n,3) f32
    indices: Optional[np.ndarray] = None     # (m,3) u32


@dataclass
class Emitter:
    type: int = EMIT_DIRECTIONAL
    position: tuple = (0.0, 0.0, 0.0)        # point
    intensity: tuple = (1.0, 1.0, 1.0)       # point / spot
    cutoff_angle: float = 0.0                # spot (radians)
    beam_width: float = 0.0
    id_texture: int = INVALID                # spot / sun
    to_world: np.ndarray = field(default_factory=lambda: IDENTITY.copy())
    direction: tuple = (0.0, 0.0, 0.0)       # directional / sun
    radiance: tuple = (0.0, 0.0, 0.0)        # directional / sun / constant
    cos_cutoff_angle: float = 0.0            # sun
    id_radiance: int = INVALID               # envmap


@dataclass
class Scene:
    camera: Camera = field(default_factory=Camera)
    integrator: Integrator = field(default_factory=Integrator)
    textures: List[Texture] = field(default_factory=list)
    bsdfs: List[Bsdf] = field(default_factory=list)
    media: List[Medium] = field(default_factory=list)
    instances: List[Instance] = field(default_factory=list)
    emitters: List[Emitter] = field(default_factory=list)


class _W:
    def __init__(self):
        self.parts = []

    def u32(self, *v):
        self.parts.append(np.asarray(v, dtype="<u4").tobytes())

    def i32(self, *v):
        self.parts.append(np.asarray(v, dtype="<i4").tobytes())

    def f32(self, *v):
        self.parts.append(np.asarray(v, dtype="<f4").reshape(-1).tobytes())

    def arr(self, a, dtype):
        self.parts.append(np.ascontiguousarray(a, dtype=dtype).tobytes())

    def bytes(self):
        return b"".join(self.parts)


def _count(a):
    return 0 if a is None else int(len(a))


def dumps(scene: Scene) -> bytes:
    w = _W()
    w.parts.append(MAGIC)
    w.u32(VERSION)
    c = scene.camera
    w.u32(c.spp)
    w.i32(c.width, c.height)
    w.f32(c.fov_x)
    w.f32(*c.eye)
    w.f32(*c.look_at)
    w.f32(*c.up)
    g = scene.integrator
    w.u32(g.type, int(g.hide_emitters))
    w.f32(g.pdf_rr)
    w.u32(g.depth_rr, g.depth_max & 0xFFFFFFFF)

    w.u32(len(scene.textures))
    for t in scene.textures:
        w.u32(t.type)
        if t.type == TEX_CONSTANT:
            w.f32(*t.color)
        elif t.type == TEX_CHECKERBOARD:
            w.f32(*t.color0)
            w.f32(*t.color1)
            w.f32(_f32(t.to_uv))
        elif t.type == TEX_BITMAP:
            w.i32(t.width, t.height, t.channel)
            w.f32(_f32(t.to_uv))
            d = _f32(t.data).reshape(-1)
            assert d.size == t.width * t.height * t.channel
            w.arr(d, "<f4")
        else:
            raise ValueError("texture type")

    w.u32(len(scene.bsdfs))
    for b in scene.bsdfs:
        w.u32(b.type, int(b.twosided), b.id_opacity, b.id_bump_map)
        p = _W()
        if b.type == BSDF_AREA_LIGHT:
            p.f32(b.weight)
            p.u32(b.id_radiance)
        elif b.type == BSDF_DIFFUSE:
            p.u32(b.id_diffuse_reflectance)
        elif b.type == BSDF_ROUGH_DIFFUSE:
            p.u32(int(b.use_fast_approx), b.id_diffuse_reflectance,
                  b.id_roughness)
        elif b.type == BSDF_CONDUCTOR:
            p.u32(b.id_roughness_u, b.id_roughness_v,
                  b.id_specular_reflectance)
            p.f32(*b.reflectivity)
            p.f32(*b.edgetint)
        elif b.type in (BSDF_DIELECTRIC, BSDF_THIN_DIELECTRIC):
            p.u32(b.id_roughness_u, b.id_roughness_v,
                  b.id_specular_reflectance, b.id_specular_transmittance)
            p.f32(b.eta)
        elif b.type == BSDF_PLASTIC:
            p.f32(b.eta)
            p.u32(b.id_roughness, b.id_diffuse_reflectance,
                  b.id_specular_reflectance)
        else:
            raise ValueError("bsdf type")
        raw = p.bytes()
        assert len(raw) <= 4 * BSDF_PAYLOAD_WORDS
        w.parts.append(raw + b"\0" * (4 * BSDF_PAYLOAD_WORDS - len(raw)))

    w.u32(len(scene.media))
    for m in scene.media:
        w.u32(0)
        w.f32(*m.sigma_a)
        w.f32(*m.sigma_s)
        w.u32(m.phase_type)
        w.f32(*m.g)

    w.u32(len(scene.instances))
    for s in scene.instances:
        w.u32(s.type, s.id_bsdf, s.id_medium_int, s.id_medium_ext,
              int(s.flip_normals))
        w.f32(_f32(s.to_world))
        w.f32(s.sphere_radius)
        w.f32(*s.sphere_center)
        w.f32(s.cyl_radius)
        w.f32(*s.cyl_p0)
        w.f32(*s.cyl_p1)
        w.u32(_count(s.texcoords), _count(s.positions), _count(s.normals),
              _count(s.tangents), _count(s.bitangents), _count(s.indices))
        for a, n in ((s.texcoords, 2), (s.positions, 3), (s.normals, 3),
                     (s.tangents, 3), (s.bitangents, 3)):
            if a is not None and len(a):
                a = _f32(a)
                assert a.ndim == 2 and a.shape[1] == n
                w.arr(a, "<f4")
        if s.indices is not None and len(s.indices):
            w.arr(np.asarray(s.indices).reshape(-1, 3), "<u4")

    w.u32(len(scene.emitters))
    for e in scene.emitters:
        w.u32(e.type)
        p = _W()
        if e.type == EMIT_POINT:
            p.f32(*e.position)
            p.f32(*e.intensity)
        elif e.type == EMIT_SPOT:
            p.f32(e.cutoff_angle, e.beam_width)
            p.u32(e.id_texture)
            p.f32(*e.intensity)
            p.f32(_f32(e.to_world))
        elif e.type == EMIT_DIRECTIONAL:
            p.f32(*e.direction)
            p.f32(*e.radiance)
        elif e.type == EMIT_SUN:
            p.f32(e.cos_cutoff_angle)
            p.u32(e.id_texture)
            p.f32(*e.direction)
            p.f32(*e.radiance)
        elif e.type == EMIT_ENVMAP:
            p.u32(e.id_radiance)
            p.f32(_f32(e.to_world))
        elif e.type == EMIT_CONSTANT:
            p.f32(*e.radiance)
        else:
            raise ValueError("emitter type")
        raw = p.bytes()
        assert len(raw) <= 4 * EMITTER_PAYLOAD_WORDS
        w.parts.append(raw + b"\0" * (4 * EMITTER_PAYLOAD_WORDS - len(raw)))
    return w.bytes()


def dump(scene: Scene, path) -> None:
    with open(path, "wb") as f:
        f.write(dumps(scene))


class _R:
    def __init__(self, raw: bytes):
        self.raw = raw
        self.off = 0

    def take(self, n, dtype):
        a = np.frombuffer(self.raw, dtype=dtype, count=n, offset=self.off)
        self.off += 4 * n
        return a

    def u32(self):
        return int(self.take(1, "<u4")[0])

    def i32(self):
        return int(self.take(1, "<i4")[0])

    def f32(self):
        return float(self.take(1, "<f4")[0])

    def vec(self, n):
        return tuple(float(x) for x in self.take(n, "<f4"))

    def mat(self):
        return self.take(16, "<f4").reshape(4, 4).copy()


def loads(raw: bytes) -> Scene:
    if raw[:4] != MAGIC:
        raise ValueError("not an MCSD file")
    r = _R(raw)
    r.off = 4
    if r.u32() != VERSION:
        raise ValueError("unsupported MCSD version")
    s = Scene()
    c = s.camera
    c.spp, c.width, c.height, c.fov_x = r.u32(), r.i32(), r.i32(), r.f32()
    c.eye, c.look_at, c.up = r.vec(3), r.vec(3), r.vec(3)
    g = s.integrator
    g.type, g.hide_emitters = r.u32(), bool(r.u32())
    g.pdf_rr, g.depth_rr, g.depth_max = r.f32(), r.u32(), r.u32()
    for _ in range(r.u32()):
        t = Texture(type=r.u32())
        if t.type == TEX_CONSTANT:
            t.color = r.vec(3)
        elif t.type == TEX_CHECKERBOARD:
            t.color0, t.color1, t.to_uv = r.vec(3), r.vec(3), r.mat()
        elif t.type == TEX_BITMAP:
            t.width, t.height, t.channel = r.i32(), r.i32(), r.i32()
            t.to_uv = r.mat()
            t.data = r.take(t.width * t.height * t.channel, "<f4").copy()
        else:
            raise ValueError("texture type")
        s.textures.append(t)
    for _ in range(r.u32()):
        b = Bsdf(type=r.u32(), twosided=bool(r.u32()), id_opacity=r.u32(),
                 id_bump_map=r.u32())
        end = r.off + 4 * BSDF_PAYLOAD_WORDS
        if b.type == BSDF_AREA_LIGHT:
            b.weight, b.id_radiance = r.f32(), r.u32()
        elif b.type == BSDF_DIFFUSE:
            b.id_diffuse_reflectance = r.u32()
        elif b.type == BSDF_ROUGH_DIFFUSE:
            b.use_fast_approx = bool(r.u32())
            b.id_diffuse_reflectance, b.id_roughness = r.u32(), r.u32()
        elif b.type == BSDF_CONDUCTOR:
            b.id_roughness_u, b.id_roughness_v = r.u32(), r.u32()
            b.id_specular_reflectance = r.u32()
            b.reflectivity, b.edgetint = r.vec(3), r.vec(3)
        elif b.type in (BSDF_DIELECTRIC, BSDF_THIN_DIELECTRIC):
            b.id_roughness_u, b.id_roughness_v = r.u32(), r.u32()
            b.id_specular_reflectance = r.u32()
            b.id_specular_transmittance = r.u32()
            b.eta = r.f32()
        elif b.type == BSDF_PLASTIC:
            b.eta = r.f32()
            b.id_roughness, b.id_diffuse_reflectance = r.u32(), r.u32()
            b.id_specular_reflectance = r.u32()
        else:
            raise ValueError("bsdf type")
        r.off = end
        s.bsdfs.append(b)
    for _ in range(r.u32()):
        r.u32()
        s.media.append(Medium(sigma_a=r.vec(3), sigma_s=r.vec(3),
                              phase_type=r.u32(), g=r.vec(3)))
    for _ in range(r.u32()):
        i = Instance(type=r.u32(), id_bsdf=r.u32(), id_medium_int=r.u32(),
                     id_medium_ext=r.u32(), flip_normals=bool(r.u32()))
        i.to_world = r.mat()
        i.sphere_radius, i.sphere_center = r.f32(), r.vec(3)
        i.cyl_radius, i.cyl_p0, i.cyl_p1 = r.f32(), r.vec(3), r.vec(3)
        n = [r.u32() for _ in range(6)]
        i.texcoords = r.take(2 * n[0], "<f4").reshape(-1, 2).copy()
        i.positions = r.take(3 * n[1], "<f4").reshape(-1, 3).copy()
        i.normals = r.take(3 * n[2], "<f4").reshape(-1, 3).copy()
        i.tangents = r.take(3 * n[3], "<f4").reshape(-1, 3).copy()
        i.bitangents = r.take(3 * n[4], "<f4").reshape(-1, 3).copy()
        i.indices = r.take(3 * n[5], "<u4").reshape(-1, 3).copy()
        s.instances.append(i)
    for _ in range(r.u32()):
        e = Emitter(type=r.u32())
        end = r.off + 4 * EMITTER_PAYLOAD_WORDS
        if e.type == EMIT_POINT:
            e.position, e.intensity = r.vec(3), r.vec(3)
        elif e.type == EMIT_SPOT:
            e.cutoff_angle, e.beam_width = r.f32(), r.f32()
            e.id_texture = r.u32()
            e.intensity, e.to_world = r.vec(3), r.mat()
        elif e.type == EMIT_DIRECTIONAL:
            e.direction, e.radiance = r.vec(3), r.vec(3)
        elif e.type == EMIT_SUN:
            e.cos_cutoff_angle, e.id_texture = r.f32(), r.u32()
            e.direction, e.radiance = r.vec(3), r.vec(3)
        elif e.type == EMIT_ENVMAP:
            e.id_radiance, e.to_world = r.u32(), r.mat()
        elif e.type == EMIT_CONSTANT:
            e.radiance = r.vec(3)
        else:
            raise ValueError("emitter type")
        r.off = end
        s.emitters.append(e)
    if r.off != len(raw):
        raise ValueError("trailing bytes in MCSD file")
    return s


def load(path) -> Scene:
    with open(path, "rb") as f:
        return loads(f.read())
