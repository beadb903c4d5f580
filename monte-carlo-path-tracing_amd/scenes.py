"""Programmatic test scenes (MCSD builders).

The scene values are those of the reference's example scenes
(resources/scene/cornell-box/scene_v0.6.xml,
resources/scene/volumetric-caustic/scene_v0.6.xml) assembled the way the
reference front end assembles them (src/parser/parser.cpp:94-179: camera,
integrator, textures, BSDFs, media, shapes, emitters; one constant texture
per BSDF parameter in parse order, parser.cpp:651-671; an area light becomes
a pseudo-BSDF of type kAreaLight, parser.cpp:1068-1100).  Matrix literals are
converted with float32 rounding of the decimal strings, like `sscanf("%f")`.
"""
from __future__ import annotations

import numpy as np

from . import mcsd
from .mcsd import (Bsdf, Camera, Emitter, Instance, Integrator, Medium, Scene,
                   Texture)

f32 = np.float32


def mat(text: str) -> np.ndarray:
    vals = [f32(float(v)) for v in text.replace(",", " ").split()]
    assert len(vals) == 16
    return np.array(vals, dtype=np.float32).reshape(4, 4)


def _dot4(a, b):
    # csrt::Dot(Vec4, Vec4): x*x + y*y + z*z + w*w, left to right, float32
    acc = f32(a[0]) * f32(b[0])
    for k in range(1, 4):
        acc = f32(acc + f32(f32(a[k]) * f32(b[k])))
    return acc


def transform_point(m, p):
    """csrt::TransformPoint (mat4.cpp:264-267 + vec4.cpp:93-97)."""
    v = [f32(p[0]), f32(p[1]), f32(p[2]), f32(1)]
    r = [_dot4(m[i], v) for i in range(4)]
    k = f32(1) / r[3]
    return tuple(float(f32(r[i] * k)) for i in range(3))


def transform_vector(m, d):
    """csrt::TransformVector: normalises its result (mat4.cpp:270-273)."""
    v = [f32(d[0]), f32(d[1]), f32(d[2]), f32(0)]
    r = [_dot4(m[i], v) for i in range(3)]
    n = f32(np.sqrt(f32(f32(f32(r[0] * r[0]) + f32(r[1] * r[1])) +
                        f32(r[2] * r[2]))))
    k = f32(1) / n
    return tuple(float(f32(r[i] * k)) for i in range(3))


def camera_from_to_world(m, fov_x, width, height, spp) -> Camera:
    """parser.cpp:345-356: eye/look_at/up from the sensor toWorld matrix."""
    return Camera(spp=spp, width=width, height=height, fov_x=fov_x,
                  eye=transform_point(m, (0, 0, 0)),
                  look_at=transform_point(m, (0, 0, 1)),
                  up=transform_vector(m, (0, 1, 0)))


class _Builder:
    def __init__(self):
        self.s = Scene()
        self.bsdf_ids = {}
        self.medium_ids = {}

    def const_tex(self, rgb) -> int:
        self.s.textures.append(Texture(type=mcsd.TEX_CONSTANT,
                                       color=tuple(float(f32(c)) for c in rgb)))
        return len(self.s.textures) - 1

    def diffuse(self, name, rgb, twosided=True) -> int:
        t = self.const_tex(rgb)
        self.s.bsdfs.append(Bsdf(type=mcsd.BSDF_DIFFUSE, twosided=twosided,
                                 id_diffuse_reflectance=t))
        self.bsdf_ids[name] = len(self.s.bsdfs) - 1
        return self.bsdf_ids[name]

    def dielectric(self, name, int_ior, ext_ior, alpha=0.001, thin=False):
        # parser.cpp:896-923: smooth dielectric = rough model with alpha 0.001
        ru = self.const_tex((alpha,) * 3)
        sr = self.const_tex((1.0,) * 3)
        st = self.const_tex((1.0,) * 3)
        eta = float(f32(int_ior) / f32(ext_ior))
        self.s.bsdfs.append(Bsdf(
            type=mcsd.BSDF_THIN_DIELECTRIC if thin else mcsd.BSDF_DIELECTRIC,
            twosided=True, id_roughness_u=ru, id_roughness_v=ru,
            id_specular_reflectance=sr, id_specular_transmittance=st, eta=eta))
        self.bsdf_ids[name] = len(self.s.bsdfs) - 1
        return self.bsdf_ids[name]

    def area_light(self, radiance) -> int:
        t = self.const_tex(radiance)
        self.s.bsdfs.append(Bsdf(type=mcsd.BSDF_AREA_LIGHT, twosided=False,
                                 weight=1.0, id_radiance=t))
        return len(self.s.bsdfs) - 1

    def shape(self, type_, to_world, id_bsdf, med_int=mcsd.INVALID,
              med_ext=mcsd.INVALID, **kw):
        self.s.instances.append(Instance(type=type_, id_bsdf=id_bsdf,
                                         id_medium_int=med_int,
                                         id_medium_ext=med_ext,
                                         to_world=to_world, **kw))


def cornell_box(width=512, height=512, spp=16) -> Scene:
    """resources/scene/cornell-box/scene_v0.6.xml (BASELINE configs 1 and 2)."""
    b = _Builder()
    b.s.camera = camera_from_to_world(
        mat("-1 0 0 0 0 1 0 1 0 0 -1 6.8 0 0 0 1"), 19.5, width, height, spp)
    b.s.integrator = Integrator(type=mcsd.INTEGRATOR_PATH, depth_max=65,
                                depth_rr=5, pdf_rr=0.95)
    white = (0.725, 0.71, 0.68)
    b.diffuse("LeftWall", (0.63, 0.065, 0.05))
    b.diffuse("RightWall", (0.14, 0.45, 0.091))
    b.diffuse("Floor", white)
    b.diffuse("Ceiling", white)
    b.diffuse("BackWall", white)
    b.diffuse("ShortBox", white)
    b.diffuse("TallBox", white)
    b.diffuse("Light", (0, 0, 0))
    R, C = mcsd.INST_RECTANGLE, mcsd.INST_CUBE
    b.shape(R, mat("0 1 0 0 0 0 2 0 1 0 0 0 0 0 0 1"), b.bsdf_ids["Floor"])
    b.shape(R, mat("-1 0 0 0 0 0 -2 2 0 -1 0 0 0 0 0 1"), b.bsdf_ids["Ceiling"])
    b.shape(R, mat("0 1 0 0 1 0 0 1 0 0 -2 -1 0 0 0 1"), b.bsdf_ids["BackWall"])
    b.shape(R, mat("0 0 2 1 1 0 0 1 0 1 0 0 0 0 0 1"), b.bsdf_ids["RightWall"])
    b.shape(R, mat("0 0 -2 -1 1 0 0 1 0 -1 0 0 0 0 0 1"), b.bsdf_ids["LeftWall"])
    b.shape(C, mat("0.0851643 0.289542 1.31134e-008 0.328631 3.72265e-009 "
                   "1.26563e-008 -0.3 0.3 -0.284951 0.0865363 5.73206e-016 "
                   "0.374592 0 0 0 1"), b.bsdf_ids["ShortBox"])
    b.shape(C, mat("0.286776 0.098229 -2.29282e-015 -0.335439 -4.36233e-009 "
                   "1.23382e-008 -0.6 0.6 -0.0997984 0.282266 2.62268e-008 "
                   "-0.291415 0 0 0 1"), b.bsdf_ids["TallBox"])
    light = b.area_light((17, 12, 4))
    b.shape(R, mat("0.235 0 0 -0.005 0 0 -0.0893 1.98 0 0.19 0 -0.03 0 0 0 1"),
            light)
    return b.s


def grazing_strips(width=128, height=64, spp=1, offset=1000.0, length=8000.0, strip_width=63.5) -> Scene:
    """TEST SCENE for the ordered walk's tie radius: two long coplanar strips (aspect ratio ~125, just below the sliver
    threshold of 128) of different instances and colours that overlap, at coordinates of `offset` .. `offset + length`, seen
    at grazing incidence — where the watertight test's distance is least accurate (a weighted mean of vertex depths with
    ill-conditioned weights) while a flat leaf box's entry distance is exact: which of the two coplanar surfaces the reference
    reports depends on its visiting order and on whether the second one's leaf box still passes.  Inside the production
    radius (5e-6 x largest |coordinate|) the ordered walk replays that and agrees; with the radius shrunk
    (mcpt_testing_set_walk_tie_scale, tests only) it does not — the case the self-guard of mcpt_renderer_create exists for."""
    K, L, hh = float(offset), float(length), float(strip_width)
    s = cornell_box(width, height, spp)
    light = s.instances[-1]
    s.instances = []

    def strip(x0, z0, angle):
        c, sn = np.cos(angle), np.sin(angle)
        pts = np.array([[0, 0], [L, 0], [L, hh], [0, hh]], np.float64)
        return np.stack([x0 + c * pts[:, 0] - sn * pts[:, 1], np.full(4, K), z0 + sn * pts[:, 0] + c * pts[:, 1]], 1).astype(np.float32)
    idx = np.array([[0, 2, 1], [0, 3, 2]], np.uint32)   # facing +y
    s.instances.append(Instance(type=mcsd.INST_MESHES, id_bsdf=0, positions=strip(K, K, 0.0), indices=idx))          # red
    s.instances.append(Instance(type=mcsd.INST_MESHES, id_bsdf=1, positions=strip(K + 3.0, K - 0.3 * hh, 0.002), indices=idx))  # green
    light = Instance(type=mcsd.INST_MESHES, id_bsdf=light.id_bsdf)
    light.positions = np.array([[K, K + 400, K - hh], [K + L, K + 400, K - hh], [K + L, K + 400, K + 2 * hh], [K, K + 400, K + 2 * hh]], np.float32)
    light.indices = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)   # facing -y
    s.instances.append(light)
    s.camera = look_at_camera((K - 50.0, K + 6.0, K + 0.4 * hh), (K + 0.7 * L, K, K + 0.4 * hh), (0, 1, 0), 3.0, width, height, spp)
    return s


def volumetric_caustic(width=1280, height=720, spp=1024, g=0.5) -> Scene:
    """resources/scene/volumetric-caustic/scene_v0.6.xml (BASELINE config 5):
    volpath, HG medium on both sides of the walls, a null-BSDF front wall
    (first <ref> is the medium: parser.cpp:1108-1117), an analytic glass
    sphere, a tiny bright area light."""
    b = _Builder()
    b.s.camera = camera_from_to_world(
        mat("-1 0 1.50996e-007 -1.05697e-006 0 1 0 1 -1.50996e-007 0 -1 7 "
            "0 0 0 1"), 19.5, width, height, spp)
    b.s.integrator = Integrator(type=mcsd.INTEGRATOR_VOLPATH, depth_max=7,
                                depth_rr=5, pdf_rr=0.95)
    b.s.media.append(Medium(sigma_a=(0, 0, 0), sigma_s=(1, 1, 1),
                            phase_type=mcsd.PHASE_HG,
                            g=(float(f32(g)),) * 3))
    med = 0
    white = (0.725, 0.71, 0.68)
    b.diffuse("LeftWall", (0.63, 0.065, 0.05))
    b.diffuse("RightWall", (0.14, 0.45, 0.091))
    b.diffuse("Floor", white)
    b.diffuse("Ceiling", white)
    b.diffuse("BackWall", white)
    b.dielectric("Sphere", 1.5, 1.0)
    b.diffuse("Light", (0, 0, 0))
    R = mcsd.INST_RECTANGLE
    b.shape(R, mat("-4.37114e-008 1 4.37114e-008 0 0 -8.74228e-008 2 0 1 "
                   "4.37114e-008 1.91069e-015 0 0 0 0 1"),
            b.bsdf_ids["Floor"], med, med)
    b.shape(R, mat("-1 -7.64274e-015 1.74846e-007 0 -8.74228e-008 "
                   "8.74228e-008 -2 2 0 -1 -4.37114e-008 0 0 0 0 1"),
            b.bsdf_ids["Ceiling"], med, med)
    b.shape(R, mat("1.91069e-015 1 1.31134e-007 0 1 3.82137e-015 "
                   "-8.74228e-008 1 -4.37114e-008 1.31134e-007 -2 -1 0 0 0 1"),
            b.bsdf_ids["BackWall"], med, med)
    b.shape(R, mat("1.91069e-015 1 -4.37114e-008 0 -1 3.82137e-015 "
                   "-8.74228e-008 1 -4.37114e-008 -4.37114e-008 2 1 0 0 0 1"),
            mcsd.INVALID, med, mcsd.INVALID)
    b.shape(R, mat("4.37114e-008 0 2 1 1 3.82137e-015 -8.74228e-008 1 "
                   "-3.82137e-015 1 -1.31134e-007 0 0 0 0 1"),
            b.bsdf_ids["RightWall"], med, med)
    b.shape(R, mat("-4.37114e-008 8.74228e-008 -2 -1 1 3.82137e-015 "
                   "-8.74228e-008 1 0 -1 -4.37114e-008 0 0 0 0 1"),
            b.bsdf_ids["LeftWall"], med, med)
    b.shape(mcsd.INST_SPHERE, mcsd.IDENTITY.copy(), b.bsdf_ids["Sphere"],
            mcsd.INVALID, med, sphere_radius=float(f32(0.3)),
            sphere_center=(float(f32(-0.22827)), float(f32(1.2)),
                           float(f32(0.152505))))
    light = b.area_light((541127, 381972, 127324))
    b.shape(R, mat("-0.0025 -1.91069e-015 4.37114e-008 -0.005 -2.18557e-010 "
                   "2.18557e-008 -0.5 1.98 0 -0.002 -8.74228e-011 -0.03 "
                   "0 0 0 1"), light, med, med)
    return b.s


# ---------------------------------------------------------------------------
# Synthetic geometry and "kitchen sink" scenes for unit / parity coverage
# ---------------------------------------------------------------------------
def uv_sphere_mesh(n_lat=12, n_lon=24, radius=1.0, center=(0, 0, 0),
                   with_normals=True, with_uv=True):
    """Latitude/longitude sphere as an indexed triangle mesh (y up)."""
    pos, nor, uv, idx = [], [], [], []
    for i in range(n_lat + 1):
        theta = np.pi * i / n_lat
        for j in range(n_lon + 1):
            phi = 2 * np.pi * j / n_lon
            d = np.array([np.sin(theta) * np.cos(phi), np.cos(theta),
                          np.sin(theta) * np.sin(phi)])
            pos.append(np.asarray(center) + radius * d)
            nor.append(d)
            uv.append((j / n_lon, i / n_lat))
    for i in range(n_lat):
        for j in range(n_lon):
            a = i * (n_lon + 1) + j
            b = a + n_lon + 1
            if i != 0:
                idx.append((a, a + 1, b))
            if i != n_lat - 1:
                idx.append((a + 1, b + 1, b))
    return dict(positions=np.array(pos, dtype=np.float32),
                normals=np.array(nor, dtype=np.float32) if with_normals else None,
                texcoords=np.array(uv, dtype=np.float32) if with_uv else None,
                indices=np.array(idx, dtype=np.uint32))


def bumpy_terrain_mesh(n=64, size=4.0, height=0.15, seed=7):
    """Deterministic displaced grid (n x n quads) in the xz plane."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-size / 2, size / 2, n + 1)
    zz, xx = np.meshgrid(xs, xs, indexing="ij")
    yy = height * (np.sin(3.1 * xx) * np.cos(2.3 * zz) + 0.3 * rng.standard_normal(xx.shape))
    pos = np.stack([xx, yy, zz], axis=-1).reshape(-1, 3).astype(np.float32)
    idx = []
    for i in range(n):
        for j in range(n):
            a = i * (n + 1) + j
            idx.append((a, a + n + 1, a + 1))
            idx.append((a + 1, a + n + 1, a + n + 2))
    return dict(positions=pos, normals=None, texcoords=None,
                indices=np.array(idx, dtype=np.uint32))


def procedural_envmap(width=64, height=32, channel=3, seed=3):
    """Smooth sky gradient + a bright blob; float32 [h, w, c] in [0, ~20]."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.linspace(0, 1, height), np.linspace(0, 1, width),
                       indexing="ij")
    base = np.stack([0.3 + 0.5 * (1 - v), 0.4 + 0.4 * (1 - v), 0.6 + 0.3 * (1 - v)], -1)
    blob = 18.0 * np.exp(-((u - 0.3) ** 2 + (v - 0.25) ** 2) / 0.004)
    img = base + blob[..., None] * np.array([1.0, 0.9, 0.7])
    img += 0.02 * rng.random(img.shape)
    if channel == 4:
        img = np.concatenate([img, np.ones_like(img[..., :1])], -1)
    return img.astype(np.float32)


def look_at_camera(eye, target, up, fov_x, width, height, spp) -> Camera:
    return Camera(spp=spp, width=width, height=height, fov_x=fov_x,
                  eye=tuple(float(f32(x)) for x in eye),
                  look_at=tuple(float(f32(x)) for x in target),
                  up=tuple(float(f32(x)) for x in up))


def translate_scale(t=(0, 0, 0), s=(1, 1, 1)) -> np.ndarray:
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = s
    m[:3, 3] = t
    return m


def rot_x(deg) -> np.ndarray:
    a = np.deg2rad(deg)
    m = np.eye(4, dtype=np.float32)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    return m


MATERIALS = ("diffuse", "rough_diffuse_fast", "rough_diffuse_full",
             "conductor", "rough_conductor", "rough_conductor_aniso",
             "dielectric", "rough_dielectric", "thin_dielectric", "plastic",
             "rough_plastic", "bumpy_diffuse", "masked_diffuse")
LIGHTINGS = ("area", "point", "spot", "directional", "sun", "envmap",
             "constant", "mixed")


def material_preview(material="rough_conductor", lighting="envmap", shape="mesh",
                     width=64, height=64, spp=8, integrator="path",
                     depth_max=mcsd.INVALID, medium=False) -> Scene:
    """A matpreview-like scene: checkerboard floor, one test object, one
    lighting set-up.  Covers every BSDF / texture / emitter / primitive kind
    of the hot path; used for oracle-vs-reference and GPU-vs-oracle parity."""
    b = _Builder()
    s = b.s
    s.camera = look_at_camera((0.0, 1.6, 4.2), (0.0, 0.6, 0.0), (0, 1, 0), 38.0,
                              width, height, spp)
    s.integrator = Integrator(
        type=mcsd.INTEGRATOR_VOLPATH if integrator == "volpath" else mcsd.INTEGRATOR_PATH,
        depth_max=depth_max, depth_rr=5, pdf_rr=0.95)
    med = mcsd.INVALID
    if medium:
        s.media.append(Medium(sigma_a=(0.02, 0.03, 0.05), sigma_s=(0.25, 0.2, 0.15),
                              phase_type=mcsd.PHASE_HG, g=(0.3, 0.3, 0.3)))
        med = 0

    # floor: checkerboard diffuse
    s.textures.append(Texture(type=mcsd.TEX_CHECKERBOARD, color0=(0.4, 0.4, 0.4),
                              color1=(0.2, 0.2, 0.2),
                              to_uv=translate_scale(s=(8, 8, 1))))
    s.bsdfs.append(Bsdf(type=mcsd.BSDF_DIFFUSE, twosided=True,
                        id_diffuse_reflectance=len(s.textures) - 1))
    floor_bsdf = len(s.bsdfs) - 1
    b.shape(mcsd.INST_RECTANGLE, (translate_scale(s=(5, 1, 5)) @ rot_x(-90)).astype(np.float32),
            floor_bsdf, med, med)

    def tex(v):
        return b.const_tex((v,) * 3 if np.isscalar(v) else v)

    m = material
    if m == "diffuse":
        bs = Bsdf(type=mcsd.BSDF_DIFFUSE, twosided=True,
                  id_diffuse_reflectance=tex((0.7, 0.3, 0.2)))
    elif m in ("rough_diffuse_fast", "rough_diffuse_full"):
        bs = Bsdf(type=mcsd.BSDF_ROUGH_DIFFUSE, twosided=True,
                  use_fast_approx=m.endswith("fast"),
                  id_diffuse_reflectance=tex((0.6, 0.6, 0.3)), id_roughness=tex(0.4))
    elif m in ("conductor", "rough_conductor", "rough_conductor_aniso"):
        au, av = {"conductor": (0.001, 0.001), "rough_conductor": (0.1, 0.1),
                  "rough_conductor_aniso": (0.05, 0.3)}[m]
        tu = tex(au)
        tv = tu if au == av else tex(av)
        bs = Bsdf(type=mcsd.BSDF_CONDUCTOR, twosided=True, id_roughness_u=tu,
                  id_roughness_v=tv, id_specular_reflectance=tex(1.0),
                  reflectivity=(0.913, 0.922, 0.924), edgetint=(0.05, 0.06, 0.08))
    elif m in ("dielectric", "rough_dielectric", "thin_dielectric"):
        a = 0.1 if m == "rough_dielectric" else 0.001
        tu = tex(a)
        bs = Bsdf(type=mcsd.BSDF_THIN_DIELECTRIC if m == "thin_dielectric" else mcsd.BSDF_DIELECTRIC,
                  twosided=True, id_roughness_u=tu, id_roughness_v=tu,
                  id_specular_reflectance=tex(1.0), id_specular_transmittance=tex((0.9, 1.0, 0.95)),
                  eta=float(f32(1.33) / f32(1.000277)))
    elif m in ("plastic", "rough_plastic"):
        bs = Bsdf(type=mcsd.BSDF_PLASTIC, twosided=True, eta=float(f32(1.5046) / f32(1.000277)),
                  id_roughness=tex(0.1 if m == "rough_plastic" else 0.001),
                  id_diffuse_reflectance=tex((0.2, 0.4, 0.7)), id_specular_reflectance=tex(1.0))
    elif m == "bumpy_diffuse":
        img = procedural_envmap(32, 32, 1 if False else 3, seed=11)
        s.textures.append(Texture(type=mcsd.TEX_BITMAP, width=32, height=32, channel=3,
                                  data=(img / img.max()).reshape(-1), to_uv=translate_scale(s=(3, 3, 1))))
        bs = Bsdf(type=mcsd.BSDF_DIFFUSE, twosided=True, id_bump_map=len(s.textures) - 1,
                  id_diffuse_reflectance=tex((0.6, 0.6, 0.6)))
    elif m == "masked_diffuse":
        img = procedural_envmap(16, 16, 4, seed=5)
        img[..., 3] = (np.indices((16, 16)).sum(0) % 2) * 0.8 + 0.1
        s.textures.append(Texture(type=mcsd.TEX_BITMAP, width=16, height=16, channel=4,
                                  data=img.reshape(-1)))
        bs = Bsdf(type=mcsd.BSDF_DIFFUSE, twosided=True, id_opacity=len(s.textures) - 1,
                  id_diffuse_reflectance=tex((0.3, 0.7, 0.3)))
    else:
        raise ValueError(material)
    s.bsdfs.append(bs)
    obj_bsdf = len(s.bsdfs) - 1

    obj_int = mcsd.INVALID
    if shape == "mesh":
        g = uv_sphere_mesh(10, 20, 0.6, (0, 0.6, 0))
        b.shape(mcsd.INST_MESHES, mcsd.IDENTITY.copy(), obj_bsdf, obj_int, med,
                positions=g["positions"], normals=g["normals"],
                texcoords=g["texcoords"], indices=g["indices"])
    elif shape == "flat_mesh":   # no normals / no uv -> synthesised
        g = uv_sphere_mesh(8, 16, 0.6, (0, 0.6, 0), with_normals=False, with_uv=False)
        b.shape(mcsd.INST_MESHES, mcsd.IDENTITY.copy(), obj_bsdf, obj_int, med,
                positions=g["positions"], indices=g["indices"])
    elif shape == "sphere":
        b.shape(mcsd.INST_SPHERE, translate_scale(t=(0, 0.6, 0)), obj_bsdf, obj_int, med,
                sphere_radius=0.6, sphere_center=(0.0, 0.0, 0.0))
    elif shape == "cube":
        b.shape(mcsd.INST_CUBE, translate_scale(t=(0, 0.5, 0), s=(0.5, 0.5, 0.5)), obj_bsdf, obj_int, med)
    elif shape == "disk":
        b.shape(mcsd.INST_DISK, (translate_scale(t=(0, 0.7, 0), s=(1.6, 1.6, 1.6)) @ rot_x(-60)).astype(np.float32),
                obj_bsdf, obj_int, med)
    elif shape == "cylinder":
        b.shape(mcsd.INST_CYLINDER, mcsd.IDENTITY.copy(), obj_bsdf, obj_int, med,
                cyl_radius=0.4, cyl_p0=(0.0, 0.05, 0.0), cyl_p1=(0.3, 1.2, 0.1))
    else:
        raise ValueError(shape)

    lights = {"mixed": ("area", "point", "directional", "constant")}.get(lighting, (lighting,))
    for l in lights:
        if l == "area":
            light = b.area_light((12, 11, 9))
            b.shape(mcsd.INST_RECTANGLE,
                    (translate_scale(t=(0.8, 2.6, 0.6), s=(0.5, 0.5, 0.5)) @ rot_x(90)).astype(np.float32),
                    light, med, med)
        elif l == "point":
            s.emitters.append(Emitter(type=mcsd.EMIT_POINT, position=(-1.5, 2.5, 1.5),
                                      intensity=(30, 30, 30)))
        elif l == "spot":
            tw = np.linalg.inv(_look_at_lh((1.5, 3.0, 1.5), (0, 0.5, 0), (0, 1, 0))).astype(np.float32)
            s.emitters.append(Emitter(type=mcsd.EMIT_SPOT, intensity=(60, 55, 50), to_world=tw,
                                      cutoff_angle=float(np.deg2rad(25)), beam_width=float(np.deg2rad(18))))
        elif l == "directional":
            d = np.array([0.3, -0.8, -0.5])
            d = d / np.linalg.norm(d)
            s.emitters.append(Emitter(type=mcsd.EMIT_DIRECTIONAL, direction=tuple(float(f32(x)) for x in d),
                                      radiance=(3, 3, 2.8)))
        elif l == "sun":
            img = procedural_envmap(32, 16, 3, seed=9)
            s.textures.append(Texture(type=mcsd.TEX_BITMAP, width=32, height=16, channel=3,
                                      data=img.reshape(-1)))
            d = np.array([-0.4, -0.7, -0.6])
            d = d / np.linalg.norm(d)
            s.emitters.append(Emitter(type=mcsd.EMIT_SUN, cos_cutoff_angle=float(np.cos(np.deg2rad(2.0))),
                                      id_texture=len(s.textures) - 1,
                                      direction=tuple(float(f32(x)) for x in d), radiance=(40, 38, 33)))
        elif l == "envmap":
            img = procedural_envmap(64, 32, 4, seed=3)
            s.textures.append(Texture(type=mcsd.TEX_BITMAP, width=64, height=32, channel=4,
                                      data=img.reshape(-1)))
            s.emitters.append(Emitter(type=mcsd.EMIT_ENVMAP, id_radiance=len(s.textures) - 1,
                                      to_world=rot_x(10)))
        elif l == "constant":
            s.emitters.append(Emitter(type=mcsd.EMIT_CONSTANT, radiance=(0.5, 0.6, 0.8)))
        else:
            raise ValueError(lighting)
    return s


def _look_at_lh(eye, target, up):
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    front = (target - eye) / np.linalg.norm(target - eye)
    right = np.cross(up, front)
    right /= np.linalg.norm(right)
    up = np.cross(front, right)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = right, up, front
    m[0, 3], m[1, 3], m[2, 3] = -right @ eye, -up @ eye, -front @ eye
    return m


def terrain_scene(n=96, width=96, height=64, spp=4) -> Scene:
    """A mid-size mesh (2*n*n triangles) under a directional light: the
    divergent-traversal stand-in for the dragon configuration."""
    b = _Builder()
    s = b.s
    s.camera = look_at_camera((0.0, 2.2, 4.5), (0.0, 0.0, 0.0), (0, 1, 0), 40.0, width, height, spp)
    s.integrator = Integrator(type=mcsd.INTEGRATOR_PATH, depth_max=17, depth_rr=5, pdf_rr=0.95)
    g = bumpy_terrain_mesh(n)
    bs = b.diffuse("ground", (0.6, 0.55, 0.5))
    b.shape(mcsd.INST_MESHES, mcsd.IDENTITY.copy(), bs, positions=g["positions"], indices=g["indices"])
    ball = b.diffuse("ball", (0.3, 0.5, 0.7))
    sp = uv_sphere_mesh(24, 48, 0.5, (0.2, 0.8, 0.3))
    b.shape(mcsd.INST_MESHES, mcsd.IDENTITY.copy(), ball, positions=sp["positions"],
            normals=sp["normals"], texcoords=sp["texcoords"], indices=sp["indices"])
    d = np.array([0.1886, -0.6923, -0.6965])
    s.emitters.append(Emitter(type=mcsd.EMIT_DIRECTIONAL, direction=tuple(float(f32(x)) for x in d),
                              radiance=(10, 10, 10)))
    return s


def blob_mesh(n_lat=128, n_lon=256, radius=1.0, center=(0, 0, 0), bumps=0.18, seed=1):
    """A lumpy closed surface: a latitude/longitude sphere displaced along its normals
    by a few spherical waves (vectorised: these go up to hundreds of thousands of
    triangles).  No normals / uv: the flat-shaded path, like the dragon OBJ files."""
    rng = np.random.default_rng(seed)
    theta = np.pi * np.arange(n_lat + 1)[:, None] / n_lat
    phi = 2 * np.pi * np.arange(n_lon + 1)[None, :] / n_lon
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) * np.ones_like(phi), np.sin(theta) * np.sin(phi)], -1)
    r = np.ones(d.shape[:2])
    for _ in range(6):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        freq, phase = rng.integers(3, 14), rng.random() * 6.28
        r += bumps / 6 * np.sin(freq * np.arccos(np.clip(d @ axis, -1, 1)) + phase) * (1 + rng.random())
    pos = (np.asarray(center) + radius * r[..., None] * d).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = (i * (n_lon + 1) + j).ravel()
    b = a + n_lon + 1
    upper = np.stack([a, a + 1, b], 1)[i.ravel() != 0]
    lower = np.stack([a + 1, b + 1, b], 1)[i.ravel() != n_lat - 1]
    return dict(positions=pos, normals=None, texcoords=None,
                indices=np.concatenate([upper, lower]).astype(np.uint32))


def blob_field_scene(n_blobs=12, n_lat=128, n_lon=256, width=1280, height=720, spp=256) -> Scene:
    """Stand-in for BASELINE config 3 (dragon/scene.xml, whose large OBJ files are not
    shipped with the reference): the same integrator settings (path, max depth 17),
    diffuse materials and directional light (direction 0.1886 -0.6923 -0.6965,
    irradiance 10) over a ground mesh and `n_blobs` lumpy closed meshes of
    2*n_lat*n_lon - 2*n_lon triangles each (12 x 65 024 + ground = 0.8 M triangles by
    default): deep, overlapping geometry with high depth complexity."""
    b = _Builder()
    s = b.s
    s.camera = look_at_camera((0.0, 3.2, 7.5), (0.0, 0.6, 0.0), (0, 1, 0), 35.0, width, height, spp)
    s.integrator = Integrator(type=mcsd.INTEGRATOR_PATH, depth_max=17, depth_rr=5, pdf_rr=0.95)
    ground = b.diffuse("ground", (0.456263,) * 3)
    g = bumpy_terrain_mesh(96, size=14.0, height=0.05)
    b.shape(mcsd.INST_MESHES, mcsd.IDENTITY.copy(), ground, positions=g["positions"], indices=g["indices"])
    body = b.diffuse("body", (0.79311,) * 3)
    rng = np.random.default_rng(42)
    for k in range(n_blobs):
        ang = 2 * np.pi * k / max(n_blobs, 1) + 0.3 * rng.random()
        dist = 0.0 if k == 0 else 1.4 + 1.9 * (k % 3) * 0.5 + 0.4 * rng.random()
        centre = (dist * np.cos(ang), 0.55 + 0.5 * rng.random(), dist * np.sin(ang))
        m = blob_mesh(n_lat, n_lon, radius=0.45 + 0.35 * rng.random(), center=centre, seed=100 + k)
        b.shape(mcsd.INST_MESHES, mcsd.IDENTITY.copy(), body, positions=m["positions"], indices=m["indices"])
    d = np.array([0.1886, -0.6923, -0.6965])
    s.emitters.append(Emitter(type=mcsd.EMIT_DIRECTIONAL, direction=tuple(float(f32(x)) for x in d),
                              radiance=(10, 10, 10)))
    return s
