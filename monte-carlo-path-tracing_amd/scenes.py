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
