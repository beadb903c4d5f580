"""The renderer configurations BASELINE.json names, by short name, at their stated film sizes.

    cornell         configs[0/1]  cornell-box 512x512 spp 256 (spp 16 for the CPU plumbing case)
    dragon          configs[2]    dragon/scene.xml 1280x720 spp 256 — STAND-IN geometry for the four OBJ files the
                                  reference repository does not ship (standins/dragon.txt: a body and two wings
                                  fitted to the silhouette of the reference's own render, 22 % of the film hit):
                                  51 140 real + 794 668 procedural triangles.  Tangent frames: the stand-ins carry
                                  none and get the reference's own UV-derived frame (scene.cpp:63-80, SURVEY.md
                                  section 8c's pin); the twelve real OBJ files have `vt 0 0` on every vertex, for
                                  which that rule gives NaN frames (tools/make_baseline_scenes.py), so they keep the
                                  restated importer's tangents (mesh_postprocess.cpp)
    matpreview-rc   configs[3]    matpreview/rough_conductor.xml 1024x1024 spp 512
    matpreview-rd   configs[3]    matpreview/rough_dielectric.xml 1024x1024 spp 512
    volumetric      configs[4]    volumetric-caustic/scene_v0.6.xml 1280x720 spp 1024

Scenes come from baseline_scenes/*.mcsd.gz (tools/make_baseline_scenes.py: the product's XML front end
run on the reference's scene files), so they exist on a box without the reference repository."""
import gzip
import json
import os

from . import capi

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = os.path.join(HERE, "baseline_scenes")
DRAGON_STANDINS = os.path.join(HERE, "standins", "dragon.txt")

#            name             fixture                         film (w, h, spp)    BASELINE.json configs index
WORKLOADS = {"cornell":       (None,                          (512, 512, 256),    1),
             "dragon":        ("dragon_real_meshes",          (1280, 720, 256),   2),
             "matpreview-rc": ("matpreview_rough_conductor",  (1024, 1024, 512),  3),
             "matpreview-rd": ("matpreview_rough_dielectric", (1024, 1024, 512),  3),
             "volumetric":    ("volumetric_caustic",          (1280, 720, 1024),  4)}

DESCRIPTION = {"cornell": "cornell-box.xml 512x512 spp=256",
               "dragon": "dragon/scene.xml 1280x720 spp=256 (12 real meshes + stand-ins for the 4 unshipped OBJ files, 845808 triangles)",
               "matpreview-rc": "matpreview/rough_conductor.xml 1024x1024 spp=512",
               "matpreview-rd": "matpreview/rough_dielectric.xml 1024x1024 spp=512",
               "volumetric": "volumetric-caustic/scene_v0.6.xml 1280x720 spp=1024"}


def _fixture(name):
    with gzip.open(os.path.join(SCENES, name + ".mcsd.gz"), "rb") as f:
        return capi.Config.from_mcsd_bytes(f.read())


def standin_lines(table_path=DRAGON_STANDINS):
    """file name -> its table lines (several lines with one name = the parts of one mesh), newline-joined"""
    lines = {}
    for line in open(table_path):
        tok = line.split()
        if tok and not tok[0].startswith("#"):
            lines[tok[0]] = (lines[tok[0]] + "\n" if tok[0] in lines else "") + line.strip()
    return lines


def config(name, width=0, height=0, spp=0):
    """capi.Config of a BASELINE workload; width / height / spp override the stated film."""
    fixture, film, _ = WORKLOADS[name]
    if fixture is None:
        cfg = capi.Config.builtin("cornell-box")
    else:
        cfg = _fixture(fixture)
        if name == "dragon":
            lines = standin_lines()
            placeholders = json.load(open(os.path.join(SCENES, "dragon_placeholders.json")))
            for instance, file_name in sorted(placeholders.items(), key=lambda kv: int(kv[0])):
                cfg.set_instance_standin(int(instance), lines[file_name])
    return cfg.set_film(width or film[0], height or film[1], spp or film[2])


def samples(name):
    w, h, spp = WORKLOADS[name][1]
    return w * h * spp
