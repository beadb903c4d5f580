#!/usr/bin/env python3
"""Build box only (needs /root/reference): translates the scene files of the BASELINE configurations with
the product's XML front end and stores the resulting renderer configurations (MCSD, gzip) under
monte-carlo-path-tracing_amd/baseline_scenes/, so that the GPU box — which has no /root/reference — can
run every BASELINE configuration (tests/test_baseline_configs.py, bench.py --workload).

These are DATA fixtures: the configuration records the front end produced from the reference's scene
files (camera, materials, meshes, environment map), not source text.

dragon/scene.xml names four OBJ files the reference repository does not ship.  The fixture holds the scene
with its twelve real meshes and a six-triangle placeholder where each missing mesh goes;
dragon_placeholders.json maps the placeholder instances to the missing file names, and
workloads.config("dragon") puts the stand-ins of standins/dragon.txt there
(mcpt_config_set_instance_standin) — tests/test_baseline_configs.py checks that the result is byte for
byte what loading the XML with that table gives."""
import gzip
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENES = "/root/reference/resources/scene/"
OUT = os.path.join(ROOT, "monte-carlo-path-tracing_amd", "baseline_scenes")
JOBS = [("matpreview_rough_conductor", "matpreview/rough_conductor.xml"),
        ("matpreview_rough_dielectric", "matpreview/rough_dielectric.xml"),
        ("volumetric_caustic", "volumetric-caustic/scene_v0.6.xml")]
PLACEHOLDER = " blob 0 0 0 1 1 1 2 3 1 0"


def store(cfg, name):
    tmp = os.path.join(OUT, name + ".mcsd")
    cfg.save_mcsd(tmp)
    with open(tmp, "rb") as f, gzip.GzipFile(tmp + ".gz", "wb", compresslevel=9, mtime=0) as g:
        g.write(f.read())
    os.remove(tmp)
    print(name, cfg.film(), os.path.getsize(tmp + ".gz"), "bytes")


def main():
    from _pkg import load_package
    pkg = load_package()
    os.makedirs(OUT, exist_ok=True)
    for name, xml in JOBS:
        store(pkg.capi.Config.load_xml(SCENES + xml), name)
    # dragon: SURVEY.md section 8c's tangent pin (MCPT_MESH_TANGENTS=uv: no per-vertex tangents, the commit builds the
    # reference's own UV-derived frame, scene.cpp:63-80) CANNOT be this fixture's setting: every one of the twelve shipped OBJ
    # files has `vt 0.000000 0.000000` on all vertices, so that rule divides by a zero UV area and every real triangle gets a
    # NaN frame — NaN scattered directions, rays that pass every box test and walk the whole hierarchy (measured on an MI355X
    # with such a fixture: 53 s for a 320 x 180 x 16 film).  The reference binary itself reads these files through assimp,
    # whose CalcTangentSpace substitutes default axes for a degenerate UV triangle; the fixture keeps the restatement of that
    # (mesh_postprocess.cpp, unpinned).  tests/test_baseline_configs.py::test_dragon_real_meshes_have_no_usable_uvs pins the
    # fact.  The stand-ins (valid UVs) carry no tangents and do get the UV-derived frame.
    # Which <shape> (= instance, in file order) names which missing file:
    xml = open(SCENES + "dragon/scene.xml").read()
    files = re.findall(r'<shape type="obj".*?name="filename" value="([^"]+)"', xml, flags=re.S)
    missing = {i: f for i, f in enumerate(files) if not os.path.exists(SCENES + "dragon/" + f)}
    table = "".join(f + PLACEHOLDER + "\n" for f in missing.values())
    store(pkg.capi.Config.load_xml(SCENES + "dragon/scene.xml", table), "dragon_real_meshes")
    json.dump({str(i): f for i, f in missing.items()}, open(os.path.join(OUT, "dragon_placeholders.json"), "w"), indent=1)
    print("dragon placeholders:", missing)


if __name__ == "__main__":
    main()
