#!/usr/bin/env python3
"""Build box only (needs /root/reference): translates the reference's scene files with
the XML front end and stores the configurations as scratch/real/*.mcsd (git-ignored;
they travel to the GPU box with gpurun), for tests/full_size_parity.py and
tools/render_scene.py.  No GPU needed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENES = "/root/reference/resources/scene/"
JOBS = [("matpreview_rough_conductor", "matpreview/rough_conductor.xml"),
        ("matpreview_rough_dielectric", "matpreview/rough_dielectric.xml"),
        ("matpreview_rough_plastic", "matpreview/rough_plastic.xml"),
        ("matpreview_thin_dielectric", "matpreview/thin_dielectric.xml"),
        ("volumetric_caustic", "volumetric-caustic/scene_v0.6.xml"),
        ("volumetric_caustic_hg", "volumetric-caustic/scene_v0.6_hg.xml")]


def main():
    from _pkg import load_package
    pkg = load_package()
    out_dir = os.path.join(ROOT, "scratch", "real")
    os.makedirs(out_dir, exist_ok=True)
    for name, xml in JOBS:
        cfg = pkg.capi.Config.load_xml(SCENES + xml)
        cfg.save_mcsd(os.path.join(out_dir, name + ".mcsd"))
        print(name, cfg.film())


if __name__ == "__main__":
    main()
