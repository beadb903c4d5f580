#!/usr/bin/env python3
"""Build box only (needs /root/reference): the reference's OTHER shipped scenes (not BASELINE configurations) through the XML
front end -> scratch/other/*.mcsd.gz (git-ignored; travels to the GPU box with gpurun), for tools/rule_vs_calibrated.py.
`sunsky` emitters (out of scope: they need the vendored Hosek-Wilkie tables) are replaced by a constant one, as in
tests/test_reference_scenes.py; scenes whose files the reference does not ship (.MISSING_LARGE_BLOBS) are reported and left out."""
import gzip
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENES = "/root/reference/resources/scene/"
JOBS = [("classroom", "classroom/scene_v0.6.xml"), ("dining-room", "dining-room/scene_v0.6.xml"), ("box", "box/scene_v0.6.xml"),
        ("lte-orb-rough-glass", "lte-orb/rough_glass.xml"), ("lte-orb-silver", "lte-orb/silver.xml"),
        ("material-testball", "material-testball/scene_v0.6.xml"), ("mercury-rough", "mercury/rough_diffuse.xml"),
        ("mercury-smooth", "mercury/smooth_diffuse.xml"), ("matpreview-rough-plastic", "matpreview/rough_plastic.xml"),
        ("matpreview-thin-dielectric", "matpreview/thin_dielectric.xml")]


def main():
    from _pkg import load_package
    pkg = load_package()
    out_dir = os.path.join(ROOT, "scratch", "other")
    os.makedirs(out_dir, exist_ok=True)
    for name, xml in JOBS:
        src = os.path.dirname(SCENES + xml)
        text = re.sub(r'<emitter type="(sunsky|sun|sky)".*?</emitter>', '<emitter type="constant"><rgb name="radiance" value="1"/></emitter>',
                      open(SCENES + xml).read(), flags=re.S)
        with tempfile.TemporaryDirectory() as tmp:
            for sub in os.listdir(src):
                os.symlink(os.path.join(src, sub), os.path.join(tmp, sub))
            open(os.path.join(tmp, "scene.xml"), "w").write(text)
            try:
                cfg = pkg.capi.Config.load_xml(os.path.join(tmp, "scene.xml"))
            except pkg.capi.McptError as e:
                print(name, "NOT CONVERTED:", str(e).splitlines()[-1][:160])
                continue
            raw = os.path.join(out_dir, name + ".mcsd")
            cfg.save_mcsd(raw)
            with open(raw, "rb") as f, gzip.GzipFile(raw + ".gz", "wb", compresslevel=6, mtime=0) as g:
                g.write(f.read())
            os.remove(raw)
            print(name, cfg.film(), os.path.getsize(raw + ".gz") >> 10, "KiB")


if __name__ == "__main__":
    main()
