"""End to end on the reference's own scene files (where /root/reference exists):
XML front end -> configuration -> the compiled reference, the oracle and the host
build of the product's kernel body must produce the same frame, bit for bit.
Covers the .serialized meshes, the PIZ environment map and the named conductor of
the material-preview scenes (BASELINE config 4) and the medium of
volumetric-caustic (config 5) at a reduced film size."""
import os
import sys

import numpy as np
import pytest

REF_SCENES = "/root/reference/resources/scene"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference scene files not present")


@pytest.fixture(scope="module")
def emulator():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import emu
    return emu.Emulator()


@pytest.mark.parametrize("xml", ["matpreview/rough_conductor.xml", "matpreview/rough_dielectric.xml",
                                 "volumetric-caustic/scene_v0.6_hg.xml"])
def test_scene_file_renders_identically_everywhere(pkg, oracle, reference, emulator, tmp_path, xml):
    w, h, spp = 48, 36, 3
    cfg = pkg.capi.Config.load_xml(os.path.join(REF_SCENES, xml)).set_film(w, h, spp)
    path = tmp_path / "scene.mcsd"
    cfg.save_mcsd(path)
    want, _ = reference.render(path, w, h)
    got_oracle, _ = oracle.render(path)
    got_kernel_body, _ = emulator.render(path, w, h)
    assert np.isfinite(want).all() and want.max() > 0
    np.testing.assert_array_equal(got_oracle, want)
    np.testing.assert_array_equal(got_kernel_body, want)


@pytest.mark.parametrize("scene", ["classroom", "dining-room"])
def test_textured_scene_renders_identically_everywhere(pkg, oracle, reference, emulator, tmp_path, scene):
    """classroom and dining-room (dozens of OBJ meshes, JPEG / PNG textures, rough plastic /
    conductor / dielectric materials) with their out-of-scope `sunsky` emitter replaced by a
    constant one.  Besides being the largest instance counts tested, classroom contains a
    triangle with identical texture coordinates at all corners: NaN shading frame, NaN
    rays (see tests/test_edge_cases.py, nan_shading_frame)."""
    import re
    src = f"{REF_SCENES}/{scene}"
    text = re.sub(r'<emitter type="sunsky".*?</emitter>',
                  '<emitter type="constant"><rgb name="radiance" value="1"/></emitter>',
                  open(f"{src}/scene_v0.6.xml").read(), flags=re.S)
    for sub in ("models", "textures"):
        os.symlink(f"{src}/{sub}", tmp_path / sub)
    (tmp_path / "scene.xml").write_text(text)
    w, h, spp = 48, 27, 2
    cfg = pkg.capi.Config.load_xml(tmp_path / "scene.xml").set_film(w, h, spp)
    path = tmp_path / "scene.mcsd"
    cfg.save_mcsd(path)
    want, _ = reference.render(path, w, h)
    got_oracle, _ = oracle.render(path)
    np.testing.assert_array_equal(got_oracle, want)
    for variant in (-1, emulator.REFERENCE):
        got, _ = emulator.render(path, w, h, variant=variant)
        np.testing.assert_array_equal(got, want)
