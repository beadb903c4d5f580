import numpy as np


def test_python_roundtrip(pkg):
    for scene in (pkg.scenes.cornell_box(16, 16, 1), pkg.scenes.volumetric_caustic(16, 9, 1),
                  pkg.scenes.material_preview("rough_conductor_aniso", "mixed", "mesh", 8, 8, 1),
                  pkg.scenes.material_preview("bumpy_diffuse", "spot", "cylinder", 8, 8, 1),
                  pkg.scenes.material_preview("thin_dielectric", "sun", "disk", 8, 8, 1)):
        raw = pkg.mcsd.dumps(scene)
        assert pkg.mcsd.dumps(pkg.mcsd.loads(raw)) == raw


def test_cxx_reader_agrees(pkg, oracle, mcsd_file):
    scene = pkg.scenes.cornell_box(24, 12, 3)
    _, info = oracle.render(mcsd_file(scene))
    assert (info["width"], info["height"], info["spp"]) == (24, 12, 3)


def test_truncated_file_is_rejected(pkg, oracle, tmp_path):
    raw = pkg.mcsd.dumps(pkg.scenes.cornell_box(8, 8, 1))
    bad = tmp_path / "bad.mcsd"
    bad.write_bytes(raw[: len(raw) // 2])
    try:
        oracle.render(str(bad))
    except RuntimeError as e:
        assert "MCSD" in str(e)
    else:
        raise AssertionError("truncated MCSD accepted")
