"""Generates tests/golden/stb_vectors.npz with the reference's vendored image library
(extern/stb, compiled by oracle/Makefile into oracle/_ref/libstb_ref.so): file bytes of a few
small JPEG / PNG / Radiance images and what stb_image decodes them to, and random float images
with what stbir_resize_float_linear makes of them at the sizes the reference's environment-map
rule would ask for.  Data only — the test that reads it is
tests/test_xml_frontend.py::test_image_vectors_of_the_reference_library.  Run where
/root/reference exists:  python tests/golden/make_stb_golden.py"""
import io
import os
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import checkers  # noqa: E402


def main():
    from test_xml_frontend import _hdr_bytes, _rgbe
    stb = checkers.Stb()
    rng = np.random.default_rng(2024)
    out = {}
    tmp = tempfile.mkdtemp()

    def decode(name, raw, float_):
        path = os.path.join(tmp, name)
        open(path, "wb").write(raw)
        return stb.loadf(path) if float_ else stb.load8(path)

    for k, (sub, (w, h), grey, quality) in enumerate([(0, (24, 16), False, 90), (2, (37, 21), False, 75),
                                                      (1, (33, 18), False, 85), (0, (19, 30), True, 90),
                                                      (2, (50, 34), False, 40)]):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([128 + 100 * np.sin(xx * 0.21), 128 + 90 * np.cos(yy * 0.17), 128 + 80 * np.sin((xx + yy) * 0.11)], -1)
        arr = np.clip(base + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
        img = Image.fromarray(arr[..., 0], "L") if grey else Image.fromarray(arr, "RGB")
        buf = io.BytesIO()
        img.save(buf, format="JPEG", quality=quality, **({} if grey else {"subsampling": sub}))
        out[f"jpeg{k}_file"] = np.frombuffer(buf.getvalue(), np.uint8)
        out[f"jpeg{k}_pixels"] = decode(f"j{k}.jpg", buf.getvalue(), False)
    for k, mode in enumerate(["RGB", "RGBA", "L", "P"]):
        arr = rng.integers(0, 256, (11, 17, 4), dtype=np.uint8)
        img = Image.fromarray(arr, "RGBA").convert(mode)
        buf = io.BytesIO()
        img.save(buf, format="PNG")
        out[f"png{k}_file"] = np.frombuffer(buf.getvalue(), np.uint8)
        out[f"png{k}_pixels"] = decode(f"p{k}.png", buf.getvalue(), False)
    img = (rng.random((9, 23, 3)) ** 3 * 40).astype(np.float32)
    img[3, 2:20] = 0.5
    for k, rle in enumerate([False, True]):
        raw = _hdr_bytes(_rgbe(img), rle)
        out[f"hdr{k}_file"] = np.frombuffer(raw, np.uint8)
        out[f"hdr{k}_pixels"] = decode(f"h{k}.hdr", raw, True)
    for k, (w, h, c, ow) in enumerate([(97, 41, 3, 32), (64, 32, 1, 12), (120, 60, 3, 82), (50, 25, 4, 7)]):
        src = (rng.random((h, w, c)) * 3).astype(np.float32)
        if c == 4:
            src[..., 3] = 1.0          # environment maps carry alpha 1
        oh = ow * h // w
        out[f"resize{k}_in"] = src
        out[f"resize{k}_out"] = stb.resize(src, ow, oh)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stb_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
