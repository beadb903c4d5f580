"""Test helper: reads the EXR files the product writes (image_io.cpp, WriteExr)."""
import numpy as np


def read_exr_zip(path):
    """Minimal reader for the writer's own layout: single-part scanline file, FLOAT channels
    B G R, ZIP blocks of 16 lines (a block that did not shrink is stored raw)."""
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:4] == b"\x76\x2f\x31\x01" and struct.unpack("<I", raw[4:8])[0] == 2
    at, attrs = 8, {}
    while raw[at] != 0:
        end = raw.index(b"\0", at)
        name = raw[at:end].decode()
        at = end + 1
        end = raw.index(b"\0", at)
        at = end + 1
        size = struct.unpack("<i", raw[at:at + 4])[0]
        attrs[name] = raw[at + 4:at + 4 + size]
        at += 4 + size
    at += 1
    assert attrs["compression"] == b"\x03" and attrs["lineOrder"] == b"\x00"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    n_blocks = (h + 15) // 16
    offsets = struct.unpack(f"<{n_blocks}Q", raw[at:at + 8 * n_blocks])
    out = np.zeros((h, w, 3), np.float32)
    for k, off in enumerate(offsets):
        y, size = struct.unpack("<ii", raw[off:off + 8])
        assert y == 16 * k
        lines = min(16, h - y)
        n = lines * w * 12
        data = raw[off + 8:off + 8 + size]
        if size < n:
            t = np.frombuffer(zlib.decompress(data), np.uint8).astype(np.int32)
            assert len(t) == n
            t = (np.cumsum(t - 128) + 128) % 256                      # undo the delta predictor
            half = (n + 1) // 2
            b = np.empty(n, np.uint8)
            b[0::2], b[1::2] = t[:half], t[half:]                     # undo the even / odd split
            data = b.tobytes()
        planes = np.frombuffer(data, "<f4").reshape(lines, 3, w)      # per line: B, G, R
        out[y:y + lines] = planes[:, ::-1, :].transpose(0, 2, 1)
    return out
