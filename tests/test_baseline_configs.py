"""Every configuration BASELINE.json names, from the committed fixtures
(monte-carlo-path-tracing_amd/baseline_scenes/, made by tools/make_baseline_scenes.py from the reference's
scene files with the product's XML front end).

CPU part: the fixtures load, and — where /root/reference exists — are byte for byte what the XML front end
produces today (for dragon: XML + stand-in table == fixture + mcpt_config_set_instance_standin).
GPU part (-m gpu): each configuration against the oracle at a reduced film, and at its FULL film size
through size-independent properties (finite, within [0, 1], sample count, deterministic, tile-composable)."""
import hashlib
import os

import numpy as np
import pytest

REF_SCENES = "/root/reference/resources/scene/"
XML = {"dragon": "dragon/scene.xml", "matpreview-rc": "matpreview/rough_conductor.xml",
       "matpreview-rd": "matpreview/rough_dielectric.xml", "volumetric": "volumetric-caustic/scene_v0.6.xml"}
NAMES = ["cornell", "dragon", "matpreview-rc", "matpreview-rd", "volumetric"]


def _bytes(cfg, tmp_path, name):
    p = str(tmp_path / (name + ".mcsd"))
    cfg.save_mcsd(p)
    return open(p, "rb").read()


@pytest.mark.parametrize("name", NAMES)
def test_fixture_loads_with_the_stated_film(pkg, name):
    cfg = pkg.workloads.config(name)
    assert cfg.film() == pkg.workloads.WORKLOADS[name][1]


def test_dragon_has_its_real_meshes_and_the_standins(pkg, tmp_path):
    cfg = pkg.workloads.config("dragon", 64, 36, 1)
    scene = pkg.mcsd.loads(_bytes(cfg, tmp_path, "dragon"))
    tris = [np.asarray(i.indices).size // 3 for i in scene.instances]
    assert len(tris) == 16 and sum(tris) == 845808
    assert sum(t for k, t in enumerate(tris) if k not in (4, 5, 6, 7)) == 51140  # the twelve real OBJ files


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="needs the reference's scene files")
@pytest.mark.parametrize("name", ["dragon", "matpreview-rc", "matpreview-rd", "volumetric"])
def test_fixture_is_what_the_xml_front_end_produces(pkg, tmp_path, name):
    film = pkg.workloads.WORKLOADS[name][1]
    standins = open(pkg.workloads.DRAGON_STANDINS).read() if name == "dragon" else None
    direct = pkg.capi.Config.load_xml(REF_SCENES + XML[name], standins).set_film(*film)
    assert _bytes(direct, tmp_path, "a") == _bytes(pkg.workloads.config(name), tmp_path, "b")


def test_dragon_real_meshes_have_no_usable_uvs(pkg, tmp_path):
    """Why SURVEY.md section 8c's tangent pin (MCPT_MESH_TANGENTS=uv, the reference's own UV-derived frame, scene.cpp:63-80)
    is not this fixture's setting: all twelve shipped OBJ files of dragon/scene.xml carry `vt 0 0` on every vertex, so every
    real triangle has zero UV area and that rule's 1 / (uv area) is a division by zero (NaN frames, NaN rays that walk the
    whole hierarchy).  The stand-ins have proper UVs and no per-vertex tangents: they do get the UV-derived frame."""
    cfg = pkg.workloads.config("dragon", 64, 36, 1)
    scene = pkg.mcsd.loads(_bytes(cfg, tmp_path, "dragon"))
    for k, inst in enumerate(scene.instances):
        uv = np.asarray(inst.texcoords, np.float64).reshape(-1, 2)
        idx = np.asarray(inst.indices).reshape(-1, 3)
        d1, d2 = uv[idx[:, 1]] - uv[idx[:, 0]], uv[idx[:, 2]] - uv[idx[:, 0]]
        flat = (d1[:, 1] * d2[:, 0] - d1[:, 0] * d2[:, 1]) == 0
        if k in (4, 5, 6, 7):   # stand-ins
            assert not flat.any() and np.asarray(inst.tangents).size == 0
        else:
            assert flat.all() and np.asarray(inst.tangents).size == np.asarray(inst.positions).size


def test_dragon_standins_cover_the_film_like_the_reference_render(pkg, oracle, tmp_path):
    """The four OBJ files of dragon/scene.xml that the reference repository does not ship are replaced by stand-ins;
    Msamples/s on that scene only mean something if the stand-ins fill the film like the real dragon does.  The
    reference's own render of the scene (resources/results/dragon.png -> tests/golden/dragon_reference_silhouette.npz)
    has 21.3 % of its pixels not black.  Here: camera rays through the centres of a 320x180 film, answered by the
    oracle's closest-hit query; the hit mask must cover 21 +- 2 % of the film and overlap that silhouette."""
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "dragon_reference_silhouette.npz"))
    h, w = (int(v) for v in want["shape"])
    silhouette = np.unpackbits(want["mask_bits"])[:h * w].reshape(h, w).astype(bool)
    assert abs(float(want["full_res_fraction"]) - 0.213) < 0.001
    cfg = pkg.workloads.config("dragon", w, h, 1)
    path = str(tmp_path / "dragon.mcsd")
    cfg.save_mcsd(path)
    cam = pkg.mcsd.loads(open(path, "rb").read()).camera
    eye, up = np.array(cam.eye, np.float64), np.array(cam.up, np.float64)
    front = np.array(cam.look_at, np.float64) - eye
    front /= np.linalg.norm(front)
    right = np.cross(front, up)
    right /= np.linalg.norm(right)
    up = np.cross(right, front)
    tan_x = np.tan(np.radians(0.5 * cam.fov_x))                       # camera.cpp:28-40: fov_y = fov_x * height / width
    tan_y = np.tan(np.radians(0.5 * cam.fov_x * h / w))
    hit = np.zeros((h, w), bool)
    with oracle.open(path) as session:
        for j in range(h):
            for i in range(w):
                d = front + (2 * (i + 0.5) / w - 1) * tan_x * right + (1 - 2 * (j + 0.5) / h) * tan_y * up
                hit[j, i] = session.intersect(eye, d / np.linalg.norm(d))[0][0] != 0
    cover = hit.mean()
    print(f"dragon stand-ins: hit mask covers {cover:.4f} of the film, IoU with the reference silhouette {(hit & silhouette).sum() / (hit | silhouette).sum():.3f}")
    assert 0.19 <= cover <= 0.23, cover
    assert (hit & silhouette).sum() / (hit | silhouette).sum() >= 0.75


def test_missing_mesh_without_a_standin_is_the_reference_error(pkg):
    if not os.path.isdir(REF_SCENES):
        pytest.skip("needs the reference's scene files")
    with pytest.raises(pkg.capi.McptError, match="Mesh012.obj"):
        pkg.capi.Config.load_xml(REF_SCENES + XML["dragon"])
    with pytest.raises(pkg.capi.McptError, match="stand-in"):
        pkg.capi.Config.load_xml(REF_SCENES + XML["dragon"], "models/Mesh012.obj torus 1 2 3\n")


# ---- GPU ---------------------------------------------------------------------------------------------
REDUCED = {"cornell": (128, 128, 64), "dragon": (160, 90, 32), "matpreview-rc": (128, 128, 32),
           "matpreview-rd": (128, 128, 32), "volumetric": (160, 90, 64)}


def _draw(pkg, cfg):
    r = pkg.capi.Renderer(cfg, device=0)
    try:
        return r.draw()
    finally:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_baseline_config_against_oracle_reduced_film(pkg, oracle, tmp_path, name):
    from test_gpu_parity import assert_parity
    w, h, spp = REDUCED[name]
    cfg = pkg.workloads.config(name, w, h, spp)
    path = str(tmp_path / "scene.mcsd")
    cfg.save_mcsd(path)
    frame, _ = _draw(pkg, cfg)
    want, _ = oracle.render(path)
    assert_parity(frame, want, name, spp=spp, has_medium=(name == "volumetric"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_baseline_config_full_size_properties(pkg, name):
    """The full film of the configuration: every pixel finite and inside [0, 1] (per-sample clamp,
    renderer.cpp:77-80), the kernel reports W*H*spp samples, a second renderer object gives the same
    frame bit for bit, and the tile sets of two ranks compose to that frame."""
    w, h, spp = pkg.workloads.WORKLOADS[name][1]
    cfg = pkg.workloads.config(name)
    r = pkg.capi.Renderer(cfg, device=0)
    try:
        frame, stats = r.draw(counted=(name == "cornell"))
        assert frame.shape == (h, w, 3) and np.isfinite(frame).all()
        assert frame.min() >= 0.0 and frame.max() <= 1.0
        if name == "cornell":
            assert stats["samples"] == w * h * spp
        digest = hashlib.sha256(frame.tobytes()).hexdigest()
        # two ranks' tile sets, rendered one after the other on this GPU, compose to the same frame
        import torch
        composed = np.zeros_like(frame)
        for rank in range(2):
            rng = pkg.capi.TileRange(rank, 2, 0)
            buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
            r.draw_device(buf.data_ptr(), rng, packed=True)
            pkg.capi.unpack_tiles(buf.cpu().numpy(), rng, w, h, composed)
        assert hashlib.sha256(composed.tobytes()).hexdigest() == digest
    finally:
        r.close()
    # a second renderer object: same frame; with the OTHER kernel formulation too where the build holds it (`make EXPERIMENTAL=1`:
    # a default build would only draw the same kernel again)
    r2 = pkg.capi.Renderer(pkg.workloads.config(name), device=0)
    try:
        r2.set_kernel(0)
        again, _ = r2.draw()
        lanes_kernel = r2.last_kernel()
        assert hashlib.sha256(again.tobytes()).hexdigest() == digest
        if pkg.capi.has_formulations():
            r2.set_kernel(1)
            streamed, st2 = r2.draw()
            assert r2.last_kernel() != lanes_kernel
            assert hashlib.sha256(streamed.tobytes()).hexdigest() == digest
            print(name, "kernels:", lanes_kernel, "|", r2.last_kernel(), "stream kernel ms", st2["kernel_milliseconds"])
    finally:
        r2.close()
    print(name, "full size", (w, h, spp), "kernel ms", stats["kernel_milliseconds"],
          "Msamples/s", w * h * spp / stats["kernel_milliseconds"] / 1e3)


WALK_CHECK = {"cornell": (512, 512, 16), "dragon": (320, 180, 8), "matpreview-rc": (256, 256, 16),
              "matpreview-rd": (256, 256, 16), "volumetric": (640, 360, 16)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_baseline_config_walk_self_check(pkg, name):
    """mcpt_renderer_check_walks on every BASELINE configuration: the production ordered walk and the
    reference-order walk give the same frame bit for bit (full-size runs: tests/full_size_walk_check.py,
    profiles/r02_walk_self_check.json)."""
    r = pkg.capi.Renderer(pkg.workloads.config(name, *WALK_CHECK[name]), device=0)
    try:
        n, first, worst = r.check_walks()
    finally:
        r.close()
    assert n == 0, f"{n} pixels differ, first {first}, max |diff| {worst}"


@pytest.mark.gpu
@pytest.mark.full_parity  # (collected only with MCPT_FULL_PARITY=1 — conftest.py: ~8 minutes of oracle time on 256 host threads)
@pytest.mark.parametrize("name", ["matpreview-rc", "matpreview-rd", "volumetric"])
def test_baseline_config_full_film_equals_the_oracle(pkg, oracle, tmp_path, name):
    """BASELINE configs 4 and 5 at their own films and spp (matpreview 1024 x 1024 spp 512, volumetric-caustic 1280 x 720
    spp 1024), GPU frame == oracle frame, every pixel, bit for bit (cornell and dragon are compared at full film in every run
    of the suite).  Behind MCPT_FULL_PARITY=1 because of the oracle's time; the builder's last run of it:
    profiles/r04_full_size_parity_bit_exact.json."""
    w, h, spp = pkg.workloads.WORKLOADS[name][1]
    cfg = pkg.workloads.config(name)
    path = str(tmp_path / "scene.mcsd")
    cfg.save_mcsd(path)
    frame, stats = _draw(pkg, cfg)
    want, info = oracle.render(path)
    print(name, "full film", (w, h, spp), "GPU kernel ms", stats["kernel_milliseconds"], "oracle seconds", info["seconds"])
    assert frame.shape == want.shape == (h, w, 3)
    assert int((frame != want).any(axis=2).sum()) == 0


FULL_FILM_SMALL_SPP = {"matpreview-rc": 16, "matpreview-rd": 16, "volumetric": 32}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["matpreview-rc", "matpreview-rd", "volumetric"])
def test_baseline_config_full_film_small_spp_equals_the_oracle(pkg, oracle, tmp_path, name):
    """BASELINE configs 4 and 5 at their OWN films — every pixel of 1024 x 1024 / 1280 x 720, so every camera-ray geometry the
    full render meets — at a sample count the oracle finishes in seconds on the GPU box's host threads (matpreview spp 16,
    volumetric-caustic spp 32; a host with few cores: spp 2 / 4).  GPU frame == oracle frame, bit for bit.  The ray query's
    exactness argument is per ray geometry, which is why the whole film at a small spp says more than a crop at
    the full one; the full spp is the test above (MCPT_FULL_PARITY=1)."""
    w, h, _ = pkg.workloads.WORKLOADS[name][1]
    spp = FULL_FILM_SMALL_SPP[name] if (os.cpu_count() or 1) >= 64 else FULL_FILM_SMALL_SPP[name] // 8
    cfg = pkg.workloads.config(name, w, h, spp)
    path = str(tmp_path / "scene.mcsd")
    cfg.save_mcsd(path)
    frame, stats = _draw(pkg, cfg)
    want, info = oracle.render(path)
    print(name, "full film", (w, h, spp), "GPU kernel ms", stats["kernel_milliseconds"], "oracle seconds", info["seconds"])
    assert frame.shape == want.shape == (h, w, 3)
    differing = int((frame != want).any(axis=2).sum())
    assert differing == 0, f"{differing} of {w * h} pixels differ, max |diff| {np.abs(frame - want).max()}"


@pytest.mark.gpu
def test_dragon_full_film_equals_the_oracle(pkg, oracle, tmp_path):
    """north_star's second target at its own film: dragon/scene.xml 1280 x 720 spp 256 (845 808 triangles), GPU frame
    == oracle frame, every pixel, bit for bit.  (About 11 s of oracle time on the GPU box's 256 host threads; a host
    with few cores compares at spp 32 — equality holds at any spp, `u = s / spp` makes the two films different images.)"""
    w, h, spp = pkg.workloads.WORKLOADS["dragon"][1]
    if (os.cpu_count() or 1) < 64:
        spp = 32
    cfg = pkg.workloads.config("dragon", w, h, spp)
    path = str(tmp_path / "dragon.mcsd")
    cfg.save_mcsd(path)
    r = pkg.capi.Renderer(cfg, device=0)
    try:
        frame, stats = r.draw()
        tickets, given, finished = (int(v) for v in r.table("market"))
        kernel = r.last_kernel()
    finally:
        r.close()
    want, info = oracle.render(path)
    print("dragon full film", (w, h, spp), "GPU kernel ms", stats["kernel_milliseconds"], "oracle seconds", info["seconds"], "path market: tickets, records given", tickets, given)
    assert frame.shape == want.shape == (h, w, 3)
    differing = int((frame != want).any(axis=2).sum())
    assert differing == 0, f"{differing} of {w * h} pixels differ, max |diff| {np.abs(frame - want).max()}"
    # (round 6: this frame's last paths travel — through LDS inside a workgroup, through device memory between workgroups, render_kernel_impl.h
    #  "TAIL SPREAD" / "PATH MARKET" — and the equality above is the equality of a frame in which thousands of them did)
    assert "most expensive first" in kernel, kernel
    assert given > 100 and tickets >= given and finished == ((w + 7) // 8) * ((h + 7) // 8) * 64, (tickets, given, finished)


_CONCURRENT_SCRIPT = r"""
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import torch
from _pkg import load_package
pkg = load_package()
w, h, spp = 1280, 720, 8
cfg = pkg.workloads.config("dragon", w, h, spp)
rs = [pkg.capi.Renderer(cfg, device=0) for _ in range(2)]
plain, _ = rs[0].draw()
streams = [torch.cuda.Stream() for _ in rs]
bufs = [torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0") for _ in rs]
for r, s, b in zip(rs, streams, bufs):   # (the first draw of a tile range reads its statistics back: not part of the overlap)
    r.draw_device(b.data_ptr(), None, packed=False, stream=s.cuda_stream, blocking=True)
given = []
for rounds in range(3):
    for b in bufs:
        b.zero_()
    torch.cuda.synchronize()
    for r, s, b in zip(rs, streams, bufs):
        r.draw_device(b.data_ptr(), None, packed=False, stream=s.cuda_stream, blocking=False)
    torch.cuda.synchronize()
    frames = [b.cpu().numpy().reshape(h, w, 3) for b in bufs]
    given.append([int(r.table("market")[1]) for r in rs])
    assert all((f == plain).all() for f in frames), "a frame of two concurrent launches differs from the single launch's"
print(json.dumps({"records_given": given, "kernel": rs[0].last_kernel()}))
"""


@pytest.mark.gpu
def test_two_renderers_at_once_neither_waits_for_the_other(pkg):
    """Two renderers of dragon/scene.xml's class draw AT THE SAME TIME on two streams of one device: each launch is sized for the whole
    GPU, so neither has all its workgroups resident — and a wavefront of the path market waits (holding its slot) only when every
    workgroup of its launch has started (RenderJob::market[96], render_kernel_impl.h).  Both frames equal the single launch's, three
    times; in a subprocess with a time limit (what this guards against is a launch that never ends)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CONCURRENT_SCRIPT, root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print(out)
    assert "pool-walk" in out["kernel"] and len(out["records_given"]) == 3


@pytest.mark.gpu
def test_walk_check_at_renderer_creation(pkg):
    """MCPT_CHECK_WALKS=<spp>: mcpt_renderer_create renders the film with both walks and reports on stderr; the renderer
    works afterwards and its frame is the usual one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, sys.argv[1]); from _pkg import load_package; pkg = load_package(); import hashlib\n"
            "r = pkg.capi.Renderer(pkg.workloads.config('matpreview-rc', 96, 96, 8), device=0)\n"
            "f, _ = r.draw(); print(hashlib.sha256(f.tobytes()).hexdigest())")
    outs = []
    for env in (dict(os.environ, MCPT_CHECK_WALKS="2"), {k: v for k, v in os.environ.items() if k != "MCPT_CHECK_WALKS"}):
        r = subprocess.run([sys.executable, "-c", code, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((r.stdout.strip().splitlines()[-1], r.stderr))
    assert "both walks agree on every pixel at 2 spp" in outs[0][1]
    assert "MCPT_CHECK_WALKS" not in outs[1][1]
    assert outs[0][0] == outs[1][0]


@pytest.mark.formulations
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dragon", "matpreview-rc", "matpreview-rd"])
def test_queued_renderer_on_the_mesh_configurations(pkg, oracle, tmp_path, name):
    """The queued renderer (mode 5) on the BASELINE mesh configurations: == oracle at the reduced film, and at a
    quarter-size film == the stream kernel's frame (hash)."""
    w, h, spp = REDUCED[name]
    cfg = pkg.workloads.config(name, w, h, spp)
    path = str(tmp_path / "scene.mcsd")
    cfg.save_mcsd(path)
    r = pkg.capi.Renderer(cfg, device=0)
    try:
        frame, _ = r.set_kernel(5).draw()
        assert "queued" in r.last_kernel(), r.last_kernel()
    finally:
        r.close()
    want, _ = oracle.render(path)
    differing = int((frame != want).any(axis=2).sum())
    assert differing == 0, f"{differing} pixels differ"
    fw, fh, fspp = pkg.workloads.WORKLOADS[name][1]
    r = pkg.capi.Renderer(pkg.workloads.config(name, fw // 2, fh // 2, max(fspp // 8, 8)), device=0)
    try:
        a, _ = r.set_kernel(5).draw()
        queued_kernel = r.last_kernel()
        b, _ = r.set_kernel(1).draw()
    finally:
        r.close()
    assert "queued" in queued_kernel
    assert hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(b.tobytes()).hexdigest()
