"""CPU-side checks of the product: C ABI surface, host commit tables, and the
kernel body's logic through the host emulator (tests/emu) — all without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from golden_cases import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="module")
def lib(pkg):
    pkg.capi.build()
    return pkg.capi.lib()


@pytest.fixture(scope="module")
def emulator():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu
    return emu.Emulator()


def test_library_exports_every_declared_symbol(pkg, lib):
    header = open(os.path.join(ROOT, "include", "mcpt.h")).read()
    declared = set(re.findall(r"\b(mcpt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(pkg.capi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_config_roundtrip_and_builtin(pkg, lib, tmp_path):
    cfg = pkg.capi.Config.builtin("cornell-box")
    assert cfg.film() == (1024, 1024, 256)
    cfg.set_film(64, 0, 8)                     # 0 keeps the scene's own value
    assert cfg.film() == (64, 1024, 8)
    cfg.set_film(0, 64, 0)
    out = tmp_path / "c.mcsd"
    cfg.save_mcsd(out)
    # the C++ built-in and the Python builder describe the same scene
    assert out.read_bytes() == pkg.mcsd.dumps(pkg.scenes.cornell_box(64, 64, 8))
    again = pkg.capi.Config.load_mcsd(out)
    assert again.film() == (64, 64, 8)


def test_no_cpu_fallback(pkg, lib):
    """Without a HIP device the renderer refuses to exist."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.capi.McptError, match="GPU only|HIP"):
        pkg.capi.Renderer(pkg.capi.Config.builtin("cornell-box").set_film(8, 8, 1))


@pytest.mark.parametrize("name", ["cornell_96_spp32", "rough_dielectric_envmap", "volpath_medium_mixed", "masked_area_flat",
                                  "terrain_directional"])
def test_host_library_renders_the_golden_frames(name, pkg, lib):
    """libmcpt_host.so (include/mcpt_host.h; what `mcpt_cli --cpu` runs): the product's kernel body on host threads,
    fed through mcpt_config_serialize, gives the compiled reference's frames bit for bit."""
    scene = cases(pkg.scenes)[name]
    golden = np.load(os.path.join(GOLDEN, name + ".npz"))["frame"]
    frame, seconds = pkg.capi.host_render(pkg.capi.Config.from_scene(scene))
    assert seconds > 0 and np.array_equal(frame, golden)


def test_tiled_renderer_needs_a_gpu_and_valid_devices(pkg, lib):
    import torch
    cfg = pkg.capi.Config.builtin("cornell-box").set_film(16, 16, 1)
    if not torch.cuda.is_available():
        assert pkg.capi.device_count() == 0
        with pytest.raises(pkg.capi.McptError, match="GPU only|HIP"):
            pkg.capi.TiledRenderer(cfg, devices=(0,))
    else:
        with pytest.raises(pkg.capi.McptError, match="listed twice"):
            pkg.capi.TiledRenderer(cfg, devices=(0, 0))
        with pytest.raises(pkg.capi.McptError, match="invalid HIP device"):
            pkg.capi.TiledRenderer(cfg, devices=(0, 99))


def test_bad_input_is_rejected(pkg, lib, tmp_path):
    with pytest.raises(pkg.capi.McptError):
        pkg.capi.Config.from_mcsd_bytes(b"not a scene")
    with pytest.raises(pkg.capi.McptError):
        pkg.capi.Config.load_mcsd(tmp_path / "missing.mcsd")
    with pytest.raises(pkg.capi.McptError):
        pkg.capi.Config.builtin("teapot")


def test_image_writers(pkg, lib, tmp_path):
    from exr_util import read_exr_zip
    from PIL import Image
    rng = np.random.default_rng(0)
    frame = rng.random((20, 30, 3)).astype(np.float32)
    pkg.capi.write_image(tmp_path / "a.png", frame)
    got = np.asarray(Image.open(tmp_path / "a.png"))
    # reference transfer curve (image_io.cpp:25-53), truncated to 8 bit
    lin = frame.astype(np.float32)
    srgb = np.where(lin <= np.float32(0.0031308), np.float32(12.92) * lin,
                    np.float32(1.055) * lin ** np.float32(1 / 2.4) - np.float32(0.055))
    want = (np.clip(srgb, 0, 1) * 255).astype(np.uint8)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    pkg.capi.write_image(tmp_path / "a.pfm", frame)
    raw = (tmp_path / "a.pfm").read_bytes()
    assert raw.startswith(b"PF\n30 20\n-1.0\n")
    body = np.frombuffer(raw[len(b"PF\n30 20\n-1.0\n"):], dtype="<f4").reshape(20, 30, 3)
    assert np.array_equal(body[::-1], frame)
    # EXR: ZIP-compressed scanline file (blocks of 16 lines), channels B, G, R as FLOAT
    yy, xx = np.mgrid[0:37, 0:50]
    smooth = np.stack([np.sin(xx * 0.1), np.cos(yy * 0.1), (xx + yy) * 0.01], -1).astype(np.float32)
    for name, img in (("a", frame), ("smooth", smooth), ("one", frame[:1, :1].copy())):
        pkg.capi.write_image(tmp_path / f"{name}.exr", img)
        got = read_exr_zip(tmp_path / f"{name}.exr")
        assert np.array_equal(got, img), name
    assert (tmp_path / "smooth.exr").stat().st_size < 0.8 * smooth.nbytes       # it does compress
    pkg.capi.write_image(tmp_path / "smooth.png", np.clip(smooth, 0, 1))
    assert (tmp_path / "smooth.png").stat().st_size < 0.75 * 37 * 50 * 3   # deflated, not stored
    with pytest.raises(pkg.capi.McptError):
        pkg.capi.write_image(tmp_path / "a.bmp", frame)


@pytest.mark.parametrize("n_cus, n_tiles", [(256, 4096), (8, 128), (256, 2400), (4, 7)])
def test_cost_ordered_table_layout(n_cus, n_tiles, pkg, lib):
    """capi.cpp::CostOrderedTable (wavefronts laid out by probed tile cost): every layout is a permutation of the tiles;
    layout 0 is the cost order; the production layout (1) gives every SIMD of a fully resident one-pixel-per-lane launch —
    wavefront g = 4 (cu + k n_cus) + w, k = 0..3, sits on SIMD (cu, w) — one tile of each quarter of the cost order, and the
    SIMDs' sums differ far less than in plain cost order."""
    rng = np.random.default_rng(n_tiles)
    steps = (rng.gamma(4.0, 150.0, n_tiles)).astype(np.uint32) + 1
    order = np.argsort(-steps.astype(np.int64), kind="stable")
    tables = {layout: pkg.capi.debug_cost_table(steps, n_cus, layout) for layout in (0, 1, 3)}
    for layout, t in tables.items():
        assert sorted(t.tolist()) == list(range(n_tiles)), layout
    np.testing.assert_array_equal(tables[0], order)
    if n_tiles == 16 * n_cus:
        g = np.arange(n_tiles)
        simd = ((g // 4) % n_cus) * 4 + (g % 4)
        rank = np.empty(n_tiles, np.int64)
        rank[order] = np.arange(n_tiles)
        quarter = rank[tables[1]] // (n_tiles // 4)
        for s_id in range(0, 4 * n_cus, max(1, n_cus // 4)):
            assert sorted(quarter[simd == s_id].tolist()) == [0, 1, 2, 3]
        spread = {layout: np.ptp(np.bincount(simd, weights=steps[t].astype(np.float64))) for layout, t in tables.items()}
        assert spread[3] < spread[1] < spread[0], spread


def test_tile_unpack(pkg, lib):
    w, h = 21, 13
    frame = np.zeros((h, w, 3), dtype=np.float32)
    want = np.zeros((h, w, 3), dtype=np.float32)
    tiles_x, tiles_y = 3, 2
    for rank in range(4):
        rng = pkg.capi.TileRange(rank, 4, 0)
        n = lib.mcpt_tile_range_size(tiles_x * tiles_y, ctypes.byref(rng))
        assert n == len(range(rank, tiles_x * tiles_y, 4))
        packed = np.zeros((n, 64, 3), dtype=np.float32)
        for k in range(n):
            t = rank + 4 * k
            for r in range(64):
                x, y = (t % tiles_x) * 8 + r % 8, (t // tiles_x) * 8 + r // 8
                packed[k, r] = (t, r, 1)
                if x < w and y < h:
                    want[y, x] = (t, r, 1)
        pkg.capi.unpack_tiles(packed, rng, w, h, frame)
    assert np.array_equal(frame, want) and (frame[..., 2] == 1).all()


# ---- host commit vs oracle: same tree, expressed with skip links ------------
def _skip_links_from_children(links):
    """Oracle nodes (leaf, left, right, object; tree-local) of ONE tree ->
    expected (skip, object) per node in the same (pre-order) numbering."""
    n = len(links)
    skip = np.full(n, 0xFFFFFFFF, dtype=np.uint64)
    for i in range(n):
        leaf, left, right, _ = links[i]
        if not leaf:
            assert left == i + 1
            skip[left] = right
            skip[right] = skip[i]
    return skip


@pytest.mark.parametrize("name", ["cornell_64_spp8", "volumetric_96x54_spp16", "terrain_directional",
                                  "rough_plastic_constant_cyl"])
def test_commit_matches_oracle_tables(name, pkg, oracle, emulator, mcsd_file):
    scene = cases(pkg.scenes)[name]
    path = mcsd_file(scene)
    o_links, o_geom = oracle.nodes(path)
    e_links, e_geom = emulator.nodes(path)
    assert len(o_links) == len(e_links)
    assert np.array_equal(o_geom, e_geom)            # areas and boxes, node for node
    # split the oracle's array into trees: TLAS has 2*inst-1 nodes, then each BLAS
    n_inst = len(scene.instances)
    n_tlas = 2 * n_inst - 1
    tlas_skip = _skip_links_from_children(o_links[:n_tlas])
    assert np.array_equal(e_links[:n_tlas, 0].astype(np.uint64), tlas_skip)
    # TLAS leaves name instances; BLAS leaves name GLOBAL primitive slots
    leaves = o_links[:n_tlas, 0] == 1
    assert np.array_equal(e_links[:n_tlas, 1][leaves], o_links[:n_tlas, 3][leaves])
    assert (e_links[:n_tlas, 1][~leaves] == 0xFFFFFFFF).all()
    # BLAS trees follow in instance order; a tree ends where a root's subtree ends
    base, prim_base = n_tlas, 0
    for _ in range(n_inst):
        # size of this tree: walk the oracle's child links from its root
        size, stack = 0, [0]
        tree = o_links[base:]
        while stack:
            i = stack.pop()
            size += 1
            if not tree[i][0]:
                stack += [tree[i][1], tree[i][2]]
        tree = o_links[base:base + size]
        want_skip = _skip_links_from_children(tree)
        got_skip = e_links[base:base + size, 0].astype(np.uint64)
        end = got_skip == 0xFFFFFFFF
        assert np.array_equal(end, want_skip == 0xFFFFFFFF)
        assert np.array_equal(got_skip[~end] - base, want_skip[~end])
        lf = tree[:, 0] == 1
        assert np.array_equal(e_links[base:base + size, 1][lf].astype(np.int64) - prim_base,
                              tree[:, 3][lf].astype(np.int64))
        base += size
        prim_base += int(lf.sum())
    assert base == len(o_links)


def _lbvh_inputs(n, seed, kind):
    rng = np.random.default_rng(seed)
    lo = (rng.random((n, 3)) * 4 - 2).astype(np.float32)
    hi = lo + (rng.random((n, 3)) * 0.5).astype(np.float32)
    if kind == "planar":
        hi[:, 2] = lo[:, 2] = 1.0                       # zero extent -> NaN Morton input
    elif kind == "duplicates":
        lo[n // 2:] = lo[:n - n // 2]                   # equal Morton codes: the index breaks ties
        hi[n // 2:] = hi[:n - n // 2]
    elif kind == "clustered":
        lo = (lo * 0.001).astype(np.float32)            # many primitives per Morton cell
        hi = (lo + 1e-4).astype(np.float32)
    return np.concatenate([lo, hi], 1), rng.random(n).astype(np.float32)


def _check_against_oracle_tree(links, geom, tree):
    """(skip, object) + (area, box) of the product's builder against the oracle's
    child-link tree (same pre-order numbering)."""
    n = len(tree["leaf"])
    assert len(links) == n
    o_links = np.stack([tree["leaf"], tree["left"], tree["right"], tree["object"]], 1)
    want_skip = _skip_links_from_children(o_links)
    assert np.array_equal(links[:, 0].astype(np.uint64), want_skip)
    leaf = tree["leaf"] == 1
    assert np.array_equal(links[:, 1][leaf], tree["object"][leaf]) and (links[:, 1][~leaf] == 0xFFFFFFFF).all()
    assert np.array_equal(geom[:, 0], tree["area"]) and np.array_equal(geom[:, 1:], tree["box"])


@pytest.mark.parametrize("n, kind", [(1, "random"), (2, "random"), (3, "random"), (37, "planar"), (500, "random"),
                                     (1000, "duplicates"), (4000, "clustered")])
def test_host_lbvh_builder_matches_oracle(n, kind, pkg, lib, oracle):
    """mcpt_debug_lbvh_build with the host builder (no GPU needed) == the oracle's
    restatement of bvh_builder.cpp, which is pinned to the compiled reference."""
    boxes, areas = _lbvh_inputs(n, 100 + n, kind)
    links, geom, _ = pkg.capi.lbvh_build(boxes, areas, on_device=False)
    _check_against_oracle_tree(links, geom, oracle.bvh_build(boxes, areas))


# ---- the ordered-walk hierarchy ------------------------------------------------
@pytest.mark.parametrize("name", ["cornell_64_spp8", "rough_conductor_envmap", "conductor_aniso_mixed",
                                  "rough_diffuse_point_disk", "terrain_directional"])
def test_walk_hierarchy_is_sound(name, pkg, emulator, mcsd_file):
    """Every primitive is exactly one leaf; triangle leaf boxes are the bounds of
    the vertices (the reference's leaf boxes, triangle.cpp:9-15); interior boxes
    are the exact unions of their children; the depth that sizes the traversal
    stack is the real depth and within the bound; ranks enumerate the reference's
    visiting order (TLAS pre-order, then each instance's BLAS pre-order)."""
    scene = cases(pkg.scenes)[name]
    path = mcsd_file(scene)
    nodes, prims, info = emulator.walk(path)
    links, _ = emulator.nodes(path)
    refs = nodes[:, :2, 3].copy().view(np.uint32)          # (n, 2): child references
    slot_prim = prims[:, 0, 3].copy().view(np.uint32)
    slot_inst = prims[:, 1, 3].copy().view(np.uint32)
    slot_rank = prims[:, 2, 3].copy().view(np.uint32)
    n_prims = len(prims)
    assert sorted(slot_prim) == list(range(n_prims)) and sorted(slot_rank) == list(range(n_prims))
    triangle_instances = {i for i, inst in enumerate(scene.instances) if inst.type in
                          (pkg.mcsd.INST_MESHES, pkg.mcsd.INST_CUBE, pkg.mcsd.INST_RECTANGLE)}
    LEAF = 0x80000000
    seen, depths = [], []

    def visit(node, depth):
        depths.append(depth + 1)
        boxes = []
        for c in range(2):
            lo, hi = nodes[node, 2 * c, :3], nodes[node, 2 * c + 1, :3]
            child = int(refs[node, c])
            if node == 0 and c == 1:
                assert (lo > hi).all()                       # the empty child of the top node
                continue
            if child & LEAF:
                slot = child & ~LEAF
                seen.append(slot)
                if int(slot_inst[slot]) in triangle_instances:
                    tri = prims[slot][:, :3]
                    np.testing.assert_array_equal(lo, tri.min(0))
                    np.testing.assert_array_equal(hi, tri.max(0))
            else:
                sub_lo, sub_hi = visit(child, depth + 1)
                np.testing.assert_array_equal(lo, sub_lo)
                np.testing.assert_array_equal(hi, sub_hi)
            boxes.append((lo, hi))
        return np.min([b[0] for b in boxes], axis=0), np.max([b[1] for b in boxes], axis=0)

    visit(0, 0)
    assert sorted(seen) == list(range(n_prims))
    assert max(depths) + 1 == info["depth"] <= 57        # stack entries = tree depth + the sentinel
    # a tree's nodes are contiguous and only its last node is a leaf whose skip link is "end"
    tree_end = [k + 1 for k in range(len(links)) if links[k, 0] == 0xFFFFFFFF and links[k, 1] != 0xFFFFFFFF]
    n_tlas = tree_end[0]
    inst_order = [int(obj) for _, obj in links[:n_tlas] if obj != 0xFFFFFFFF]
    per_inst = [[int(obj) for _, obj in links[a:b] if obj != 0xFFFFFFFF] for a, b in zip(tree_end[:-1], tree_end[1:])]
    assert len(per_inst) == len(scene.instances)
    want = [p for inst in inst_order for p in per_inst[inst]]
    assert [int(p) for p in slot_prim[np.argsort(slot_rank)]] == want


# ---- kernel-body logic on CPU (bit-exact against reference goldens) ---------
@pytest.mark.parametrize("walk", ["ordered", "reference"])
@pytest.mark.parametrize("name", sorted(MANIFEST["frames"]))
def test_kernel_body_emulated_matches_golden(name, walk, pkg, emulator, mcsd_file):
    """Both ray queries — the production ordered walk of the SAH hierarchy and the
    reference-order walk — reproduce the compiled reference's frames bit for bit
    (scenes with opacity masks always take the reference-order walk)."""
    scene = cases(pkg.scenes)[name]
    golden = np.load(os.path.join(GOLDEN, name + ".npz"))["frame"]
    frame, _ = emulator.render(mcsd_file(scene), scene.camera.width, scene.camera.height,
                               variant=-1 if walk == "ordered" else emulator.REFERENCE)
    assert np.array_equal(frame, golden), f"max diff {np.abs(frame - golden).max():.3e}"


@pytest.mark.parametrize("slots", [64, 192])
@pytest.mark.parametrize("name", sorted(MANIFEST["frames"]))
def test_stream_formulation_emulated_matches_golden(name, slots, pkg, emulator, mcsd_file):
    """csrc/stream_core.h on the host — workgroups of `slots` path slots, rounds of shade (deferred shadow
    answers, ray emission) and trace (the lane loop with retire / fetch) — reproduces the compiled reference's
    frames bit for bit: the pooled formulation consumes each pixel's random stream in the reference's order."""
    scene = cases(pkg.scenes)[name]
    golden = np.load(os.path.join(GOLDEN, name + ".npz"))["frame"]
    try:
        frame, _ = emulator.render_stream(mcsd_file(scene), scene.camera.width, scene.camera.height, slots)
    except RuntimeError as e:
        assert "not a scene for the stream kernel" in str(e)
        pytest.skip(str(e))
    assert np.array_equal(frame, golden), f"max diff {np.abs(frame - golden).max():.3e}"


@pytest.mark.parametrize("name", sorted(MANIFEST["frames"]))
def test_wide_quantised_hierarchy_emulated_matches_golden(name, pkg, emulator, mcsd_file):
    """The 4-wide quantised form of the ordered-walk hierarchy (device_scene.h: wide_nodes — four children per 64-byte
    node, child boxes as 8-bit offsets on the node's grid, decoded boxes containing the exact ones) walked with the short
    stack (short_stack.h): every golden frame of the compiled reference, bit for bit.  Larger boxes only ever add visits;
    what they let through is stopped by the exact leaf-box test at the primitive (test_slot, kLeafCheck)."""
    scene = cases(pkg.scenes)[name]
    golden = np.load(os.path.join(GOLDEN, name + ".npz"))["frame"]
    try:
        frame, _ = emulator.render(mcsd_file(scene), scene.camera.width, scene.camera.height,
                                   variant=1 | 2 | 4 | 8 | 16 | emulator.ORDERED | emulator.WIDE)
    except RuntimeError as e:
        assert "opacity masks" in str(e)
        pytest.skip(str(e))
    assert np.array_equal(frame, golden), f"max diff {np.abs(frame - golden).max():.3e}"


def test_wide_quantised_hierarchy_answers_like_the_reference_walk(pkg, emulator, tmp_path):
    """Closest-hit queries on a mesh: rays from everywhere, rays that graze, rays along mesh edges and through vertices
    (where hits tie), rays that start on the surface — primitive and distance equal the reference-order walk's."""
    scene = pkg.scenes.terrain_scene(96, 64, 48, 1)
    path = tmp_path / "terrain.mcsd"
    pkg.mcsd.dump(scene, path)
    rng = np.random.default_rng(3)
    n = 200000
    o = np.stack([rng.uniform(-3, 3, n), rng.uniform(0.0, 3, n), rng.uniform(-3, 3, n)], 1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # towards points ON mesh edges (shared by two triangles: the hits tie or nearly tie).  (Rays through a mesh VERTEX —
    # six coincident hits — are beyond the pairwise replay of test_slot in either hierarchy: DESIGN.md section 2.)
    mesh = scene.instances[0]
    P = np.asarray(mesh.positions, np.float32).reshape(-1, 3)
    I = np.asarray(mesh.indices, np.int64).reshape(-1, 3)
    k = 50000
    tri = I[rng.integers(0, len(I), k)]
    a, b = P[tri[:, 0]], P[tri[:, 1]]
    u = rng.uniform(0.05, 0.95, (k, 1)).astype(np.float32)
    tgt = a * (1 - u) + b * u
    o2 = tgt + np.stack([rng.normal(0, 1, k), rng.uniform(0.5, 3, k), rng.normal(0, 1, k)], 1)
    d2 = tgt - o2
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    rays = np.concatenate([np.concatenate([o, d], 1), np.concatenate([o2, d2], 1)]).astype(np.float32)
    w_prim, w_t = emulator.closest(path, rays, ordered=2)
    r_prim, r_t = emulator.closest(path, rays, ordered=0)
    assert 0.2 < (r_prim >= 0).mean() < 1.0
    bad = (w_prim != r_prim) | (w_t != r_t)
    assert not bad.any(), (int(bad.sum()), rays[bad][:3], w_prim[bad][:3], r_prim[bad][:3])


@pytest.mark.parametrize("slots", [0, 700])
@pytest.mark.parametrize("name", sorted(MANIFEST["frames"]))
def test_queued_formulation_emulated_matches_golden(name, slots, pkg, emulator, mcsd_file):
    """csrc/queue_core.h on the host — a pool of path slots (one per pixel, or 700 that take pixel after pixel), rounds
    of trace (the extension ray's answer goes to the shade queue of the material group it hit; an unoccluded shadow ray
    adds the contribution it carries to its slot's radiance) and one shade pass per material group, each compiled with
    that group's BSDF model only — reproduces the compiled reference's frames bit for bit."""
    scene = cases(pkg.scenes)[name]
    golden = np.load(os.path.join(GOLDEN, name + ".npz"))["frame"]
    try:
        frame, rounds = emulator.render_queued(mcsd_file(scene), scene.camera.width, scene.camera.height, slots)
    except RuntimeError as e:
        assert "not a scene for the queued renderer" in str(e)
        pytest.skip(str(e))
    assert rounds > scene.camera.spp
    assert np.array_equal(frame, golden), f"max diff {np.abs(frame - golden).max():.3e}"


@pytest.mark.parametrize("strategy", [1, 2, 3])
@pytest.mark.parametrize("name", ["cornell_64_spp8", "thin_dielectric_sun", "conductor_aniso_mixed",
                                  "volumetric_iso_64x36_spp8"])
def test_image_does_not_depend_on_the_walk_tree(name, strategy, pkg, emulator, mcsd_file):
    """The ordered walk's answer must not depend on how its hierarchy was built: with an
    exact-sweep SAH tree (1), a median-split tree (2) and the production tree with every
    node's children swapped (3) the frames are still the compiled reference's, bit for
    bit.  (Before hits within rounding distance of each other were decided by replaying
    the reference on the pair — traversal.h, test_slot — tree 1 changed 4 pixels of the
    cornell frame, where rays with dx == dy hit the floor / wall seam, and 18 pixels of
    thin_dielectric_sun.)"""
    scene = cases(pkg.scenes)[name]
    golden = np.load(os.path.join(GOLDEN, name + ".npz"))["frame"]
    emulator.set_walk_tree(strategy)
    try:
        frame, _ = emulator.render(mcsd_file(scene), scene.camera.width, scene.camera.height)
    finally:
        emulator.set_walk_tree(0)
    assert np.array_equal(frame, golden), f"max diff {np.abs(frame - golden).max():.3e}"


def test_kernel_variants_agree(pkg, emulator, mcsd_file):
    """The specialised kernel instantiations compute the same image as the
    general one, with either walk."""
    scene = pkg.scenes.cornell_box(48, 48, 4)
    path = mcsd_file(scene)
    lean, info = emulator.render(path, 48, 48)
    assert info["features"] == 0
    for variant in (0, 2, 26, 31, 32, 34, 58, 63):
        other, _ = emulator.render(path, 48, 48, variant=variant)
        assert np.array_equal(lean, other), variant


def test_masks_force_the_reference_walk(pkg, emulator, mcsd_file):
    scene = cases(pkg.scenes)["masked_area_flat"]
    path = mcsd_file(scene)
    assert emulator.walk(path)[2]["has_masks"]
    with pytest.raises(RuntimeError, match="opacity masks"):
        emulator.render(path, 48, 48, variant=31 | emulator.ORDERED)


def test_emulated_counters_match_oracle(pkg, oracle, emulator, mcsd_file):
    """Reference-order walk: the same rays AND the same node / primitive tests as the
    oracle.  Ordered walk: the same rays (the image is the same), its own test counts."""
    scene = pkg.scenes.cornell_box(48, 48, 4)
    path = mcsd_file(scene)
    _, want = oracle.render(path, with_stats=True)
    _, got = emulator.render(path, 48, 48, variant=emulator.REFERENCE, counted=True)
    for key in ("closest_rays", "shadow_rays", "node_tests", "prim_tests"):
        assert got[key] == want[key], key
    assert got["samples"] == 48 * 48 * 4
    _, ordered = emulator.render(path, 48, 48, counted=True)
    for key in ("closest_rays", "shadow_rays", "shaded_hits", "samples"):
        assert ordered[key] == got[key], key
    assert 0 < ordered["prim_tests"] and ordered["node_tests"] % 2 == 0


def test_slivers_are_decided_like_the_reference(pkg, emulator, tmp_path):
    """A window-frame strip as in the reference's classroom scene: two 40-unit-long, 0.04-wide triangles
    meeting a vertical face along an edge, at coordinates of 7 .. 20.  The reference's triangle distance
    for such a sliver is off by up to ~1e-3 (weighted mean of vertex depths with ill-conditioned
    weights), so which of two surfaces it reports near the edge depends on its visiting order and on
    whether the sliver's flat leaf box still passes — and a ray that starts on a sliver may or may not
    "hit" it again just beyond t_min.  500 000 closest queries, ordered walk == reference-order walk in
    primitive and distance.  (Without the sliver handling of commit.cpp / traversal.h::test_slot:
    10 mismatches.)"""
    S, M = pkg.scenes, pkg.mcsd
    s = S.cornell_box(8, 8, 1)
    s.instances = s.instances[-1:]
    P = np.array([[6.87156, 1.0016, 20.268], [6.87156, 1.0016, -20.294], [6.9114, 1.0016, -20.294], [6.9114, 1.0016, 20.268],
                  [6.87156, 1.13605, 20.268], [6.87156, 1.13605, -20.294]], np.float32)
    s.instances.append(M.Instance(type=M.INST_MESHES, id_bsdf=0, positions=P,
                                  indices=np.array([[0, 1, 2], [0, 2, 3], [4, 1, 0], [4, 5, 1]], np.uint32)))
    Q = np.array([[-30, 0, -30], [30, 0, -30], [30, 0, 30], [-30, 0, 30], [6.95, -5, -30], [6.95, 5, -30], [6.95, 5, 30],
                  [6.95, -5, 30]], np.float32)
    s.instances.append(M.Instance(type=M.INST_MESHES, id_bsdf=1, positions=Q,
                                  indices=np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]], np.uint32)))
    path = tmp_path / "sliver.mcsd"
    M.dump(s, path)
    _, prims, _ = emulator.walk(path)
    flags = prims[:, 2, 3].copy().view(np.uint32) >> 31
    slot_prim = prims[:, 0, 3].copy().view(np.uint32)
    # primitives 0, 1 = the light; 2..5 = the strip (40 x 0.04) and its face (40 x 0.13): aspect ratios 1000 and 300
    assert sorted(slot_prim[flags == 1]) == [2, 3, 4, 5]
    rng = np.random.default_rng(1)
    n = 400000
    target = np.stack([6.87156 + rng.normal(0, 0.01, n), 1.0016 + rng.normal(0, 0.01, n), rng.uniform(-20, 20, n)], 1)
    origin = target + np.stack([-rng.uniform(1, 6, n), rng.uniform(0.2, 4, n), rng.normal(0, 4, n)], 1)
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    m = 100000                                                     # rays that start ON the slivers
    o2 = np.stack([rng.uniform(6.87156, 6.9114, m), np.full(m, 1.0016), rng.uniform(-20, 20, m)], 1)
    d2 = rng.normal(size=(m, 3))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    rays = np.concatenate([np.concatenate([origin, d], 1), np.concatenate([o2, d2], 1)]).astype(np.float32)
    a_prim, a_t = emulator.closest(path, rays, ordered=True)
    b_prim, b_t = emulator.closest(path, rays, ordered=False)
    assert (b_prim >= 0).mean() > 0.9
    bad = (a_prim != b_prim) | (a_t != b_t)
    assert not bad.any(), (int(bad.sum()), rays[bad][:3], a_prim[bad][:3], b_prim[bad][:3])
    # ... and the 4-wide quantised hierarchy (larger boxes, explicit leaf-box test at the primitive)
    c_prim, c_t = emulator.closest(path, rays, ordered=2)
    bad = (c_prim != b_prim) | (c_t != b_t)
    assert not bad.any(), (int(bad.sum()), rays[bad][:3], c_prim[bad][:3], b_prim[bad][:3])


@pytest.mark.parametrize("which", ["cornell", "terrain", "preview-sphere"])
def test_pool_hierarchy_is_the_binary_one_collapsed(pkg, emulator, tmp_path, which):
    """DeviceScene::pool_nodes (commit.cpp, BuildPoolNodes: the hierarchy of the wavefront-cooperative pool walk): four children per
    node, each child's box bit-identical to a child box of the binary ordered-walk hierarchy whose subtree it stands for, every
    primitive slot referenced exactly once, unused children with a box no ray enters, depth about half the binary one."""
    S = pkg.scenes
    scene = {"cornell": lambda: S.cornell_box(8, 8, 1), "terrain": lambda: S.terrain_scene(24, 8, 8, 1),
             "preview-sphere": lambda: S.material_preview("diffuse", "area", "sphere", 8, 8, 1)}[which]()
    path = tmp_path / "scene.mcsd"
    pkg.mcsd.dump(scene, path)
    bnodes, prims, info = emulator.walk(path)
    planes, refs, depth = emulator.pool_nodes(path)
    LEAF = np.uint32(0x80000000)
    brefs = bnodes[:, :2, 3].copy().view(np.uint32)          # child references of the binary nodes
    bbox = np.stack([bnodes[:, 0, :3], bnodes[:, 1, :3], bnodes[:, 2, :3], bnodes[:, 3, :3]], 1)   # lo0 hi0 lo1 hi1

    def binary_boxes_under(node):
        """(reference -> (lo, hi)) of every node / leaf below binary node `node`, itself excluded"""
        out, todo = {}, [node]
        while todo:
            n = todo.pop()
            for c in range(2):
                lo, hi = bbox[n, 2 * c], bbox[n, 2 * c + 1]
                if lo[0] > hi[0]:
                    continue
                ref = int(brefs[n, c])
                out[ref] = (lo, hi)
                if not ref & int(LEAF):
                    todo.append(ref)
        return out

    below_root = binary_boxes_under(0)
    seen_slots, n_children = [], 0
    for k in range(len(planes)):
        for c in range(4):
            lo, hi, ref = planes[k, :3, c], planes[k, 3:, c], int(refs[k, c])
            if lo[0] > hi[0]:
                assert (lo == np.float32(3.402823466e+38)).all() and (hi == -np.float32(3.402823466e+38)).all() and ref == int(LEAF)
                continue
            n_children += 1
            if ref & int(LEAF):
                seen_slots.append(ref & 0x7FFFFFFF)
                want = below_root[ref]
                assert np.array_equal(lo, want[0]) and np.array_equal(hi, want[1])
            else:
                assert 0 < ref < len(planes)
                # an inner child's box is a box of the binary hierarchy too
                assert any(np.array_equal(lo, b[0]) and np.array_equal(hi, b[1]) for r, b in below_root.items() if not r & int(LEAF))
    assert sorted(seen_slots) == list(range(len(prims)))
    assert 1 <= depth <= info["depth"] and (len(prims) < 8 or 2 * depth <= info["depth"] + 3)
    assert len(planes) <= len(bnodes)
