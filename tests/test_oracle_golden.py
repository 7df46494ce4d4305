"""Pins the oracle (oracle/prt_oracle.c) to the golden vectors produced by the COMPILED REFERENCE
(tests/golden/make_golden.py).  Bit-exact everywhere: these are the same IEEE binary32 operations
in the same order."""
import ctypes as C
import os

import numpy as np
import pytest

import prt_testlib as T

G = T.GOLDEN


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a, b = bits(a), bits(b)
    bad = np.nonzero(a != b)
    assert len(bad[0]) == 0, f"{what}: {len(bad[0])} mismatches, first at {tuple(x[0] for x in bad)}"


def test_reference_known_answer_triangle(oracle_lib):
    # the reference's only hot-path assertion: tests/tests.cpp:109-127, t == 1.0f exactly
    L = oracle_lib
    f3 = T.f3
    ijk = (C.c_float * 3)()
    t = L.orc_intersect_triangle(f3((0.4, 0.4, -1)), f3((0, 0, 1)), 0, 0, f3((0, 0, 0)), f3((1, 0, 0)), f3((0, 1, 0)), ijk)
    assert t == 1.0


def test_leaf_vectors(oracle_lib):
    L = oracle_lib
    z = np.load(os.path.join(G, "leaf_vectors.npz"))
    inp, ref = z["inputs"], z["outputs"]
    n = len(inp)
    out = np.zeros((n, 24), dtype=np.float32)
    inv = (C.c_float * 3)()
    sx, sy = C.c_int(), C.c_int()
    ijk = (C.c_float * 3)()
    for r in range(n):
        p = inp[r]
        org, d, p0, p1, p2, lo, hi = (T.fp(np.ascontiguousarray(p[a:a + 3])) for a in (0, 3, 6, 9, 12, 15, 18))
        maxT = float(p[21])
        L.orc_ray_prepare_soa(d, inv, C.byref(sx), C.byref(sy))
        out[r, 16:19] = inv[:]
        out[r, 19], out[r, 20] = sx.value, sy.value
        t = L.orc_intersect_triangle(org, d, sx.value, sy.value, p0, p1, p2, ijk)
        out[r, 0] = t
        if t != -1.0:
            out[r, 1:4] = ijk[:]
        out[r, 14] = L.orc_bbox_intersect_soa(lo, hi, org, inv, maxT)
        L.orc_ray_prepare_single(d, inv, C.byref(sx), C.byref(sy))
        out[r, 21], out[r, 22] = sx.value, sy.value
        t = L.orc_intersect_triangle(org, d, sx.value, sy.value, p0, p1, p2, ijk)
        out[r, 4] = t
        if t != -1.0:
            out[r, 5:8] = ijk[:]
        t = L.orc_intersect_triangle_scalar(org, d, p0, p1, p2, ijk)
        out[r, 8] = t
        if t != -1.0:
            out[r, 9:12] = ijk[:]
        out[r, 12] = L.orc_bbox_intersect_t(lo, hi, org, inv)
        out[r, 13] = L.orc_bbox_intersect_bool(lo, hi, org, inv, maxT)
    cols = [c for c in range(23) if c != 15]  # col 15 (SoA box -> t) is not on the path (bvh.cpp never calls it)
    # NaN payloads may differ between x86 and the restatement; compare NaN-ness, bits elsewhere
    a, b = out[:, cols], ref[:, cols]
    nan = np.isnan(a) & np.isnan(b)
    assert_bits_equal(np.where(nan, 0, a), np.where(nan, 0, b), "leaf vectors")


def test_bvh_build_matches_reference():
    z = np.load(os.path.join(G, "bvh_cornell_teapot.npz"))
    desc = T.cornell_scene(512, 512)
    assert_bits_equal(desc.meshes[1].normals, z["teapot_normals"], "calculateVertexNormals")
    s = T.OracleScene(desc)
    for i in range(2):
        ref_nodes = z[f"nodes{i}"].view(T.NODE_DTYPE).reshape(-1)
        nodes = s.nodes(i)
        assert len(nodes) == len(ref_nodes)
        for f in ("primOrSecondNodeIndex", "primCount", "splitAxis"):
            assert (nodes[f] == ref_nodes[f]).all(), f
        leaf = nodes["primCount"] != 0xF
        assert (nodes["triVectorIndex"][leaf] == ref_nodes["triVectorIndex"][leaf]).all()
        assert_bits_equal(nodes["lower"], ref_nodes["lower"], "lower")
        assert_bits_equal(nodes["upper"], ref_nodes["upper"], "upper")
        assert (s.prim_remap(i) == z[f"remap{i}"]).all()
        assert int(leaf.sum()) == int(z["leaf_counts"][i])
    assert_bits_equal(np.ctypeslib.as_array(s.L.orc_scene_bbox(s.scene), shape=(6,)), z["scene_bbox"], "scene bbox")
    assert bits(np.float32(s.radius())) == bits(z["radius"])
    assert len(s.nodes(0)) == 11 and len(s.nodes(1)) == 5255  # SURVEY.md 3.2 probe values


def test_rays_match_reference():
    z = np.load(os.path.join(G, "rays_cornell_teapot.npz"))
    s = T.OracleScene(T.cornell_scene(512, 512))
    far = float(z["max_t"])
    single, occ1 = s.intersect_single(z["org"], z["dir"], far)
    packet, occ8 = s.intersect_packet(z["org"], z["dir"], far)
    rs = z["single"].view(T.HIT_DTYPE).reshape(-1)
    rp = z["packet"].view(T.HIT_DTYPE).reshape(-1)
    for got, ref, what in ((single, rs, "single"), (packet, rp, "packet")):
        assert_bits_equal(got["t"], ref["t"], what + ".t")
        hit = ref["t"] != -1
        for f in ("i", "j", "k"):
            assert_bits_equal(got[f][hit], ref[f][hit], what + "." + f)
        assert (got["primId"][hit] == ref["primId"][hit]).all() and (got["meshId"][hit] == ref["meshId"][hit]).all()
    assert (occ1 == z["occluded_single"]).all()
    assert (occ8 == z["occluded_packet"]).all()


def test_camera_and_rng_match_reference(oracle_lib):
    L = oracle_lib
    z = np.load(os.path.join(G, "camera_packets.npz"))
    s = T.OracleScene(T.cornell_scene(512, 512, with_teapot=False))
    for (x, y, state), ref in zip(z["xys"], z["out"]):
        rng = C.c_uint32(int(state))
        org = np.zeros((8, 3), dtype=np.float32)
        d = np.zeros((8, 3), dtype=np.float32)
        avg = np.zeros(3, dtype=np.float32)
        L.orc_camera_packet(C.byref(s.camera), C.byref(rng), int(x), int(y), T.vptr(org), T.vptr(d), T.fp(avg))
        rec = ref[:88].reshape(8, 11)
        assert_bits_equal(org, rec[:, 0:3], "org")
        assert_bits_equal(d, rec[:, 3:6], "dir")
        inv = (C.c_float * 3)()
        sx, sy = C.c_int(), C.c_int()
        for l in range(8):
            L.orc_ray_prepare_soa(T.fp(d[l]), inv, C.byref(sx), C.byref(sy))
            assert_bits_equal(np.array(inv[:], dtype=np.float32), rec[l, 6:9], "invDir")
            assert (sx.value, sy.value) == (int(rec[l, 9]), int(rec[l, 10]))
        assert_bits_equal(avg, ref[88:91], "avgDir")
        assert rng.value == int(ref[91:92].view(np.uint32)[0])
        g = [L.orc_rng_float(C.byref(rng)) for _ in range(2)]
        g += [np.float32(2.0) * np.float32(L.orc_rng_float(C.byref(rng))) - np.float32(1.0) for _ in range(2)]
        assert_bits_equal(np.array(g, dtype=np.float32), ref[92:96], "rng floats")
        cam = np.array(list(s.camera.pos) + list(s.camera.dir) + list(s.camera.up) + list(s.camera.right), dtype=np.float32)
        assert_bits_equal(cam, ref[96:108], "camera basis")


def test_radiance_matches_reference():
    z = np.load(os.path.join(G, "radiance_c1_crop.npz"))
    x0, y0, x1, y1 = (int(v) for v in z["rect"])
    s = T.OracleScene(T.cornell_scene(512, 512))
    rgb, st = s.trace_block(x0, y0, x1, y1, 16)
    assert_bits_equal(rgb, z["rgb"], "C1 crop radiance")
    assert st["raysTraced"] == int(z["rays"][0]) and st["occludedTraced"] == int(z["rays"][1])
    s2 = T.OracleScene(T.cornell_scene(128, 128, with_teapot=False))
    rgb2, st2 = s2.render(16)
    assert_bits_equal(rgb2, z["cornell_only_rgb"], "cornell-only radiance")
    assert st2["raysTraced"] == int(z["cornell_only_rays"][0]) == 1126145
    m = rgb2.reshape(-1, 3).astype(np.float64).mean(0)
    assert np.allclose(m, [0.198639526, 0.130990393, 0.039316328], rtol=0, atol=5e-9)


def test_any_hit_answers_do_not_depend_on_the_visiting_order(oracle_lib):
    """The GPU path visits the nearer child first in occlusion queries (the reference: child 0 first).  With the GPU's
    accounting on, the oracle walks every occlusion ray both ways and abort()s on a differing answer; the image must be
    the same bits and only the occlusion share of nBox/nTri may change."""
    import prt_amd
    scene, camera, exposure = prt_amd.setup_atrium_standin(96, 54, tris=20000)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    s = T.OracleScene(desc)
    try:
        oracle_lib.orc_set_anyhit_accounting(0)
        ref, st0 = s.render(8, max_depth=8)
        oracle_lib.orc_set_anyhit_accounting(1)
        img, st1 = s.render(8, max_depth=8)
    finally:
        oracle_lib.orc_set_anyhit_accounting(0)
    assert_bits_equal(img, ref, "image under the two accountings")
    assert st0["occludedTraced"] > 1000
    for k in ("raysTraced", "occludedTraced", "nHit", "nPx", "rngDraws"):
        assert st0[k] == st1[k], k
    assert st1["nBox"] < st0["nBox"] and st1["nTri"] < st0["nTri"]


def test_env_light_matches_reference():
    """InfiniteAreaLight::create / sample (light.cpp:30-128) and an env-lit render against the compiled reference's output
    (fixture written by make_golden.dump_env_light): CDF tables, sampled directions and radiance, all bit for bit."""
    z = np.load(os.path.join(G, "env_light.npz"))
    for name in ("sky", "black_rows"):
        w, h, black = (int(v) for v in z[f"{name}_size"])
        desc = T.cornell_scene(96, 96, with_teapot=False)
        desc.env = T.sky_env(w, h, black_rows=bool(black))
        s = T.OracleScene(desc)
        vp, hp = s.env_tables()
        assert_bits_equal(vp, z[f"{name}_vertical"], f"{name} vertical CDF")
        assert_bits_equal(hp, z[f"{name}_horizontal"], f"{name} horizontal CDF")
        d, c = s.env_sample(T.env_test_u(4096))
        assert_bits_equal(d, z[f"{name}_dir"], f"{name} sampled direction")
        assert_bits_equal(c, z[f"{name}_color"], f"{name} sampled radiance")
    desc = T.cornell_scene(96, 96, with_teapot=False)
    desc.env = T.sky_env(64, 32)
    s = T.OracleScene(desc)
    x0, y0, x1, y1 = (int(v) for v in z["rect"])
    rgb, st = s.render_rect((x0, y0, x1, y1), 16, max_depth=14)
    assert_bits_equal(rgb, z["rgb"], "env-lit Cornell crop")
    assert st["raysTraced"] == int(z["rays"][0]) and st["occludedTraced"] == int(z["rays"][1])


def test_reference_own_test_program_passes():
    """tests/tests.cpp of the reference, compiled from its own sources by oracle/Makefile (vecmath lane operations, the
    `t == 1.0f` triangle that is also leaf fixture 0, the thread pool): a failed assertion prints and traps."""
    exe = os.path.join(T.ORACLE_DIR, "_ref", "ref_tests")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_tests is only built where /root/reference exists")
    import subprocess
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "assertion failed" not in p.stdout
    for name in ("test_vecmath()", "test_triangle_intersection()", "test_thread_pool()"):
        assert name in p.stdout


def test_gbuffer_visualizer_matches_reference():
    """GbufferVisualizer (gbuffer_visualizer.cpp:17-51) with the single jittered camera ray (camera.cpp:12-33) against the
    compiled reference's images (fixture written by make_golden.dump_gbuffer)."""
    z = np.load(os.path.join(G, "gbuffer.npz"))
    s = T.OracleScene(T.cornell_scene(96, 96, with_teapot=True))
    rect = tuple(int(v) for v in z["rect"])
    for k in (0, 1, 2):
        assert_bits_equal(s.gbuffer(k, rect), z[f"kind{k}"], f"gbuffer kind {k}")


def test_c1_full_image_digest_matches_reference():
    """The oracle's whole C1 image (512x512, 16 spp, depth 14) against the digest of the compiled reference's render."""
    import hashlib
    z = np.load(os.path.join(G, "c1_full_checksums.npz"))
    s = T.OracleScene(T.cornell_scene(512, 512, with_teapot=True))
    rgb, st = s.render(16, max_depth=14)
    assert st["raysTraced"] == int(z["rays"][0]) and st["occludedTraced"] == int(z["rays"][1])
    digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(rgb, dtype="<f4").tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(digest, z["sha256"])
    f64 = rgb.reshape(-1, 3).astype(np.float64)
    assert np.array_equal(f64.sum(0), z["sum"]) and np.array_equal((f64 * f64).sum(0), z["sumsq"])


@pytest.mark.parametrize("name,setup,kw,w,h", [("c2", "setup_bunny_standin", dict(tris=20000), 192, 192),
                                               ("c3", "setup_atrium_standin", dict(tris=40000), 192, 108)])
def test_scene_digests_match_reference(name, setup, kw, w, h):
    """The oracle's whole images of the C2-class and C3-class test scenes against the compiled reference's digests."""
    import hashlib
    import prt_amd
    z = np.load(os.path.join(G, "scene_digests.npz"))
    scene, camera, exposure = getattr(prt_amd, setup)(w, h, **kw)
    rgb, st = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure)).render(16, max_depth=14)
    assert st["raysTraced"] == int(z[f"{name}_rays"][0]) and st["occludedTraced"] == int(z[f"{name}_rays"][1])
    digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(rgb, dtype="<f4").tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(digest, z[f"{name}_sha256"])


def test_c5_tile_row_fixture_is_oracle_output_for_one_tile():
    """tests/golden/c5_tile_rows.npz (tools/c5_split_check.py): SHA-256 digests of 16x16 tiles of BASELINE config 5 (3840x2160,
    1024 spp, depth 12, exposure 64) as the ORACLE rendered them -- hours of CPU, done once.  Here: the fixture is well formed, and
    one of its tiles, rendered again by the oracle, gives the stored digest (a tile of 1024 spp on a 5 M triangle scene is about a
    minute of one core's work; the scene comes from the product's generator, as it does for the GPU test that uses the fixture)."""
    import hashlib
    import prt_amd
    z = np.load(os.path.join(G, "c5_tile_rows.npz"))
    W, H, spp, depth, exposure = int(z["width"]), int(z["height"]), int(z["spp"]), int(z["max_depth"]), float(z["exposure"])
    assert (W, H, spp, depth, exposure, int(z["seed"])) == (3840, 2160, 1024, 12, 64.0, 12345)
    rows = [int(r) for r in z["rows"]]
    assert len(set(rows)) == len(rows) and all(0 <= r < H // 16 for r in rows) and z["sha"].shape == (len(rows), W // 16, 32)
    if os.environ.get("PRT_SKIP_SLOW_ORACLE"):
        return
    scene, camera, _ = prt_amd.setup_atrium_standin(W, H, tris=5000000, seed=5, emissive_fraction=0.1, light=False)
    s = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure))
    q, c = len(rows) // 2, 7  # a tile near the left edge (cheap pixels) of the middle fixture row
    x0, y0 = c * 16, rows[q] * 16
    tile, _ = s.render_rect((x0, y0, x0 + 15, y0 + 15), spp, max_depth=depth, stats=False)
    assert hashlib.sha256(np.ascontiguousarray(tile, dtype=np.float32).view(np.uint32).tobytes()).digest() == z["sha"][q, c].tobytes()


@pytest.mark.parametrize("name,setup,kw,tile", [
    ("c2", "setup_bunny_standin", {}, (31, 40)),
    ("c3", "setup_atrium_standin", dict(tris=262000, seed=1), (60, 33)),
    ("c4", "setup_atrium_standin", dict(tris=2500000, seed=4), (11, 50)),
])
def test_whole_frame_digest_fixtures_are_oracle_output(name, setup, kw, tile):
    """tests/golden/frame_digests_c{2,3,4}.npz (tools/whole_frame_digests.py): SHA-256 of every 16x16 tile of a BASELINE
    configuration's whole frame as the ORACLE rendered it, which the GPU suite compares its whole frames with.  Here: each fixture
    is well formed and one of its tiles, rendered again by the oracle, gives the stored digest."""
    import hashlib
    import prt_amd
    z = np.load(os.path.join(G, f"frame_digests_{name}.npz"))
    W, H, spp, depth, exposure = int(z["width"]), int(z["height"]), int(z["spp"]), int(z["max_depth"]), float(z["exposure"])
    assert z["sha"].shape == ((H + 15) // 16, (W + 15) // 16, 32) and int(z["seed"]) == 12345 and int(z["rays"]) >= spp * W * H
    scene, camera, exp = getattr(prt_amd, setup)(W, H, **kw)
    assert float(np.float32(exp)) == exposure
    s = T.OracleScene(T.scene_desc_from_product(scene, camera, exp))
    c, r = tile
    x0, y0 = c * 16, r * 16
    crop, _ = s.render_rect((x0, y0, min(W, x0 + 16) - 1, min(H, y0 + 16) - 1), spp, max_depth=depth, stats=False)
    assert hashlib.sha256(np.ascontiguousarray(crop, dtype=np.float32).view(np.uint32).tobytes()).digest() == z["sha"][r, c].tobytes()
