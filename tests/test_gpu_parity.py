"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against
the oracle on the same inputs and against the golden vectors of the compiled reference.

Bar: BIT-EXACT.  The path is IEEE binary32 + integer work with no contraction, and the kernels carry their own
sin/cos/pow that reproduce the box's libm, so per-pixel radiance is not merely within the north star's 1e-4
relative tolerance -- it is identical; every comparison below is on bit patterns (tolerance 0)."""
import os

import numpy as np
import pytest

import prt_amd
import prt_testlib as T

pytestmark = pytest.mark.gpu
G = T.GOLDEN


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a, b = bits(a), bits(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = np.nonzero(a != b)
    assert len(bad[0]) == 0, f"{what}: {len(bad[0])} of {a.size} differ, first at {tuple(int(x[0]) for x in bad)}"


@pytest.fixture(scope="module", autouse=True)
def gpu_event_accounting():
    """The oracle's nBox/nTri/nTap for occlusion queries follow the GPU's near-first visit in this module (the images
    and ray counts are the reference traversal's either way; the oracle aborts if the two visits ever disagree)."""
    T.oracle().orc_set_anyhit_accounting(1)
    yield
    T.oracle().orc_set_anyhit_accounting(0)


@pytest.fixture(scope="module")
def tracer():
    prt_amd.build()
    t = prt_amd.PathTracer()
    yield t
    t.close()


@pytest.fixture(scope="module")
def rows():
    """A context of the TEST build of the library (libprt_hip_test.so: the product's sources + the row-level entry points
    of include/prt_hip_test.h).  Whole-image tests run on `tracer`, the product library."""
    prt_amd.build()
    t = prt_amd.PathTracer(test_entry_points=True)
    yield t
    t.close()


@pytest.fixture(scope="module")
def c1(tracer):
    """Config C1 scene (Cornell + teapot, 512x512) uploaded, with the oracle's twin."""
    scene, camera, exposure = prt_amd.setup_cornell_box(512, 512, teapot_mesh=T.teapot_product_mesh())
    desc = T.scene_desc_from_product(scene, camera, exposure)
    return scene, camera, desc


def upload(tracer, scene, camera):
    tracer.upload_scene(scene)
    tracer.set_camera(camera)


# ----------------------------------------------------------------------------- leaf rows (a3, a4, a10, box tests)
def test_leaf_math_matches_reference_vectors(rows):
    z = np.load(os.path.join(G, "leaf_vectors.npz"))
    out = rows.test_leaf(z["inputs"])
    ref = z["outputs"]
    cols = [c for c in range(23) if c not in (8, 9, 10, 11, 15)]  # scalar triangle + SoA box->t are not on the path
    a, b = out[:, cols], ref[:, cols]
    nan = np.isnan(a) & np.isnan(b)
    assert_bits_equal(np.where(nan, 0, a), np.where(nan, 0, b), "leaf vectors")
    assert out[0, 0] == 1.0  # the reference's own known answer, tests/tests.cpp:109-127


def test_sincos_matches_libm_on_all_path_arguments(rows, oracle_lib):
    k = np.arange(1 << 23, dtype=np.uint32)
    r1 = (k | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    theta = (np.float32(2.0) * np.float32(3.14159265358979323846) * r1).astype(np.float32)
    s, c = rows.test_sincos(theta)
    rs, rc = np.zeros_like(theta), np.zeros_like(theta)
    oracle_lib.orc_libm_sincos(len(theta), T.vptr(theta), T.vptr(rs), T.vptr(rc))
    assert_bits_equal(s, rs, "sinf")
    assert_bits_equal(c, rc, "cosf")
    # Environment light (light.cpp:121-125): theta = 2 pi (u + 0.5) and phi = pi v with (u, v) from the inverted CDFs -- a bright
    # texel in the first row or column makes them any size and sign (profiles/r02_parity_sweeps.txt: found by the sweep), so the
    # whole float line matters: every 997th bit pattern, all of [-300, 300] around the switch of reductions at 120, infinities, NaN.
    bits = np.concatenate([np.arange(0, 1 << 32, 997, dtype=np.uint64).astype(np.uint32),
                           np.arange(np.float32(100).view(np.uint32), np.float32(300).view(np.uint32), dtype=np.uint32),
                           np.arange(np.float32(100).view(np.uint32), np.float32(300).view(np.uint32), dtype=np.uint32) | np.uint32(0x80000000),
                           np.array([0x7F800000, 0xFF800000, 0x7FC00000, 0x7F7FFFFF, 0xFF7FFFFF], dtype=np.uint32)])
    y = bits.view(np.float32)
    s, c = rows.test_sincos(y)
    rs, rc = np.zeros_like(y), np.zeros_like(y)
    oracle_lib.orc_libm_sincos(len(y), T.vptr(y), T.vptr(rs), T.vptr(rc))
    nan = np.isnan(rs)
    assert np.array_equal(np.isnan(s), nan) and np.array_equal(np.isnan(c), np.isnan(rc))
    assert_bits_equal(s[~nan], rs[~nan], "sinf over the float line")
    assert_bits_equal(c[~nan], rc[~nan], "cosf over the float line")


def test_powf_matches_libm(rows, oracle_lib):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 1, 1 << 20), np.arange(256) / 255.0, [0.0, 1.0, 1e-30, 2.0 ** -24], rng.uniform(1, 2, 1 << 16),
                        np.exp(rng.uniform(np.log(1e-44), np.log(1e-7), 1 << 16))]).astype(np.float32)  # a texel mix can round just outside [0, 1]
    y = rows.test_powf(x)
    ref = np.zeros_like(x)
    oracle_lib.orc_libm_powf22(len(x), T.vptr(x), T.vptr(ref))
    assert_bits_equal(y, ref, "powf(x, 2.2)")


def test_camera_packets_match_reference(rows, c1):
    scene, camera, _ = c1
    upload(rows, scene, camera)
    z = np.load(os.path.join(G, "camera_packets.npz"))
    for (x, y, state), ref in zip(z["xys"], z["out"]):
        out = rows.test_camera(int(x), int(y), int(state))
        assert_bits_equal(out[:92], ref[:92], f"camera packet at ({x},{y})")


# ----------------------------------------------------------------------------- traversal rows (a5-a9, a11)
def test_traversals_match_reference_rays(rows, c1):
    scene, camera, _ = c1
    upload(rows, scene, camera)
    z = np.load(os.path.join(G, "rays_cornell_teapot.npz"))
    far = float(z["max_t"])
    for mode, key in ((0, "single"), (1, "packet")):
        got = rows.trace_rays(mode, z["org"], z["dir"], far)
        ref = z[key].view(T.HIT_DTYPE).reshape(-1)
        assert_bits_equal(got["t"], ref["t"], key + ".t")
        hit = ref["t"] != -1
        for f in ("i", "j", "k"):
            assert_bits_equal(got[f][hit], ref[f][hit], key + "." + f)
        assert (got["primId"][hit] == ref["primId"][hit]).all() and (got["meshId"][hit] == ref["meshId"][hit]).all()
    assert (rows.trace_rays(2, z["org"], z["dir"], far)["t"] == z["occluded_single"]).all()
    assert (rows.trace_rays(3, z["org"], z["dir"], far)["t"] == z["occluded_packet"]).all()


def test_traversals_match_oracle_on_fresh_rays(rows, c1):
    scene, camera, desc = c1
    upload(rows, scene, camera)
    s = T.OracleScene(desc)
    rng = np.random.default_rng(99)
    n = 2048
    org = rng.uniform([-0.95, 0.05, -0.95], [0.95, 1.9, 0.95], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d[: n // 8, 0] = 0.0  # exactly-zero components: inf invDir, NaN slabs
    d[n // 8: n // 4, 2] = d[n // 8: n // 4, 1]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    far = float(np.float32(2.0) * np.float32(s.radius()))
    os_, oc1 = s.intersect_single(org, d, far)
    op, oc8 = s.intersect_packet(org, d, far)
    for mode, ref in ((0, os_), (1, op)):
        got = rows.trace_rays(mode, org, d, far)
        assert got.tobytes() == ref.tobytes(), f"mode {mode}"
    assert (rows.trace_rays(2, org, d, far)["t"] == oc1).all()
    assert (rows.trace_rays(3, org, d, far)["t"] == oc8).all()


# ----------------------------------------------------------------------------- the whole path (a1, a2, a12, a13)
def test_radiance_matches_reference_crop(tracer, c1):
    """C1 (Cornell + teapot 512x512, 16 spp, depth cap 14): the 64x64 crop rendered by the compiled reference."""
    scene, camera, _ = c1
    upload(tracer, scene, camera)
    z = np.load(os.path.join(G, "radiance_c1_crop.npz"))
    x0, y0, x1, y1 = (int(v) for v in z["rect"])
    rgb = tracer.trace_block(x0, y0, x1, y1, 16)
    st = tracer.last_stats
    assert_bits_equal(rgb, z["rgb"], "C1 crop radiance")
    assert st["raysTraced"] == int(z["rays"][0]) and st["occludedTraced"] == int(z["rays"][1])
    assert st["nPx"] == 64 * 64 and st["stackOverflow"] == 0


def test_radiance_matches_reference_cornell_only(tracer):
    scene, camera, _ = prt_amd.setup_cornell_box(128, 128)
    upload(tracer, scene, camera)
    z = np.load(os.path.join(G, "radiance_c1_crop.npz"))
    rgb = tracer.render(16)
    assert_bits_equal(rgb, z["cornell_only_rgb"], "cornell-only radiance")
    assert tracer.last_stats["raysTraced"] == int(z["cornell_only_rays"][0]) == 1126145


@pytest.mark.parametrize("max_depth", [4, 14])
def test_c1_full_image_matches_oracle(tracer, c1, max_depth):
    """Config C1 whole image, both the BASELINE depth (4) and the reference's literal (14); also the traffic
    counters that define the algorithmic bytes of the roofline."""
    scene, camera, desc = c1
    upload(tracer, scene, camera)
    rgb = tracer.render(16, max_depth=max_depth, count_traffic=True)
    st = tracer.last_stats
    s = T.OracleScene(desc)
    ref, ost = s.render(16, max_depth=max_depth)
    assert_bits_equal(rgb, ref, f"C1 depth {max_depth}")
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])


def test_directional_light_scene_matches_oracle(tracer):
    """Config C2's scene (Cornell + bunny-class stand-in + directional light) at a test size: exercises both
    occlusion traversals (packet when more than two paths are alive, single otherwise)."""
    scene, camera, exposure = prt_amd.setup_bunny_standin(160, 160, tris=20000)
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    rgb = tracer.render(16, count_traffic=True)
    st = tracer.last_stats
    s = T.OracleScene(desc)
    ref, ost = s.render(16)
    assert ost["occludedTraced"] > 0
    assert_bits_equal(rgb, ref, "C2-class radiance")
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])


def test_textured_atrium_matches_oracle(tracer):
    """Config C3's scene class (Sponza-like stand-in: alpha-masked cards, bump-mapped floor, directional light,
    vertex normals, depth cap 8) at a test size: rows a12 (surface fetch) and a14 (alpha / bump / diffuse taps, powf)."""
    scene, camera, exposure = prt_amd.setup_atrium_standin(192, 108, tris=40000)
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    rgb = tracer.render(16, max_depth=8, count_traffic=True)
    st = tracer.last_stats
    s = T.OracleScene(desc)
    ref, ost = s.render(16, max_depth=8)
    assert ost["nTap"] > 0 and ost["occludedTraced"] > 0
    assert_bits_equal(rgb, ref, "C3-class radiance")
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])


def test_emissive_scene_without_light_matches_oracle(tracer):
    """Zero-Day-class stand-in (C5's scene class): emissive cards, no directional light, exposure 64 (main.cpp:93-105)."""
    scene, camera, _ = prt_amd.setup_atrium_standin(128, 72, tris=30000, emissive_fraction=0.5, light=False)
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, 64.0)
    rgb = tracer.render(16, max_depth=12, exposure=64.0)
    st = tracer.last_stats
    ref, ost = T.OracleScene(desc).render(16, max_depth=12)
    assert ost["occludedTraced"] == 0 and st["occludedTraced"] == 0
    assert_bits_equal(rgb, ref, "C5-class radiance")
    assert st["raysTraced"] == ost["raysTraced"]


def test_ragged_rectangles_and_rank_interleave(tracer, c1):
    """Edge cases of the boundary: 1-pixel and ragged rectangles, image sizes that are not multiples of the tile,
    and the multi-GPU tile interleave (union of ranks == single-rank image)."""
    scene, camera, _ = prt_amd.setup_cornell_box(100, 52)
    upload(tracer, scene, camera)
    full = tracer.render(8)
    for rect in ((0, 0, 0, 0), (99, 51, 99, 51), (3, 5, 40, 17), (16, 16, 31, 31), (90, 0, 99, 51)):
        x0, y0, x1, y1 = rect
        part = tracer.trace_block(x0, y0, x1, y1, 8)
        assert_bits_equal(part, full[y0:y1 + 1, x0:x1 + 1], f"rect {rect}")
    import ctypes as C
    acc = np.zeros((52, 100, 3), dtype=np.float32)
    for rank in range(3):
        tracer.render_async(0, 0, 99, 51, 8, rank=rank, nranks=3)
        img = np.zeros((52, 100, 3), dtype=np.float32)
        prt_amd._check(prt_amd.lib().prt_hip_download(tracer._ctx, img.ctypes.data_as(C.c_void_p), 0, 0, 99, 51), "download")
        st = tracer.stats()
        ty, tx = np.meshgrid(np.arange(52) // 16, np.arange(100) // 16, indexing="ij")
        own = ((ty * 7 + tx) % 3) == rank
        assert st["nPx"] == int(own.sum())
        acc[own] = img[own]
    assert_bits_equal(acc, full, "rank-interleaved union")


def test_error_behaviour(tracer, rows, c1):
    scene, camera, _ = c1
    upload(tracer, scene, camera)
    with pytest.raises(prt_amd.PrtError, match="outside the image"):
        tracer.trace_block(0, 0, 512, 10, 8)
    upload(rows, scene, camera)
    with pytest.raises(prt_amd.PrtError):
        rows.trace_rays(0, np.zeros((7, 3)), np.zeros((7, 3)), 1.0)
    with pytest.raises(prt_amd.PrtError, match="libprt_hip_test"):  # the product library has no row-level entry points
        tracer.trace_rays(0, np.zeros((8, 3)), np.zeros((8, 3)), 1.0)


# ----------------------------------------------------------------------------- BASELINE full size: properties

def assert_frame_equals_oracle_digests(img, stats, name, spp, depth, exposure):
    """The WHOLE frame against the oracle's frame, tile by tile: tests/golden/<name> holds the SHA-256 of every 16x16 tile of the frame
    as the oracle rendered it at the configuration's own sample count (tools/whole_frame_digests.py: minutes to an hour of all
    cores, done once) and the oracle's ray counts."""
    import hashlib
    z = np.load(os.path.join(G, name))
    H, W, _ = img.shape
    assert (int(z["width"]), int(z["height"]), int(z["spp"]), int(z["max_depth"]), float(z["exposure"]), int(z["seed"])) == (W, H, spp, depth, float(exposure), 12345)
    ty, tx = (H + 15) // 16, (W + 15) // 16
    assert z["sha"].shape == (ty, tx, 32)
    differing = []
    for r in range(ty):
        for c in range(tx):
            tile = np.ascontiguousarray(img[r * 16:r * 16 + 16, c * 16:c * 16 + 16]).view(np.uint32).tobytes()
            if hashlib.sha256(tile).digest() != z["sha"][r, c].tobytes():
                differing.append((c, r))
    assert not differing, f"{len(differing)} of {ty * tx} tiles differ from the oracle's frame ({name}), first {differing[:4]}"
    assert (stats["raysTraced"], stats["occludedTraced"]) == (int(z["rays"]), int(z["occluded"]))


def test_full_size_c2_properties(tracer):
    """Config C2 at BASELINE size (1024x1024, 64 spp): too slow for the CPU oracle in a unit test, so check
    size-independent properties -- determinism, the closed form of the primary ray count, invariance to how the
    image is cut into launches, and exact agreement with the oracle on sampled tiles."""
    scene, camera, exposure = prt_amd.setup_bunny_standin(1024, 1024)
    upload(tracer, scene, camera)
    a = tracer.render(64)
    sa = tracer.last_stats
    b = tracer.render(64)
    sb = tracer.last_stats
    assert a.tobytes() == b.tobytes() and sa["raysTraced"] == sb["raysTraced"]
    assert sa["nPx"] == 1024 * 1024 and sa["raysTraced"] >= 64 * 1024 * 1024
    assert np.isfinite(a).all() and (a >= 0).all()
    assert_frame_equals_oracle_digests(a, sa, "frame_digests_c2.npz", 64, 14, exposure)  # the whole frame == the oracle's frame
    top = tracer.trace_block(0, 0, 1023, 511, 64)
    bot = tracer.trace_block(0, 512, 1023, 1023, 64)
    assert np.concatenate([top, bot], 0).tobytes() == a.tobytes()
    desc = T.scene_desc_from_product(scene, camera, exposure)
    s = T.OracleScene(desc)
    for (x0, y0) in ((496, 600), (200, 300), (0, 0), (1008, 1008)):
        ref, _ = s.trace_block(x0, y0, x0 + 15, y0 + 15, 64)
        assert_bits_equal(a[y0:y0 + 16, x0:x0 + 16], ref, f"tile at {(x0, y0)}")


# ----------------------------------------------------------------------------- stack spill, overflow, large scenes
def _geometric_soup(kmax):
    """Triangles at x = 2^k: binned SAH peels them off a few at a time, giving a tree about kmax/2 deep."""
    tris = []
    for k in range(kmax):
        x = np.float32(2.0) ** k
        s = x * np.float32(0.25)
        tris.append([[x, -s, -s], [x, s, -s], [x, 0, s]])
    pos = np.asarray(tris, dtype=np.float32).reshape(-1, 3)
    idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
    mats = np.array([T.make_material(diffuse=(0.5, 0.5, 0.5))], dtype=T.MATERIAL_DTYPE)
    return idx, pos, np.zeros(len(idx), dtype=np.uint32), mats


def _tree_depth(nodes):
    depth, stack = 0, [(0, 1)]
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        if nodes["primCount"][i] == 0xF:
            stack.append((i + 1, d + 1))
            stack.append((int(nodes["primOrSecondNodeIndex"][i]), d + 1))
    return depth


def test_deep_tree_uses_the_stack_spill_and_matches_oracle(tracer, rows):
    """A tree deeper than the 12 stack entries kept in LDS (PRT_STACK_LDS; 6 for the packet traversal; the rest of the 64 spill to HBM)."""
    idx, pos, pm, mats = _geometric_soup(56)
    scene = prt_amd.Scene()
    scene.add(prt_amd.Mesh.from_arrays(idx, pos, pm, mats.view(prt_amd.MATERIAL_DTYPE)))
    camera = prt_amd.Camera().create((-4, 0, 0), (1, 0, 0), 32, 32)
    upload(tracer, scene, camera)
    upload(rows, scene, camera)
    desc = T.scene_desc_from_product(scene, camera)
    assert 17 < _tree_depth(desc.product_arrays["meshes"][0]["nodes"]) < 60  # deeper than the 12 LDS entries
    s = T.OracleScene(desc)
    rng = np.random.default_rng(1)
    n = 1024
    org = np.zeros((n, 3), dtype=np.float32)
    org[:, 0] = -3
    org[:, 1:] = rng.uniform(-0.2, 0.2, (n, 2))
    d = np.zeros((n, 3), dtype=np.float32)
    d[:, 0] = 1
    d[:, 1:] = rng.uniform(-0.02, 0.02, (n, 2))
    d[: n // 2, 0] = -1  # rays that leave the scene: every box behind the origin still passes the reference's test
    org[: n // 2, 0] = np.float32(2.0) ** 57
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    far = float(np.float32(2.0) * np.float32(s.radius()))
    os_, oc1 = s.intersect_single(org, d, far)
    op, oc8 = s.intersect_packet(org, d, far)
    assert rows.trace_rays(0, org, d, far).tobytes() == os_.tobytes()
    assert rows.trace_rays(1, org, d, far).tobytes() == op.tobytes()
    assert (rows.trace_rays(2, org, d, far)["t"] == oc1).all()
    assert (rows.trace_rays(3, org, d, far)["t"] == oc8).all()
    rgb = tracer.render(8)
    ref, _ = s.render(8)
    assert_bits_equal(rgb, ref, "deep-tree radiance")


def _chain_scene_desc(D=80):
    """A hand-made chain BVH D levels deep (every level pushes its second child and descends into the first), as a raw C-ABI
    scene descriptor; returns (descriptor, objects that must stay alive while it is used)."""
    import ctypes as C
    n_nodes, n_prims = 2 * D + 1, D + 1
    nodes = np.zeros(n_nodes, dtype=T.NODE_DTYPE)
    nodes["lower"] = (-1e6, -1e6, -1e6)
    nodes["upper"] = (1e6, 1e6, 1e6)
    for i in range(D):
        nodes["primCount"][i] = 0xF
        nodes["primOrSecondNodeIndex"][i] = 2 * D - i
    for k in range(n_prims):  # leaves in DFS order hold triangle k
        nodes["primCount"][D + k] = 1
        nodes["primOrSecondNodeIndex"][D + k] = k
        nodes["triVectorIndex"][D + k] = k
    pos = np.zeros((3 * n_prims, 3), dtype=np.float32)
    for k in range(n_prims):
        pos[3 * k:3 * k + 3] = [[5000 + k, 0, 0], [5000 + k, 1, 0], [5000 + k, 0, 1]]
    idx = np.arange(3 * n_prims, dtype=np.uint32)
    remap = np.arange(n_prims, dtype=np.uint32)
    pm = np.zeros(n_prims, dtype=np.uint32)
    mats = np.array([T.make_material(diffuse=(0.5, 0.5, 0.5))], dtype=T.MATERIAL_DTYPE)
    md = prt_amd.MeshDesc()
    md.nodeCount, md.nodes = n_nodes, nodes.ctypes.data_as(C.POINTER(prt_amd.BvhNode))
    md.primCount, md.primRemapping = n_prims, remap.ctypes.data_as(C.POINTER(C.c_uint32))
    md.vertexCount, md.indices = 3 * n_prims, idx.ctypes.data_as(C.POINTER(C.c_uint32))
    md.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
    md.materialCount, md.primMaterial = 1, pm.ctypes.data_as(C.POINTER(C.c_uint32))
    md.materials = mats.ctypes.data_as(C.POINTER(prt_amd.Material))
    sd = prt_amd.SceneDesc()
    sd.meshCount, sd.meshes, sd.radius = 1, C.pointer(md), 1e4
    return sd, (nodes, pos, idx, remap, pm, mats, md)


def test_stack_overflow_is_reported_like_the_reference_assert(rows):
    """The reference asserts when its 64-entry stack overflows (bvh.cpp:552, 627); the C-ABI returns PRT_HIP_ESTACK.
    The binned-SAH builder does not produce such trees from sane input, so a hand-made chain BVH (every level pushes its
    second child and descends into the first) is uploaded through the C-ABI descriptor."""
    import ctypes as C
    sd, keep = _chain_scene_desc(80)
    rows._chk(rows._L.prt_hip_upload_scene(rows._ctx, C.byref(sd)), "upload")
    org = np.tile(np.array([[0.0, 0.2, 0.2]], dtype=np.float32), (8, 1))
    d = np.tile(np.array([[1.0, 0, 0]], dtype=np.float32), (8, 1))
    for mode in (0, 1, 2, 3):
        rows.trace_rays(mode, org, d, 100.0)
        if mode == 0:
            rows.stats()  # near-first single traversal pops as it goes on this chain: no overflow
            continue
        with pytest.raises(prt_amd.PrtError, match="64 stack entries"):
            rows.stats()


def test_an_earlier_frames_error_survives_later_renders(tracer, c1):
    """A stack overflow in ONE frame of an asynchronous sequence must not be wiped by the frames rendered after it (every render
    clears its own counters): the frame kernel also records it in words no render clears, and prt_hip_download /
    prt_hip_get_stats report it -- once; the call after that is clean again."""
    import ctypes as C
    sd, keep = _chain_scene_desc(80)
    t = prt_amd.PathTracer(device=0, max_depth=4, seed=1)
    try:
        t._chk(t._L.prt_hip_upload_scene(t._ctx, C.byref(sd)), "upload")
        t.set_camera(prt_amd.Camera().create((0.0, 0.2, 0.2), (1.0, 0.0, 0.0), 32, 32))
        t.render_async(0, 0, 31, 31, 8)          # scatter rays walk the chain: more than 64 stack entries
        scene, camera, _ = c1
        t.upload_scene(scene)
        t.set_camera(camera)
        t.render_async(0, 0, 63, 63, 8)          # a clean frame behind it
        img = np.zeros((512, 512, 3), dtype=np.float32)
        rc = t._L.prt_hip_download(t._ctx, img.ctypes.data_as(C.c_void_p), 0, 0, 63, 63)
        assert rc == -6 and b"64 stack entries" in t._L.prt_hip_last_error()  # PRT_HIP_ESTACK, pixels delivered all the same
        assert np.abs(img[:64, :64]).sum() > 0
        with pytest.raises(prt_amd.PrtError, match="64 stack entries"):
            t.stats()
        t.stats()                                 # reported once
        t.render_async(0, 0, 63, 63, 8)
        t.stats()
    finally:
        t.close()


def test_an_overflow_frame_raises_once_and_the_next_render_is_clean(c1):
    """prt_hip_download reports a sticky error without clearing it, so render() / trace_block() / gbuffer() used to
    fail for ever after one bad frame.  They consume it now: the overflow frame raises, the same tracer then renders a clean
    scene, whose image equals a fresh tracer's."""
    import ctypes as C
    sd, keep = _chain_scene_desc(80)
    t = prt_amd.PathTracer(device=0, max_depth=4, seed=1)
    try:
        t._chk(t._L.prt_hip_upload_scene(t._ctx, C.byref(sd)), "upload")
        t.set_camera(prt_amd.Camera().create((0.0, 0.2, 0.2), (1.0, 0.0, 0.0), 32, 32))
        with pytest.raises(prt_amd.PrtError, match="64 stack entries"):
            t.render(8)
        scene, camera, _ = c1
        t.upload_scene(scene)
        t.set_camera(camera)
        a = t.trace_block(0, 0, 63, 63, 8)     # clean again: nothing left over from the bad frame
        g = t.gbuffer(1, 0, 0, 63, 63)
        assert np.isfinite(g).all()
        t2 = prt_amd.PathTracer(device=0, max_depth=4, seed=1)
        try:
            t2.upload_scene(scene)
            t2.set_camera(camera)
            assert_bits_equal(a, t2.trace_block(0, 0, 63, 63, 8), "after an overflow frame")
        finally:
            t2.close()
    finally:
        t.close()


def test_large_scene_4k_one_launch_matches_oracle_tiles(tracer):
    """C5-class at its BASELINE resolution: 3840x2160 (8.3 M pixel groups, ONE launch of the frame kernel: its per-launch
    state does not grow with the image), a 1 M-triangle emissive scene without directional light, depth cap 12, exposure 64;
    8 spp to keep the oracle side short.  Sample tiles from the corners and the middle must match the oracle bit for bit."""
    scene, camera, _ = prt_amd.setup_atrium_standin(3840, 2160, tris=1000000, emissive_fraction=0.25, light=False)
    upload(tracer, scene, camera)
    img = tracer.render(8, max_depth=12, exposure=64.0)
    st = tracer.last_stats
    assert st["nPx"] == 3840 * 2160 and st["stackOverflow"] == 0
    desc = T.scene_desc_from_product(scene, camera, 64.0)
    s = T.OracleScene(desc)
    for (x0, y0) in ((0, 0), (1904, 1072), (3824, 2144), (640, 1600)):
        ref, _ = s.trace_block(x0, y0, x0 + 15, y0 + 15, 8, max_depth=12)
        assert_bits_equal(img[y0:y0 + 16, x0:x0 + 16], ref, f"4K tile at {(x0, y0)}")


def test_infinite_area_light_matches_reference_and_oracle(tracer):
    """environment light sampled in the kernels (bisection instead of the reference's linear CDF scan, the two
    extra draws per diffuse bounce, per-slot light direction for the occlusion rays).  Cornell box lit by a seeded float
    map: the 64x64 crop equals the compiled reference's output bit for bit (tests/golden/env_light.npz); a map with
    all-black rows/columns (zero-pdf entries, NaN CDF rows) and the specular teapot equals the oracle, counters included."""
    z = np.load(os.path.join(G, "env_light.npz"))
    scene, camera, exposure = prt_amd.setup_cornell_box(96, 96)
    scene.set_infinite_area_light(T.sky_env(64, 32))
    upload(tracer, scene, camera)
    x0, y0, x1, y1 = (int(v) for v in z["rect"])
    rgb = tracer.trace_block(x0, y0, x1, y1, 16)
    assert_bits_equal(rgb, z["rgb"], "env-lit Cornell crop vs compiled reference")
    st = tracer.last_stats
    assert st["raysTraced"] == int(z["rays"][0]) and st["occludedTraced"] == int(z["rays"][1])

    scene, camera, exposure = prt_amd.setup_cornell_box(128, 128, teapot_mesh=T.teapot_product_mesh())
    scene.set_infinite_area_light(T.sky_env(48, 24, black_rows=True))
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    rgb = tracer.render(16, max_depth=8, count_traffic=True)
    st = tracer.last_stats
    s = T.OracleScene(desc)
    ref, ost = s.render(16, max_depth=8)
    assert ost["occludedTraced"] > 0
    assert_bits_equal(rgb, ref, "env-lit Cornell + teapot vs oracle")
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    # A very bright texel in the first row and column: entry 0 of both CDFs then exceeds most draws, the scan settles on the first
    # entry that differs from its predecessor and the offsets (u - prev) / pdf are large and negative -- the angles theta and phi
    # leave every small range (sinf / cosf by the reduction for |x| >= 120).  Found by tools/misc_sweep.py.
    env = T.sky_env(24, 12)
    env[0, 0, :3] = 5000.0
    scene, camera, exposure = prt_amd.setup_cornell_box(96, 64)
    scene.set_infinite_area_light(env)
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    rgb = tracer.render(16, max_depth=6, count_traffic=True)
    st = tracer.last_stats
    ref, ost = T.OracleScene(desc).render(16, max_depth=6)
    assert_bits_equal(rgb, ref, "environment map with a bright first texel")
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    # a directional light set afterwards switches the environment light off (scene.h:30-35)
    scene.set_directional_light((0.2, 1.0, 0.2), (3.0, 3.0, 3.0))
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    assert desc.env is None
    rgb = tracer.render(8, max_depth=4)
    ref, _ = T.OracleScene(desc).render(8, max_depth=4)
    assert_bits_equal(rgb, ref, "directional light after environment light")


def test_nan_rays_are_answered_without_the_full_walk(tracer):
    """A NaN shading normal (degenerate vertex normals in real assets) gives NaN scatter rays.  Under the reference's
    min/max semantics such a ray passes every box test and misses every triangle: the timed build answers "miss" at once,
    the counting build walks the whole tree like the reference; both must give the oracle's image, the latter its counters."""
    m = T.oracle_vertex_normals(T.load_cornell_mesh())
    n = m.normals.copy()
    floor = np.where(np.abs(m.positions[:, 1]) < 1e-6)[0]
    assert len(floor) >= 4
    n[floor[:2]] = np.nan  # two floor vertices: every hit on their triangles interpolates a NaN normal
    pm = prt_amd.Mesh.from_arrays(m.indices, m.positions, m.prim_material, m.materials.view(prt_amd.MATERIAL_DTYPE), normals=n)
    pm.calculate_bounds()
    scene = prt_amd.Scene()
    scene.add(pm)
    scene.set_directional_light((0.2, 1.0, 0.2), (3.0, 3.0, 3.0))
    camera = prt_amd.Camera().create((0, 0.965, 2.6), (0, 0, -1.0), 96, 96)
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, 1.0)
    s = T.OracleScene(desc)
    ref, ost = s.render(16, max_depth=8)
    assert np.isnan(ref).any(), "the scene must actually produce NaN paths"
    nan = np.isnan(ref)  # a NaN's sign and payload differ between SSE (0xffc00000) and the GPU (0x7fc00000): compare "is NaN"

    def same(img, what):
        assert np.array_equal(np.isnan(img), nan), what
        assert_bits_equal(img[~nan], ref[~nan], what)

    same(tracer.render(16, max_depth=8), "NaN-normal scene, timed build")
    st = tracer.last_stats
    assert st["raysTraced"] == ost["raysTraced"] and st["occludedTraced"] == ost["occludedTraced"]
    same(tracer.render(16, max_depth=8, count_traffic=True), "NaN-normal scene, counting build")
    st = tracer.last_stats
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])


def test_single_process_gather_of_several_contexts(tracer, c1):
    """prt_hip_gather (the single-process arrangement): three contexts render rank 0/1/2 of 3 of a ragged
    rectangle; the assembled image equals one context rendering it alone, bit for bit.  (Three contexts on one device
    here; on a multi-GPU host each would sit on its own.)"""
    scene, camera, desc = c1
    upload(tracer, scene, camera)
    rect = (37, 21, 300, 210)
    whole = tracer.trace_block(*rect, 8, max_depth=4)
    others = [prt_amd.PathTracer(device=0, max_depth=4, seed=12345) for _ in range(3)]
    try:
        for i, t in enumerate(others):
            t.upload_scene(scene)
            t.set_camera(camera)
            t.render_async(*rect, 8, rank=i, nranks=3)
        img = prt_amd.gather_contexts(others, *rect)
        x0, y0, x1, y1 = rect
        assert_bits_equal(img[y0:y1 + 1, x0:x1 + 1], whole, "gathered image")
        with pytest.raises(prt_amd.PrtError):
            prt_amd.gather_contexts(others[:2], *rect)  # rendered as 3 ranks, gathered as 2
    finally:
        for t in others:
            t.close()


def _host_bvh(idx, pos):
    import ctypes as C
    L = prt_amd.lib()
    n = len(idx)
    nodes_p, cnt, remap_p = C.POINTER(prt_amd.BvhNode)(), C.c_uint32(), C.POINTER(C.c_uint32)()
    assert L.prt_host_bvh_build(n, idx.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), 0, C.byref(nodes_p), C.byref(cnt),
                                C.byref(remap_p)) == 0
    nodes = np.frombuffer(C.string_at(nodes_p, cnt.value * C.sizeof(prt_amd.BvhNode)), dtype=T.NODE_DTYPE).copy()
    remap = np.ctypeslib.as_array(remap_p, shape=(n,)).copy()
    L.prt_host_free(nodes_p)
    L.prt_host_free(remap_p)
    return nodes, remap


def _assert_same_tree(nodes, remap, ref_nodes, ref_remap, what):
    assert len(nodes) == len(ref_nodes), (what, len(nodes), len(ref_nodes))
    for f in ("primOrSecondNodeIndex", "triVectorIndex", "primCount", "splitAxis"):
        assert (nodes[f] == ref_nodes[f]).all(), (what, f, int(np.argmax(nodes[f] != ref_nodes[f])))
    # bounds: equal as floats (the min / max of the same vertices; only the sign of a zero bound can depend on the merge order)
    assert np.array_equal(nodes["lower"], ref_nodes["lower"]) and np.array_equal(nodes["upper"], ref_nodes["upper"]), what
    assert (remap == ref_remap).all(), what


@pytest.mark.parametrize("n,seed", [(1, 0), (8, 1), (9, 2), (300, 3), (5000, 4), (70000, 5)])
def test_gpu_bvh_build_matches_host_builder_on_soups(tracer, n, seed):
    """prt_hip_build_bvh (level-parallel binned SAH on the device) produces the node array, leaf order and
    primRemapping of the host builder, which the CPU tests pin against the reference's own Bvh::build -- on the soups that
    exercise the zero-extent axis (bvh.cpp:66), the unsigned bucket compare (:88), NaN costs of empty sides (:98-125) and
    the fallback to the middle (:150-153)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-1, 1, (3 * n, 3)).astype(np.float32)
    if seed == 3:
        pos[:, 1] = 0.25
    if seed == 4:
        pos[: 3 * 64] = pos[0]
    if seed == 5:
        pos[: 3 * 2000, 0] = np.float32(0.5)  # a slab of triangles with equal x centroids
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    nodes, remap, ms = tracer.build_bvh(idx, pos)
    ref_nodes, ref_remap = _host_bvh(idx, pos)
    _assert_same_tree(nodes, remap, ref_nodes, ref_remap, f"soup n={n} seed={seed}")


def test_gpu_bvh_build_matches_reference_fixture_and_large_scenes(tracer):
    """The device-built arrays equal the REFERENCE builder's for the Cornell box and the teapot (tests/golden/
    bvh_cornell_teapot.npz, dumped from the compiled reference), and the host builder's on the 262 k and 2.5 M triangle
    stand-ins; build times are printed."""
    z = np.load(os.path.join(G, "bvh_cornell_teapot.npz"))
    desc = T.cornell_scene(512, 512)
    for i in range(2):
        mesh = desc.meshes[i]
        nodes, remap, ms = tracer.build_bvh(mesh.indices, mesh.positions)
        ref = z[f"nodes{i}"].view(T.NODE_DTYPE).reshape(-1)
        assert len(nodes) == len(ref), i
        for f in ("primOrSecondNodeIndex", "primCount", "splitAxis"):
            assert (nodes[f] == ref[f]).all(), (i, f)
        leaf = nodes["primCount"] != 0xF
        assert (nodes["triVectorIndex"][leaf] == ref["triVectorIndex"][leaf]).all()
        assert np.array_equal(nodes["lower"], ref["lower"]) and np.array_equal(nodes["upper"], ref["upper"])
        assert (remap == z[f"remap{i}"]).all(), i
    for tris, seed in ((262000, 1), (2500000, 4)):
        scene, camera, _ = prt_amd.setup_atrium_standin(64, 64, tris=tris, seed=seed)
        m = scene.arrays()["meshes"][0]
        import time
        t0 = time.time()
        nodes, remap, ms = tracer.build_bvh(m["indices"], m["positions"])
        wall = time.time() - t0
        _assert_same_tree(nodes, remap, m["nodes"], m["remap"], f"atrium {tris}")
        print(f"GPU BVH build, {len(m['indices'])} triangles: {ms:.1f} ms on the device ({wall * 1e3:.0f} ms with transfers), {len(nodes)} nodes")


def test_rccl_gather_single_rank_communicator(tracer, c1):
    """prt_hip_gather_rccl through a real RCCL communicator of ONE rank (all a 1-GPU box allows: RCCL refuses two ranks on one
    device): librccl is found and loaded, ncclCommInitRank / grouped send-recv bookkeeping / stream ordering run, the
    root's image is left exactly as rendered, and the payload is the rank's share of the tile-padded image.  The >1-rank
    data movement (ownership, tile order, de-interleave) is covered by the same kernels in
    test_single_process_gather_of_several_contexts and by the gloo tests on host tensors."""
    scene, camera, desc = c1
    t = prt_amd.PathTracer(device=0, max_depth=4, seed=12345)
    try:
        t.upload_scene(scene)
        t.set_camera(camera)
        with pytest.raises(prt_amd.PrtError, match="no communicator"):
            t.gather_rccl()
        t.comm_init(prt_amd.comm_unique_id(), 0, 1)
        t.render_async(0, 0, 511, 299, 8)
        before = _download(t, np.zeros((512, 512, 3), dtype=np.float32))
        t.gather_rccl(root=0)
        after = _download(t, before)
        assert after.tobytes() == before.tobytes() and np.abs(after[:300]).sum() > 0
        assert t.gather_payload_bytes() == 32 * 32 * 256 * 12
        t.render_async(0, 0, 511, 299, 8, rank=1, nranks=2)
        with pytest.raises(prt_amd.PrtError, match="rank/nranks"):
            t.gather_rccl(root=0)  # rendered as rank 1 of 2, the communicator has one rank
        assert t.gather_payload_bytes() == 16 * 32 * 256 * 12
    finally:
        t.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_alpha_cell_classes_at_the_threshold_bytes(tracer, tmp_path, seed):
    """The cooperative leaf rounds decide most alpha tests from a 2-bit class per bilinear cell (prt_device.h DevScene::alphaClass:
    four texels >= 128 pass, four <= 126 fail, anything else takes Texture::testAlpha's blend, texture.cpp:31-100).  An alpha map
    made of blocks whose bytes sit ON the threshold -- 125, 126, 127, 128, 129 beside 0 and 255, plus blocks of per-texel noise
    from the same set -- so that cells of every class, and cells whose four texels are all 127 (the blend gives exactly 127: fails)
    or straddle 126 / 128, are hit by thousands of candidates; uv inside and far outside [0, 1]; scatter, packet-occlusion and
    single-occlusion rays (directional light).  Image and ray counts equal the oracle's bit for bit, in the timed build (the cell
    classes) and in the counting build (the serial leaf form: the blend every time)."""
    from test_host_cpu import _png
    rng = np.random.default_rng(9100 + seed)
    td = str(tmp_path)
    bw, bh = int(rng.integers(5, 9)), int(rng.integers(4, 8))
    w, h = 8 * bw + int(rng.integers(0, 5)), 8 * bh + int(rng.integers(0, 5))   # not multiples of the block size
    vals = np.array([0, 125, 126, 127, 128, 129, 255], dtype=np.uint8)
    a = rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
    for by in range(0, h, 8):
        for bx in range(0, w, 8):
            blk = a[by:by + 8, bx:bx + 8, 3]
            if rng.random() < 0.25:
                blk[...] = vals[rng.integers(0, len(vals), size=blk.shape)]     # noise on the threshold
            else:
                blk[...] = vals[int(rng.integers(0, len(vals)))]                # a flat block
    _png(os.path.join(td, "mask.png"), a, level=6)
    with open(os.path.join(td, "m.mtl"), "w") as f:
        f.write("newmtl holes\nKd 0.9 0.9 0.9\nmap_Kd mask.png\nnewmtl wall\nKd 0.7 0.6 0.5\n")
    n = 160
    centre = rng.uniform(-1, 1, size=(n, 1, 3))
    size = np.exp(rng.uniform(np.log(0.15), np.log(0.9), size=(n, 1, 1)))
    tri = centre + size * rng.normal(size=(n, 3, 3))
    uv = rng.uniform(-2.5, 3.5, size=(n, 3, 2)) if seed % 2 else rng.uniform(0, 1, size=(n, 3, 2))
    with open(os.path.join(td, "s.obj"), "w") as f:
        f.write("mtllib m.mtl\n")
        for t in tri.reshape(-1, 3):
            f.write("v %.9g %.9g %.9g\n" % tuple(t))
        for t in uv.reshape(-1, 2):
            f.write("vt %.9g %.9g\n" % tuple(t))
        for k in range(n):
            f.write("usemtl %s\n" % ("holes" if k % 4 else "wall"))
            i = 3 * k + 1
            f.write(f"f {i}/{i} {i + 1}/{i + 1} {i + 2}/{i + 2}\n")
    mesh = prt_amd.Mesh.load_obj(os.path.join(td, "s.obj"))
    mesh.calculate_bounds()
    scene = prt_amd.Scene()
    scene.add(mesh)
    assert int(scene.arrays()["meshes"][0]["materials"]["alphaTest"].sum()) == 1
    scene.set_directional_light((0.3, 0.5, 0.81), (6.0, 5.0, 4.0))
    W, H = 96, 64
    camera = prt_amd.Camera().create((0.1, -0.05, 3.4), (-0.02, 0.01, -1.0), W, H)
    upload(tracer, scene, camera)
    osc = T.OracleScene(T.scene_desc_from_product(scene, camera, 1.0))
    ref, ost = osc.render(32, max_depth=6)
    for count in (False, True):
        img = tracer.render(32, max_depth=6, count_traffic=count)
        st = tracer.last_stats
        assert_bits_equal(img, ref, "alpha map on the threshold, " + ("counting" if count else "timed") + " build")
        assert st["raysTraced"] == ost["raysTraced"] and st["occludedTraced"] == ost["occludedTraced"]


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 6, 9])
def test_textured_obj_scenes_through_the_asset_loaders(tracer, tmp_path, seed):
    """SURVEY 8f.2 on the GPU: a scene that reaches the kernels THROUGH THE ASSET LOADERS -- a triangle soup written as OBJ + MTL with
    PNG and TGA maps (random sizes 1..40 texels, not powers of two; RGB / RGBA-with-holes / grey diffuse maps, grey and RGB bump
    maps; texture coordinates far outside [0, 1] and negative), loaded by Mesh::loadObj (mesh.cpp:151-271: usemtl, map_Kd,
    map_bump, Ke, convertNormalToBump, isAlphaTestRequired), rendered by the frame kernel and by the oracle from the SAME loaded
    scene: images, ray counts and every event counter (texture taps included) in the counting build, the timed build's image,
    and the three G-buffer kinds.  (The decoders themselves are pinned by the CPU suite; what this adds is that loader output --
    texel layout, alpha flags, per-triangle materials, bump records -- is what the device scene expects.)"""
    from test_host_cpu import _png, _tga
    rng = np.random.default_rng(70000 + seed)
    td = str(tmp_path)

    def tex(name, ch):
        w, h = int(rng.integers(1, 41)), int(rng.integers(1, 41))
        a = rng.integers(0, 256, size=(h, w, ch)).astype(np.uint8)
        if ch == 4:
            a[..., 3] = np.where(rng.random((h, w)) < 0.4, rng.integers(0, 120, (h, w)), 255)
        img = a[..., 0] if ch == 1 else a
        if name.endswith(".tga"):
            _tga(os.path.join(td, name), img, rle=bool(seed & 1), top_down=bool(seed & 2))
        else:
            _png(os.path.join(td, name), img, level=(0, 1, 6, 9)[seed % 4])
    tex("rgb.png", 3)
    tex("rgba.tga" if seed % 3 == 0 else "rgba.png", 4)
    tex("grey.png", 1)
    tex("bumpg.tga" if seed % 2 else "bumpg.png", 1)
    tex("bumpn.png", 3)
    with open(os.path.join(td, "m.mtl"), "w") as f:
        f.write("newmtl a\nKd 0.9 0.8 0.7\nmap_Kd rgb.png\nmap_bump %s\n" % ("bumpg.tga" if seed % 2 else "bumpg.png"))
        f.write("newmtl b\nKd 1 1 1\nmap_Kd %s\n" % ("rgba.tga" if seed % 3 == 0 else "rgba.png"))
        f.write("newmtl c\nKd 0.6 0.9 0.6\nmap_Kd grey.png\nmap_bump bumpn.png\n")
        f.write("newmtl d\nKd 0.5 0.5 0.9\n")
        f.write("newmtl e\nKd 0.2 0.2 0.2\nKe 3 2 1\n")
    n = int(rng.integers(20, 200))
    centre = rng.uniform(-1, 1, size=(n, 1, 3))
    size = np.exp(rng.uniform(np.log(0.05), np.log(0.9), size=(n, 1, 1)))
    tri = centre + size * rng.normal(size=(n, 3, 3))
    uv = rng.uniform(-3, 4, size=(n, 3, 2)) if seed % 2 else rng.uniform(0, 1, size=(n, 3, 2))
    with open(os.path.join(td, "s.obj"), "w") as f:
        f.write("mtllib m.mtl\n")
        for t in tri.reshape(-1, 3):
            f.write("v %.9g %.9g %.9g\n" % tuple(t))
        for t in uv.reshape(-1, 2):
            f.write("vt %.9g %.9g\n" % tuple(t))
        for k in range(n):
            f.write("usemtl %s\n" % "abcde"[int(rng.integers(0, 5))])
            i = 3 * k + 1
            f.write(f"f {i}/{i} {i + 1}/{i + 1} {i + 2}/{i + 2}\n")
    mesh = prt_amd.Mesh.load_obj(os.path.join(td, "s.obj"))
    if seed % 3:
        mesh.calculate_vertex_normals()
    mesh.calculate_bounds()
    scene = prt_amd.Scene()
    scene.add(mesh)
    arrays = scene.arrays()
    assert len(arrays["textures"]) == 5 and int(arrays["meshes"][0]["materials"]["alphaTest"].sum()) >= 1  # the maps arrived, one with holes
    if seed % 4 != 0:
        d = rng.normal(size=3)
        d[2] = abs(d[2]) + 0.3
        d = d / np.linalg.norm(d)
        scene.set_directional_light(tuple(d.astype(np.float32)), tuple(rng.uniform(2, 12, 3)))
    w, h = int(rng.integers(24, 90)), int(rng.integers(16, 60))
    eye = rng.uniform(-1, 1, 3) * 0.3 + np.array([0, 0, 3.2])
    camera = prt_amd.Camera().create(tuple(eye), tuple(-eye + rng.normal(size=3) * 0.15), w, h)
    depth, spp = int(rng.choice([2, 6, 14])), int(rng.choice([8, 16]))
    upload(tracer, scene, camera)
    osc = T.OracleScene(T.scene_desc_from_product(scene, camera, 1.0))
    ref, ost = osc.render(spp, max_depth=depth)
    nan = np.isnan(ref)

    def same(img, what):
        assert np.array_equal(np.isnan(img), nan), what
        assert_bits_equal(img[~nan], ref[~nan], what)
    same(tracer.render(spp, max_depth=depth, count_traffic=True), "counting build")
    st = tracer.last_stats
    assert ost["nTap"] > 0
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    same(tracer.render(spp, max_depth=depth), "timed build")
    assert tracer.last_stats["raysTraced"] == ost["raysTraced"]
    for kind in (0, 1, 2):
        g, gr = tracer.gbuffer(kind), osc.gbuffer(kind, (0, 0, w - 1, h - 1))
        gn = np.isnan(gr)
        assert np.array_equal(np.isnan(g), gn)
        assert_bits_equal(g[~gn], gr[~gn], f"G-buffer kind {kind}")


def test_rccl_gather_two_and_three_ranks_through_the_standin():
    """The > 1-rank branch of prt_hip_gather_rccl (prt_gather.hip: pack on the owners, the root's staging offset table, grouped
    ncclSend / ncclRecv, one de-interleave per peer) EXECUTED on this one-GPU box: PRT_RCCL_LIB points the product at
    tests/fake_rccl.cpp, a stand-in whose ranks are threads of one process with a context each and whose send/recv rendezvous
    and move the bytes with hipMemcpyAsync, refusing a receive whose size differs from its send.  2 and 3 ranks, ragged images,
    roots 0 / 1 / 2, more ranks than tiles, tile sizes 8 / 16 / 32, each gather twice: the root's image must equal the one-GPU
    image bit for bit and the bytes moved must be exactly the peers' tiles.  Then a failing ncclSend and a failing ncclRecv
    inside the group: the error is reported, the group is closed, the communicator still works.  (The real RCCL is exercised by
    test_rccl_gather_single_rank_communicator; a real multi-GPU run is the driver's SCALE step.)"""
    import subprocess
    import sys
    fake = T.build_fake_rccl()
    env = dict(os.environ, PRT_RCCL_LIB=fake)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_threads_child.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK 6" in out.stdout, out.stdout + out.stderr


def test_full_size_c3_properties(tracer):
    """BASELINE config 3 as bench.py runs it (Sponza-class stand-in, 262 k triangles, 1920x1080, 64 spp, depth 8), checked
    through size-independent properties: repeatable, independent of how the image is cut into launches and ranks
    (two pipelines, rank interleave, ragged bands), and equal to the oracle on tiles sampled across the frame."""
    W, H, spp, depth = 1920, 1080, 64, 8
    scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
    upload(tracer, scene, camera)
    a = tracer.render(spp, max_depth=depth, exposure=exposure)
    sa = tracer.last_stats
    b = tracer.render(spp, max_depth=depth, exposure=exposure)
    assert a.tobytes() == b.tobytes() and tracer.last_stats["raysTraced"] == sa["raysTraced"]
    assert sa["nPx"] == W * H and sa["raysTraced"] >= spp * W * H and sa["occludedTraced"] > 0
    assert np.isfinite(a).all() and (a >= 0).all()
    assert_frame_equals_oracle_digests(a, sa, "frame_digests_c3.npz", spp, depth, exposure)  # the whole frame == the oracle's frame
    # ranks 0..2 of 3 into one image: every tile has exactly one owner
    union = np.zeros_like(a)
    rays = 0
    for r in range(3):
        tracer.render_async(0, 0, W - 1, H - 1, spp, max_depth=depth, exposure=exposure, rank=r, nranks=3)
        part = _download(tracer, a)
        mask = prt_amd.owned_pixel_mask(W, H, r, 3)
        union[mask] = part[mask]
        rays += tracer.stats()["raysTraced"]
    assert union.tobytes() == a.tobytes() and rays == sa["raysTraced"]
    # ragged bands (not multiples of the tile size)
    bands = [tracer.trace_block(0, y0, W - 1, y1, spp, max_depth=depth, exposure=exposure) for (y0, y1) in ((0, 406), (407, 1079))]
    assert np.concatenate(bands, 0).tobytes() == a.tobytes()
    desc = T.scene_desc_from_product(scene, camera, exposure)
    s = T.OracleScene(desc)
    for (x0, y0) in ((952, 532), (64, 1000), (1800, 40), (1904, 1064)):
        ref, _ = s.trace_block(x0, y0, x0 + 15, y0 + 15, spp, max_depth=depth)
        assert_bits_equal(a[y0:y0 + 16, x0:x0 + 16], ref, f"tile at {(x0, y0)}")


def _download(tracer, like):
    out = np.zeros_like(like)
    H, W, _ = like.shape
    tracer._chk(tracer._L.prt_hip_download(tracer._ctx, out.ctypes.data_as(prt_amd.C.c_void_p), 0, 0, W - 1, H - 1), "download")
    return out


def test_full_size_c4_workload(tracer):
    """BASELINE config 4 AT ITS WORKLOAD: San-Miguel-class stand-in (2.5 M triangles, alpha-masked cards, bump map, directional
    light; light and camera set-up as main.cpp:75-91), 1920x1080, 256 spp, depth cap 14 (the reference's literal), whole
    frame on one GPU.  Size-independent properties: repeatable; the union of the 8 ranks' shares -- what the 8 GPUs of the
    config render -- is the single-GPU image with the same ray total; tiles across the frame equal the oracle at 256 spp."""
    W, H, spp, depth = 1920, 1080, 256, 14
    scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=2500000, seed=4)
    upload(tracer, scene, camera)
    a = tracer.render(spp, max_depth=depth, exposure=exposure)
    sa = tracer.last_stats
    assert sa["nPx"] == W * H and sa["stackOverflow"] == 0 and sa["raysTraced"] >= spp * W * H and sa["occludedTraced"] > 0
    assert np.isfinite(a).all() and (a >= 0).all()
    print(f"C4 frame: {sa['raysTraced'] / 1e9:.2f} G rays in {sa['kernelMs']:.0f} ms = {sa['raysTraced'] / sa['kernelMs'] / 1e3:.0f} Mray/s")
    assert_frame_equals_oracle_digests(a, sa, "frame_digests_c4.npz", spp, depth, exposure)  # the whole frame == the oracle's frame
    b = tracer.render(spp, max_depth=depth, exposure=exposure)
    assert a.tobytes() == b.tobytes() and tracer.last_stats["raysTraced"] == sa["raysTraced"]
    union = np.zeros_like(a)
    rays = occl = px = 0
    for r in range(8):
        tracer.render_async(0, 0, W - 1, H - 1, spp, max_depth=depth, exposure=exposure, rank=r, nranks=8)
        part = _download(tracer, a)
        mask = prt_amd.owned_pixel_mask(W, H, r, 8)
        union[mask] = part[mask]
        st = tracer.stats()
        rays, occl, px = rays + st["raysTraced"], occl + st["occludedTraced"], px + st["nPx"]
        assert st["nPx"] == int(mask.sum())
    assert union.tobytes() == a.tobytes()
    assert (rays, occl, px) == (sa["raysTraced"], sa["occludedTraced"], W * H)
    s = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure))
    for (x0, y0) in ((944, 528), (48, 1008), (1808, 32), (1904, 1064)):
        ref, _ = s.render_rect((x0, y0, x0 + 15, y0 + 15), spp, max_depth=depth, stats=False)
        assert_bits_equal(a[y0:y0 + 16, x0:x0 + 16], ref, f"C4 tile at {(x0, y0)}")


def test_full_size_c5_workload_one_rank_of_eight(tracer):
    """BASELINE config 5 AT ITS WORKLOAD, as ONE of its 8 GPUs runs it: Zero-Day-class stand-in (5 M triangles, 10 % of them
    emissive, no directional light, exposure 64; main.cpp:93-105), 3840x2160, 1024 spp, depth cap 12, rank 3 of 8 (every 8th
    16x16 tile).  Checked: exactly the rank's pixels are written, the closed form of the primary ray count, and tiles of
    the rank equal the oracle at 1024 spp.  (The whole 4K frame at 8 spp: test_large_scene_4k_one_launch_matches_oracle_tiles.)"""
    W, H, spp, depth, rank, nranks = 3840, 2160, 1024, 12, 3, 8
    scene, camera, _ = prt_amd.setup_atrium_standin(W, H, tris=5000000, seed=5, emissive_fraction=0.1, light=False)
    exposure = 64.0
    upload(tracer, scene, camera)
    tracer.render_async(0, 0, W - 1, H - 1, spp, max_depth=depth, exposure=exposure, rank=rank, nranks=nranks)
    img = _download(tracer, np.zeros((H, W, 3), dtype=np.float32))
    st = tracer.stats()
    mask = prt_amd.owned_pixel_mask(W, H, rank, nranks)
    assert st["nPx"] == int(mask.sum()) and st["stackOverflow"] == 0 and st["occludedTraced"] == 0
    assert st["raysTraced"] >= spp * int(mask.sum())
    assert np.isfinite(img[mask]).all() and (img[mask] >= 0).all()
    print(f"C5 share: {st['raysTraced'] / 1e9:.2f} G rays in {st['kernelMs']:.0f} ms = {st['raysTraced'] / st['kernelMs'] / 1e3:.0f} Mray/s")
    tiles_x = W // 16
    s = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure))
    picked = []
    for (tx, ty) in ((119, 67), (3, 2), (236, 133), (40, 100)):  # the nearest tile this rank owns
        t = ty * tiles_x + tx
        t += (rank - t % nranks) % nranks
        picked.append((t % tiles_x * 16, t // tiles_x * 16))
    for (x0, y0) in picked:
        assert mask[y0, x0]
        ref, _ = s.render_rect((x0, y0, x0 + 15, y0 + 15), spp, max_depth=depth, stats=False)
        assert_bits_equal(img[y0:y0 + 16, x0:x0 + 16], ref, f"C5 tile at {(x0, y0)}")
    # Full-width tile rows of the WHOLE frame against the oracle's digests (tests/golden/c5_tile_rows.npz: SHA-256 of every 16x16
    # tile of the rows the oracle rendered at 1024 spp -- hours of CPU, done once by tools/c5_split_check.py in the build container).
    import hashlib
    z = np.load(os.path.join(G, "c5_tile_rows.npz"))
    assert (int(z["width"]), int(z["height"]), int(z["spp"]), int(z["max_depth"]), float(z["exposure"])) == (W, H, spp, depth, exposure)
    rows = sorted(int(r) for r in z["rows"])
    at = {r: q for q, r in enumerate(int(r) for r in z["rows"])}
    runs = []  # contiguous runs of tile rows, each rendered as one band
    for r in rows:
        if runs and runs[-1][1] == r - 1:
            runs[-1][1] = r
        else:
            runs.append([r, r])
    differing, rays, ms = [], 0, 0.0
    for r0, r1 in runs:
        y0, y1 = r0 * 16, r1 * 16 + 15
        tracer.render_async(0, y0, W - 1, y1, spp, max_depth=depth, exposure=exposure)
        band = _download(tracer, np.zeros((H, W, 3), dtype=np.float32))
        st = tracer.stats()
        assert st["nPx"] == (y1 - y0 + 1) * W and st["stackOverflow"] == 0
        rays += st["raysTraced"]
        ms += st["kernelMs"]
        for r in range(r0, r1 + 1):
            for c in range(tiles_x):
                tile = np.ascontiguousarray(band[r * 16:r * 16 + 16, c * 16:c * 16 + 16]).view(np.uint32).tobytes()
                if hashlib.sha256(tile).digest() != z["sha"][at[r], c].tobytes():
                    differing.append((c, r))
    assert not differing, f"{len(differing)} of {len(rows) * tiles_x} tiles of the C5 frame differ from the oracle's digests, first {differing[:4]}"
    y0, y1 = rows[0] * 16, rows[-1] * 16 + 15
    print(f"C5 tile rows {[tuple(r) for r in runs]}: {len(rows) * tiles_x} tiles ({100.0 * len(rows) * 16 / H:.1f} % of the frame) equal the oracle's digests; {rays / 1e9:.2f} G rays in {ms:.0f} ms")


@pytest.mark.parametrize("spp,max_depth,seed,tile", [(24, 14, 12345, 16), (8, 1, 777, 16), (4, 14, 12345, 16), (16, 2, 1, 8), (16, 6, 99, 32),
                                                      (40, 0, 5, 16)])
def test_parameter_corners_match_oracle(tracer, spp, max_depth, seed, tile):
    """Sample counts that are not a multiple of 8 (the divisor stays `samples`, path_tracer.cpp:28,65; fewer than 8 renders
    black), depth caps 0/1/2, other seeds, other tile sizes (a launch-partition parameter: it must not change a pixel)."""
    scene, camera, exposure = prt_amd.setup_bunny_standin(96, 80, tris=5000)
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    t = prt_amd.PathTracer(device=0, max_depth=max_depth, seed=seed)
    try:
        t.upload_scene(scene)
        t.set_camera(camera)
        rgb = t.render(spp, exposure=exposure, tile=tile, count_traffic=True)
        st = t.last_stats
    finally:
        t.close()
    ref, ost = T.OracleScene(desc).render(spp, max_depth=max_depth, seed=seed)
    assert_bits_equal(rgb, ref, f"spp {spp} depth {max_depth} seed {seed} tile {tile}")
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])


@pytest.mark.parametrize("rr_depth", [0, 2, 9])
def test_russian_roulette_depth_parameter(tracer, rr_depth):
    """prt_render_params.rrDepth (4 in the reference, path_tracer.cpp:258): the roulette draw changes the generator stream
    of every later bounce, so a wrong prefix count shows at once."""
    scene, camera, exposure = prt_amd.setup_bunny_standin(64, 64, tris=3000)
    desc = T.scene_desc_from_product(scene, camera, exposure)
    t = prt_amd.PathTracer(device=0, max_depth=12, rr_depth=rr_depth, seed=4242)
    L = T.oracle()
    try:
        t.upload_scene(scene)
        t.set_camera(camera)
        rgb = t.render(16, exposure=exposure)
        st = t.last_stats
        L.orc_set_rr_depth(rr_depth)
        ref, ost = T.OracleScene(desc).render(16, max_depth=12, seed=4242)
    finally:
        L.orc_set_rr_depth(4)
        t.close()
    assert_bits_equal(rgb, ref, f"rrDepth {rr_depth}")
    assert st["raysTraced"] == ost["raysTraced"] and st["occludedTraced"] == ost["occludedTraced"]


def test_cpp_drop_in_driver_matches_python_host(tracer, tmp_path):
    """examples/main.cpp -- the reference's main.cpp call sequence compiled against prt_amd/csrc/host/prt.h (the drop-in
    C++ surface: Scene/Camera/Mesh/Bvh/Image/PathTracer) -- renders the bunny-class scene; its float image must equal
    the one the ctypes host produces through the C-ABI, bit for bit."""
    import subprocess
    ex = os.path.join(T.ROOT, "examples")
    subprocess.check_call(["make", "-s", "-C", ex])
    W, H, spp = 160, 120, 16
    out = subprocess.run([os.path.join(ex, "prt_main"), "bunny", str(W), str(H), str(spp)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    with open(tmp_path / "render.pfm", "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0  # little-endian
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)[::-1]  # PFM rows run bottom to top
    scene, camera, exposure = prt_amd.setup_bunny_standin(W, H, tris=69451)
    upload(tracer, scene, camera)
    ref = tracer.render(spp, exposure=exposure)
    assert (w, h) == (W, H)
    assert_bits_equal(np.ascontiguousarray(img), ref, "C++ driver vs Python host")
    assert f"{tracer.last_stats['raysTraced']} rays" in out.stdout
    # the reference's own calling pattern: one TraceBlock per 16x16 tile from a thread pool, a PathTracer per task
    # (main.cpp:132-160).  The second tile renders the whole frame once; all tiles come out of it; totals are the frame's.
    out2 = subprocess.run([os.path.join(ex, "prt_main"), "bunny", str(W), str(H), str(spp), "tiles"], cwd=tmp_path, capture_output=True, text=True,
                          timeout=300)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    with open(tmp_path / "render.pfm", "rb") as f:
        f.readline(); f.readline(); f.readline()
        img2 = np.frombuffer(f.read(), dtype="<f4").reshape(H, W, 3)[::-1]
    assert_bits_equal(np.ascontiguousarray(img2), ref, "C++ driver, per-tile calls")
    assert f"{tracer.last_stats['raysTraced']} rays" in out2.stdout


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0  # little-endian
        return np.ascontiguousarray(np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)[::-1])  # rows run bottom to top


def test_cpp_driver_two_scenes_from_one_stack_slot(tracer, tmp_path):
    """The reference's main() calls raytrace_scene() for one scene after another, each with a stack `Scene scene;`
    (main.cpp:107-118, 192-200): the second Scene lives at the first one's address and here has the SAME camera.  The
    host's device-side cache is keyed on (address, process-wide revision) and dropped by ~Scene, so the second call
    must upload and render the second scene -- not serve the first one's BVH or its kept frame."""
    import subprocess
    ex = os.path.join(T.ROOT, "examples")
    subprocess.check_call(["make", "-s", "-C", ex])
    W, H, spp = 96, 80, 8
    out = subprocess.run([os.path.join(ex, "prt_main"), "twoscenes", str(W), str(H), str(spp)], cwd=tmp_path, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    a, b = read_pfm(tmp_path / "render_a.pfm"), read_pfm(tmp_path / "render_b.pfm")
    scene, camera, exposure = prt_amd.setup_bunny_standin(W, H, tris=69451)
    upload(tracer, scene, camera)
    assert_bits_equal(a, tracer.render(spp, exposure=exposure), "first scene (bunny stand-in)")
    scene, camera, exposure = prt_amd.setup_cornell_box(W, H)
    upload(tracer, scene, camera)
    assert_bits_equal(b, tracer.render(spp, exposure=exposure), "second scene (Cornell box) from the same stack slot")
    assert a.tobytes() != b.tobytes()


def test_cpp_multi_process_example_one_rank(tmp_path):
    """examples/prt_mp.cpp -- a C++ host with one process per GPU: forked ranks, the communicator id over pipes,
    prt_hip_comm_init / prt_hip_render(rank, nranks) / prt_hip_gather_rccl, and rank 0 requiring the gathered image to equal
    its own one-GPU render.  This box has one GPU and RCCL takes one rank per device, so the run is the 1-rank case (a real
    communicator, the library's pack / de-interleave path); asking for 2 ranks must be refused with a message, not hang."""
    import subprocess
    ex = os.path.join(T.ROOT, "examples")
    subprocess.check_call(["make", "-s", "-C", ex])
    out = subprocess.run([os.path.join(ex, "prt_mp"), "1", "cornell", "160", "96", "16"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "EQUALS the one-GPU image" in out.stdout, out.stdout
    assert (tmp_path / "prt_mp.exr").stat().st_size > 0
    if prt_amd.device_count() < 2:
        out = subprocess.run([os.path.join(ex, "prt_mp"), "2", "cornell", "64", "64", "8"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert out.returncode == 3 and "need 2 devices" in out.stderr, out.stdout + out.stderr


def test_gbuffer_visualizer_matches_reference_and_oracle(tracer, c1):
    """GbufferVisualizer (gbuffer_visualizer.cpp:17-51) on the traversal kernels' wave loop: Cornell + teapot
    against the compiled reference's images (tests/golden/gbuffer.npz), the textured atrium against the oracle."""
    z = np.load(os.path.join(G, "gbuffer.npz"))
    scene, camera, exposure = prt_amd.setup_cornell_box(96, 96, teapot_mesh=T.teapot_product_mesh())
    upload(tracer, scene, camera)
    x0, y0, x1, y1 = (int(v) for v in z["rect"])
    for k in (0, 1, 2):
        assert_bits_equal(tracer.gbuffer(k, x0, y0, x1, y1), z[f"kind{k}"], f"gbuffer kind {k} vs compiled reference")
    scene, camera, exposure = prt_amd.setup_atrium_standin(192, 108, tris=40000)
    upload(tracer, scene, camera)
    s = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure))
    for k in (0, 1):
        assert_bits_equal(tracer.gbuffer(k, exposure=exposure), s.gbuffer(k, (0, 0, 191, 107)), f"atrium gbuffer kind {k}")
    with pytest.raises(prt_amd.PrtError):
        tracer.gbuffer(3)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_soups_match_oracle(tracer, seed):
    """Seeded random scenes: triangle soups of mixed sizes (slivers, large overlapping triangles, duplicates), random diffuse /
    specular / emissive materials over two meshes, random camera, directional light on or off.  Image, ray counts and event
    counters equal the oracle's; NaN pixels (degenerate geometry can produce them in the reference too) must be NaN in both."""
    rng = np.random.default_rng(1000 + seed)
    scene = prt_amd.Scene()
    for m in range(2):
        n = int(rng.integers(40, 400))
        centre = rng.uniform(-1, 1, size=(n, 1, 3))
        size = np.exp(rng.uniform(np.log(0.01), np.log(0.8), size=(n, 1, 1)))
        tri = (centre + size * rng.normal(size=(n, 3, 3))).astype(np.float32)
        if seed % 2 == 0:
            tri[:n // 20] = tri[n // 20:2 * (n // 20)]            # exact duplicates: ties at equal t
            tri[-3:, 2] = tri[-3:, 1]                              # degenerate (zero-area) triangles
        pos = tri.reshape(-1, 3)
        idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
        kinds = rng.integers(0, 3, size=6)
        mats = np.array([T.make_material(diffuse=tuple(rng.uniform(0.2, 0.9, 3)), reflection=int(k == 1),
                                         emissive=tuple(rng.uniform(1, 6, 3)) if k == 2 else (0, 0, 0)) for k in kinds], dtype=T.MATERIAL_DTYPE)
        pm = prt_amd.Mesh.from_arrays(idx, pos, rng.integers(0, 6, size=n).astype(np.uint32), mats.view(prt_amd.MATERIAL_DTYPE))
        if m == 1:
            pm.calculate_vertex_normals()
        pm.calculate_bounds()
        scene.add(pm)
    if seed % 3 != 0:
        scene.set_directional_light(_unit(rng.normal(size=3)), tuple(rng.uniform(1, 8, 3)))
    eye = rng.uniform(-1, 1, 3) * 0.3 + np.array([0, 0, 3.0])
    camera = prt_amd.Camera().create(tuple(eye), tuple(-eye + rng.normal(size=3) * 0.2), 56, 40)
    upload(tracer, scene, camera)
    desc = T.scene_desc_from_product(scene, camera, 1.0)
    rgb = tracer.render(16, max_depth=6, count_traffic=True)
    st = tracer.last_stats
    ref, ost = T.OracleScene(desc).render(16, max_depth=6)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(rgb), nan)
    assert_bits_equal(rgb[~nan], ref[~nan], f"random soup {seed}")
    for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    timed = tracer.render(16, max_depth=6)
    assert np.array_equal(np.isnan(timed), nan)
    assert_bits_equal(timed[~nan], ref[~nan], f"random soup {seed}, timed build")


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return tuple((v / np.linalg.norm(v)).astype(np.float32))


def test_c1_full_image_digest_matches_compiled_reference(tracer, c1):
    """BASELINE config 1 whole (512x512, 16 spp, depth literal 14): SHA-256 of the float image, ray count and row sums equal those
    of the compiled reference's render (tests/golden/c1_full_checksums.npz; 18 080 026 rays)."""
    import hashlib
    scene, camera, desc = c1
    upload(tracer, scene, camera)
    z = np.load(os.path.join(G, "c1_full_checksums.npz"))
    rgb = tracer.render(16, max_depth=14)
    assert tracer.last_stats["raysTraced"] == int(z["rays"][0]) == 18080026 and tracer.last_stats["occludedTraced"] == 0
    assert np.array_equal(rgb.astype(np.float64).sum(axis=(1, 2)), z["row_sums"])
    digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(rgb, dtype="<f4").tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(digest, z["sha256"])


@pytest.mark.parametrize("name,setup,kw,w,h", [("c2", "setup_bunny_standin", dict(tris=20000), 192, 192),
                                               ("c3", "setup_atrium_standin", dict(tris=40000), 192, 108)])
def test_scene_digests_match_compiled_reference(tracer, name, setup, kw, w, h):
    """Whole images of a C2-class scene (directional light: packet and single occlusion rays) and a C3-class scene (alpha masks,
    bump map, vertex normals) at the reference's depth literal 14: digest, ray counts and row sums of the COMPILED REFERENCE's
    render of the same scene (tests/golden/scene_digests.npz; its surface and texture members come from the oracle)."""
    import hashlib
    z = np.load(os.path.join(G, "scene_digests.npz"))
    scene, camera, exposure = getattr(prt_amd, setup)(w, h, **kw)
    upload(tracer, scene, camera)
    rgb = tracer.render(16, max_depth=14, exposure=exposure)
    st = tracer.last_stats
    assert st["raysTraced"] == int(z[f"{name}_rays"][0]) and st["occludedTraced"] == int(z[f"{name}_rays"][1])
    assert np.array_equal(rgb.astype(np.float64).sum(axis=(1, 2)), z[f"{name}_row_sums"])
    digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(rgb, dtype="<f4").tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(digest, z[f"{name}_sha256"])
