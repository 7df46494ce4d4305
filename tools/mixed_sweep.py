"""A third parity sweep (GPU against the oracle, bit for bit, counters included): textured atrium + up to 4 more meshes (random soups,
a displaced sphere) = up to 6 BVHs in one scene, at a random global scale (1e-3 .. 1e3), lit by a directional light, an environment
map or nothing, camera anywhere around; path tracer and the three G-buffer kinds.  usage: mixed_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
T.oracle().orc_set_anyhit_accounting(1)
tr = prt_amd.PathTracer(device=0)
bad = 0


def soup(rng, n, extent):
    centre = rng.uniform(-1, 1, size=(n, 1, 3)) * extent
    size = np.exp(rng.uniform(np.log(0.01), np.log(0.5), size=(n, 1, 1))) * extent
    pos = (centre + size * rng.normal(size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
    kinds = rng.integers(0, 3, size=3)
    mats = np.array([T.make_material(diffuse=tuple(rng.uniform(0.2, 0.9, 3)), reflection=int(k == 1),
                                     emissive=tuple(rng.uniform(1, 6, 3)) if k == 2 else (0, 0, 0)) for k in kinds], dtype=T.MATERIAL_DTYPE)
    m = prt_amd.Mesh.from_arrays(idx, pos, rng.integers(0, 3, size=n).astype(np.uint32), mats.view(prt_amd.MATERIAL_DTYPE))
    if rng.integers(0, 2):
        m.calculate_vertex_normals()
    m.calculate_bounds()
    return m


for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(9000 + seed)
    scale = float(rng.choice([1.0, 1.0, 1e-3, 37.5, 1e3]))
    scene = prt_amd.Scene()
    a = prt_amd.Mesh.atrium(int(rng.integers(2000, 30000)), seed, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), float(rng.choice([0.0, 0.2])))
    a.transform(scale, (0.0, 0.0, 0.0))
    a.calculate_vertex_normals()
    a.calculate_bounds()
    scene.add(a)
    extra = int(rng.integers(0, 5))
    for k in range(extra):
        m = soup(rng, int(rng.integers(10, 200)), 4.0 * scale)
        scene.add(m)
    if rng.integers(0, 3) == 0:
        mat = prt_amd.Material.make(diffuse=(0.8, 0.7, 0.6), reflection=int(rng.integers(0, 2)))
        s = prt_amd.Mesh.displaced_sphere(int(rng.integers(500, 5000)), 1.5 * scale, (float(-10 * scale), float(3 * scale), 0.0), mat, seed)
        s.calculate_vertex_normals()
        s.calculate_bounds()
        scene.add(s)
    light = int(rng.integers(0, 3))
    if light == 0:
        d = rng.normal(size=3); d[1] = abs(d[1]) + 0.2; d = d / np.linalg.norm(d)
        scene.set_directional_light(tuple(d.astype(np.float32)), tuple(rng.uniform(1, 20, 3)))
    elif light == 1:
        ew, eh = int(rng.integers(2, 32)), int(rng.integers(2, 16))
        env = np.ones((eh, ew, 4), dtype=np.float32)
        env[..., :3] = rng.random((eh, ew, 3), dtype=np.float32) * 3.0
        if seed % 4 == 0:
            env[0, 0, :3] = 800.0
        scene.set_infinite_area_light(env)
    w, h = int(rng.integers(24, 120)), int(rng.integers(16, 80))
    eye = np.array([-15.0, 4.0, 0.5]) + rng.normal(size=3) * np.array([6.0, 2.0, 3.0])
    look = np.array([1.0, 0.08, -0.05]) + rng.normal(size=3) * 0.4
    camera = prt_amd.Camera().create(tuple((eye * scale).astype(np.float32)), tuple(look.astype(np.float32)), w, h)
    depth, spp, exposure = int(rng.choice([1, 3, 8, 14])), int(rng.choice([8, 16])), float(rng.choice([1.0, 8.0]))
    tr.max_depth = depth
    tr.upload_scene(scene); tr.set_camera(camera)
    rgb = np.asarray(tr.render(spp, exposure=exposure, count_traffic=True))
    st = tr.last_stats
    osc = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure))
    ref, ost = osc.render(spp, max_depth=depth)
    ref = np.asarray(ref)
    nan = np.isnan(ref)
    ok = np.array_equal(np.isnan(rgb), nan) and np.array_equal(rgb[~nan].view(np.uint32), ref[~nan].view(np.uint32)) and \
        all(st[k] == ost[k] for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"))
    timed = np.asarray(tr.render(spp, exposure=exposure))
    ok = ok and np.array_equal(np.isnan(timed), nan) and np.array_equal(timed[~nan].view(np.uint32), ref[~nan].view(np.uint32))
    gok = True
    for kind in (0, 1, 2):
        g = np.asarray(tr.gbuffer(kind, exposure=exposure))
        gr = np.asarray(osc.gbuffer(kind, (0, 0, w - 1, h - 1)))
        gn = np.isnan(gr)
        gok = gok and np.array_equal(np.isnan(g), gn) and np.array_equal(g[~gn].view(np.uint32), gr[~gn].view(np.uint32))
    tag = (w, h, 1 + extra, scale, ("dir", "env", "none")[light], depth, spp)
    if not (ok and gok):
        bad += 1
        print("seed", seed, tag, "MISMATCH path" if not ok else "", "MISMATCH gbuffer" if not gok else "",
              {k: (st[k], ost[k]) for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap")}, flush=True)
    elif seed % 5 == 0:
        print("seed", seed, tag, "ok", st["raysTraced"], "rays", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
