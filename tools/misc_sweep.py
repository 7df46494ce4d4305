"""Extended sweeps of two more parity tests (GPU against the oracle / against itself, bit for bit):
  env:   random soups lit by random float environment maps (black rows and columns, bright spots), counters included;
  ranks: random image sizes, rectangles, tile sizes and rank counts: every rank writes exactly its tiles' pixels inside the
         rectangle and the union of the ranks' pixels is the one-GPU image.
usage: misc_sweep.py FIRST LAST"""
import os, sys
import ctypes as C
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
T.oracle().orc_set_anyhit_accounting(1)
tr = prt_amd.PathTracer(device=0)
bad = 0


def soup(rng, n):
    centre = rng.uniform(-1, 1, size=(n, 1, 3))
    size = np.exp(rng.uniform(np.log(0.02), np.log(0.7), size=(n, 1, 1)))
    pos = (centre + size * rng.normal(size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
    kinds = rng.integers(0, 3, size=4)
    mats = np.array([T.make_material(diffuse=tuple(rng.uniform(0.2, 0.9, 3)), reflection=int(k == 1), emissive=(0, 0, 0)) for k in kinds], dtype=T.MATERIAL_DTYPE)
    m = prt_amd.Mesh.from_arrays(idx, pos, rng.integers(0, 4, size=n).astype(np.uint32), mats.view(prt_amd.MATERIAL_DTYPE))
    m.calculate_vertex_normals()
    m.calculate_bounds()
    return m


for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(7000 + seed)
    # ---- environment light
    ew, eh = int(rng.integers(2, 40)), int(rng.integers(2, 24))
    env = np.zeros((eh, ew, 4), dtype=np.float32)
    env[..., :3] = rng.random((eh, ew, 3), dtype=np.float32) * rng.choice([0.5, 4.0, 50.0])
    env[..., 3] = 1.0
    if seed % 2:
        env[rng.integers(0, eh), :, :3] = 0.0
        env[:, rng.integers(0, ew), :3] = 0.0
    if seed % 5 == 0:
        env[rng.integers(0, eh), rng.integers(0, ew), :3] = 5000.0
    scene = prt_amd.Scene()
    scene.add(soup(rng, int(rng.integers(30, 300))))
    scene.set_infinite_area_light(env)
    w, h = int(rng.integers(17, 80)), int(rng.integers(9, 50))
    eye = rng.uniform(-1, 1, 3) * 0.3 + np.array([0, 0, 3.0])
    camera = prt_amd.Camera().create(tuple(eye), tuple(-eye + rng.normal(size=3) * 0.2), w, h)
    depth, spp = int(rng.choice([3, 6, 14])), int(rng.choice([8, 16]))
    tr.max_depth = depth
    tr.upload_scene(scene); tr.set_camera(camera)
    rgb = np.asarray(tr.render(spp, count_traffic=True))
    st = tr.last_stats
    ref, ost = T.OracleScene(T.scene_desc_from_product(scene, camera, 1.0)).render(spp, max_depth=depth)
    nan = np.isnan(ref)
    ok = np.array_equal(np.isnan(rgb), nan) and np.array_equal(rgb[~nan].view(np.uint32), np.asarray(ref)[~nan].view(np.uint32)) and \
        all(st[k] == ost[k] for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"))
    if not ok:
        bad += 1
        same_nan = np.array_equal(np.isnan(rgb), nan)
        diff = (rgb.view(np.uint32) != np.asarray(ref).view(np.uint32)) & ~nan
        print("seed", seed, "image:", "nan pattern", same_nan, "differing values", int(diff.sum()), "nan pixels", int(nan.any(-1).sum()),
              "timed-build image equal:", np.array_equal(np.asarray(tr.render(spp)).view(np.uint32)[~nan], np.asarray(ref).view(np.uint32)[~nan]), flush=True)
        print("seed", seed, "ENV MISMATCH", (ew, eh, w, h, depth, spp), {k: (st[k], ost[k]) for k in ("raysTraced", "occludedTraced", "nBox", "nTri")}, flush=True)
    # ---- ranks and rectangles (same scene)
    full = np.asarray(tr.render(spp))
    tile = int(rng.choice([8, 16, 16, 32]))
    nranks = int(rng.integers(2, 9))
    x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
    x1, y1 = int(rng.integers(x0, w)), int(rng.integers(y0, h))
    acc = np.full((h, w, 3), -7.0, dtype=np.float32)
    okr = True
    for rank in range(nranks):
        tr.render_async(x0, y0, x1, y1, spp, rank=rank, nranks=nranks, tile=tile)
        img = np.full((h, w, 3), -7.0, dtype=np.float32)
        prt_amd._check(prt_amd.lib().prt_hip_download(tr._ctx, img.ctypes.data_as(C.c_void_p), x0, y0, x1, y1), "download")
        stt = tr.stats()
        ty, tx = np.meshgrid(np.arange(h) // tile, np.arange(w) // tile, indexing="ij")
        tilesx = (w + tile - 1) // tile
        own = ((ty * tilesx + tx) % nranks) == rank
        inside = np.zeros((h, w), bool)
        inside[y0:y1 + 1, x0:x1 + 1] = True
        own &= inside
        okr = okr and stt["nPx"] == int(own.sum())
        acc[own] = img[own]
    fn = np.isnan(full)
    okr = okr and np.array_equal(np.isnan(acc[inside]), fn[inside]) and np.array_equal(acc[inside & ~fn.any(-1)].view(np.uint32), full[inside & ~fn.any(-1)].view(np.uint32))
    if not okr:
        bad += 1
        print("seed", seed, "RANK MISMATCH", (w, h, tile, nranks, (x0, y0, x1, y1)), flush=True)
    if seed % 10 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
