"""Parity sweep over TEXTURES (rows a12 / a14 through the kernels' bilinear, alpha, diffuse and bump taps): triangle soups written as
OBJ + MTL with PNG maps of random sizes (1..40 texels a side, not powers of two), RGB / RGBA-with-holes / grey diffuse maps, grey and RGB
bump maps, texture coordinates far outside [0, 1] and negative; loaded through Mesh::loadObj, rendered on the GPU and by the oracle:
images, ray counts and all event counters (taps included), counting and timed build, plus the three G-buffer kinds.  Needs Pillow to
write the PNGs.  usage: texture_sweep.py FIRST LAST"""
import os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from PIL import Image
import prt_amd
import prt_testlib as T
T.oracle().orc_set_anyhit_accounting(1)
tr = prt_amd.PathTracer(device=0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(70000 + seed)
    with tempfile.TemporaryDirectory() as td:
        def tex(name, mode):
            w, h = int(rng.integers(1, 41)), int(rng.integers(1, 41))
            ch = {"L": 1, "RGB": 3, "RGBA": 4}[mode]
            a = rng.integers(0, 256, size=(h, w, ch)).astype(np.uint8)
            if mode == "RGBA":
                a[..., 3] = np.where(rng.random((h, w)) < 0.4, rng.integers(0, 120, (h, w)), 255)
            Image.fromarray(a[..., 0] if ch == 1 else a, mode).save(os.path.join(td, name))
        tex("rgb.png", "RGB"); tex("rgba.png", "RGBA"); tex("grey.png", "L"); tex("bumpg.png", "L"); tex("bumpn.png", "RGB")
        with open(os.path.join(td, "m.mtl"), "w") as f:
            f.write("newmtl a\nKd 0.9 0.8 0.7\nmap_Kd rgb.png\nmap_bump bumpg.png\n")
            f.write("newmtl b\nKd 1 1 1\nmap_Kd rgba.png\n")
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
    if seed % 4 != 0:
        d = rng.normal(size=3); d[2] = abs(d[2]) + 0.3; d = d / np.linalg.norm(d)
        scene.set_directional_light(tuple(d.astype(np.float32)), tuple(rng.uniform(2, 12, 3)))
    w, h = int(rng.integers(24, 90)), int(rng.integers(16, 60))
    eye = rng.uniform(-1, 1, 3) * 0.3 + np.array([0, 0, 3.2])
    camera = prt_amd.Camera().create(tuple(eye), tuple(-eye + rng.normal(size=3) * 0.15), w, h)
    depth, spp = int(rng.choice([2, 6, 14])), int(rng.choice([8, 16]))
    tr.max_depth = depth
    tr.upload_scene(scene); tr.set_camera(camera)
    rgb = np.asarray(tr.render(spp, count_traffic=True))
    st = tr.last_stats
    osc = T.OracleScene(T.scene_desc_from_product(scene, camera, 1.0))
    ref, ost = osc.render(spp, max_depth=depth)
    ref = np.asarray(ref)
    nan = np.isnan(ref)
    ok = np.array_equal(np.isnan(rgb), nan) and np.array_equal(rgb[~nan].view(np.uint32), ref[~nan].view(np.uint32)) and \
        all(st[k] == ost[k] for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"))
    timed = np.asarray(tr.render(spp))
    ok = ok and np.array_equal(np.isnan(timed), nan) and np.array_equal(timed[~nan].view(np.uint32), ref[~nan].view(np.uint32))
    for kind in (0, 1, 2):
        g = np.asarray(tr.gbuffer(kind))
        gr = np.asarray(osc.gbuffer(kind, (0, 0, w - 1, h - 1)))
        gn = np.isnan(gr)
        ok = ok and np.array_equal(np.isnan(g), gn) and np.array_equal(g[~gn].view(np.uint32), gr[~gn].view(np.uint32))
    if not ok:
        bad += 1
        print("seed", seed, (w, h, n, depth, spp), "MISMATCH", {k: (st[k], ost[k]) for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap")}, flush=True)
    elif seed % 10 == 0:
        print("seed", seed, (w, h, n, depth, spp), "ok", st["raysTraced"], "rays", ost["nTap"], "taps", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
