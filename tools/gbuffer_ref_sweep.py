"""The oracle's G-buffer integrator against the REFERENCE's compiled GbufferVisualizer (oracle/_ref/ref_path; build container only) on
random soup scenes of 1-3 meshes with and without vertex normals, the three kinds.  (Surface fetch and bump taps are the oracle's own on
both sides, oracle/ref_glue.cpp; the integrator, camera, RNG and traversal are the reference's.)  usage: gbuffer_ref_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
if T.ref_binary("ref_path") is None:
    sys.exit("oracle/_ref/ref_path is not built (needs /root/reference)")
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(90000 + seed)
    scene = prt_amd.Scene()
    for m in range(int(rng.integers(1, 4))):
        n = int(rng.integers(10, 300))
        centre = rng.uniform(-1, 1, size=(n, 1, 3))
        size = np.exp(rng.uniform(np.log(0.02), np.log(0.8), size=(n, 1, 1)))
        pos = (centre + size * rng.normal(size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
        idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
        mats = np.array([T.make_material(diffuse=tuple(rng.uniform(0.1, 1.0, 3))) for _ in range(3)], dtype=T.MATERIAL_DTYPE)
        mesh = prt_amd.Mesh.from_arrays(idx, pos, rng.integers(0, 3, size=n).astype(np.uint32), mats.view(prt_amd.MATERIAL_DTYPE))
        if rng.integers(0, 2):
            mesh.calculate_vertex_normals()
        mesh.calculate_bounds()
        scene.add(mesh)
    w, h = int(rng.integers(16, 80)), int(rng.integers(12, 60))
    eye = rng.uniform(-1, 1, 3) * 0.4 + np.array([0, 0, 3.0])
    camera = prt_amd.Camera().create(tuple(eye), tuple(-eye + rng.normal(size=3) * 0.2), w, h)
    exposure = float(rng.choice([1.0, 2.5]))
    desc = T.scene_desc_from_product(scene, camera, exposure)
    s = T.OracleScene(desc)
    for kind in (0, 1, 2):
        ref = T.ref_gbuffer(desc, kind, (0, 0, w - 1, h - 1))
        img = np.asarray(s.gbuffer(kind, (0, 0, w - 1, h - 1)))
        nan = np.isnan(ref)
        if not (np.array_equal(np.isnan(img), nan) and np.array_equal(img[~nan].view(np.uint32), ref[~nan].view(np.uint32))):
            bad += 1
            print("seed", seed, "kind", kind, (w, h), "MISMATCH", flush=True)
    if seed % 25 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
