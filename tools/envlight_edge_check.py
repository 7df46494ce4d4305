"""The cases of InfiniteAreaLight::sample the reference leaves undefined (its vertical scan runs off the table: one-row maps, a black
last row) are DEFINED in this build ("no column exceeds the draw"): the GPU must equal the oracle on them.  usage: envlight_edge_check.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
T.oracle().orc_set_anyhit_accounting(1)
tr = prt_amd.PathTracer(device=0, max_depth=6)
bad = 0
for seed in range(40):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 20)), int(rng.choice([1, 1, 2, 3, 5]))
    env = np.ones((h, w, 4), dtype=np.float32)
    env[..., :3] = rng.random((h, w, 3), dtype=np.float32) * 4
    if h > 1 and seed % 2:
        env[-1, :, :3] = 0.0  # black last row
    if seed % 5 == 0:
        env[0, 0, :3] = 2000.0
    scene, camera, _ = prt_amd.setup_cornell_box(48, 40)
    scene.set_infinite_area_light(env)
    tr.upload_scene(scene); tr.set_camera(camera)
    rgb = np.asarray(tr.render(16, count_traffic=True))
    st = tr.last_stats
    ref, ost = T.OracleScene(T.scene_desc_from_product(scene, camera, 1.0)).render(16, max_depth=6)
    ref = np.asarray(ref)
    nan = np.isnan(ref)
    ok = np.array_equal(np.isnan(rgb), nan) and np.array_equal(rgb[~nan].view(np.uint32), ref[~nan].view(np.uint32)) and \
        all(st[k] == ost[k] for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"))
    if not ok:
        bad += 1
        print("seed", seed, (w, h), "MISMATCH", flush=True)
print("edge check finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
