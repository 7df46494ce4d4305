"""Extended sweep of the textured-scene parity test (GPU against the oracle, bit for bit, event counters included): atrium stand-ins
of several seeds, sizes, triangle counts, depth caps and material mixes.  usage: atrium_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
import prt_testlib as T
T.oracle().orc_set_anyhit_accounting(1)  # the oracle counts occlusion queries the way the GPU visits them (and checks both visits agree)
tr = prt_amd.PathTracer(device=0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 120))
    tris = int(rng.integers(3000, 60000))
    kw = dict(tris=tris, seed=seed, alpha=bool(rng.integers(0, 2)), bump=bool(rng.integers(0, 2)),
              emissive_fraction=float(rng.choice([0.0, 0.0, 0.1, 0.5])), light=bool(rng.integers(0, 4) != 0))
    depth, spp, exposure = int(rng.choice([2, 4, 8, 14])), int(rng.choice([8, 16, 24])), float(rng.choice([1.0, 64.0]))
    scene, camera, _ = prt_amd.setup_atrium_standin(w, h, **kw)
    tr.max_depth = depth
    tr.upload_scene(scene); tr.set_camera(camera)
    rgb = tr.render(spp, exposure=exposure, count_traffic=True)
    st = tr.last_stats
    ref, ost = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure)).render(spp, max_depth=depth)
    ok = np.array_equal(np.asarray(rgb).view(np.uint32), np.asarray(ref).view(np.uint32)) and all(st[k] == ost[k] for k in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx"))
    timed = tr.render(spp, exposure=exposure)
    ok = ok and np.array_equal(np.asarray(timed).view(np.uint32), np.asarray(ref).view(np.uint32))
    if not ok:
        bad += 1
        print("seed", seed, kw, (w, h, depth, spp), "MISMATCH", {k: (st[k], ost[k]) for k in ("raysTraced", "nBox", "nTri", "nHit", "nTap")}, flush=True)
    else:
        print("seed", seed, (w, h, tris, depth, spp), "ok", st["raysTraced"], "rays", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
