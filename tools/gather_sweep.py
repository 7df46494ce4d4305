"""Sweep of the image gather (prt_hip_gather: the pack and de-interleave kernels the RCCL path shares, with device-to-device copies in
between): 2..6 contexts on one device, random image sizes (not multiples of the tile), tile sizes 8 / 16 / 32 / 64, full frames and
ragged rectangles; the assembled image must equal one context's render bit for bit, and the one-rank RCCL communicator path must leave
the frame unchanged.  usage: gather_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
bad = 0
pool = [prt_amd.PathTracer(device=0, max_depth=4, seed=12345) for _ in range(6)]
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(80000 + seed)
    w, h = int(rng.integers(9, 300)), int(rng.integers(9, 200))
    scene, camera, _ = prt_amd.setup_cornell_box(w, h)
    n = int(rng.integers(2, 7))
    tile = int(rng.choice([8, 16, 32, 64]))
    if seed % 3 == 0:
        rect = (0, 0, w - 1, h - 1)
    else:
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        rect = (x0, y0, int(rng.integers(x0, w)), int(rng.integers(y0, h)))
    spp = 8
    pool[0].upload_scene(scene); pool[0].set_camera(camera)
    whole = np.asarray(pool[0].trace_block(*rect, spp, tile=tile))
    for i in range(n):
        pool[i].upload_scene(scene); pool[i].set_camera(camera)
        pool[i].render_async(*rect, spp, rank=i, nranks=n, tile=tile)
    img = np.asarray(prt_amd.gather_contexts(pool[:n], *rect))
    x0, y0, x1, y1 = rect
    ok = np.array_equal(img[y0:y1 + 1, x0:x1 + 1].view(np.uint32), whole.view(np.uint32))
    if not ok:
        bad += 1
        print("seed", seed, (w, h, n, tile, rect), "MISMATCH", int((img[y0:y1 + 1, x0:x1 + 1].view(np.uint32) != whole.view(np.uint32)).any(-1).sum()), "pixels", flush=True)
    if seed % 20 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
for t in pool:
    t.close()
sys.exit(1 if bad else 0)
