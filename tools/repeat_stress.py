"""Race hunt: the same frame rendered over and over must give the same bytes and the same ray counts every time (the frame kernel's
hand-offs go through LDS words, rings and a lock; a lost update would show as a changed pixel or count).  C1 x 300, C3 x 40, a 1/8 share of
C3 x 60, C2 x 60.  usage: repeat_stress.py"""
import os, sys, hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import ctypes as C
import prt_amd
bad = 0
for name, setup, kw, W, H, spp, depth, reps, rk in (("c1", prt_amd.setup_cornell_box, {}, 512, 512, 16, 4, 300, (0, 1)),
                                                     ("c2", prt_amd.setup_bunny_standin, dict(tris=69451), 1024, 1024, 64, 14, 60, (0, 1)),
                                                     ("c3", prt_amd.setup_atrium_standin, dict(tris=262000, seed=1), 1920, 1080, 64, 8, 40, (0, 1)),
                                                     ("c3 rank 5 of 8", prt_amd.setup_atrium_standin, dict(tris=262000, seed=1), 1920, 1080, 64, 8, 60, (5, 8))):
    scene, camera, exposure = setup(W, H, **kw)
    tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
    tr.upload_scene(scene); tr.set_camera(camera)
    first = None
    img = np.zeros((H, W, 3), dtype=np.float32)
    for i in range(reps):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, rank=rk[0], nranks=rk[1])
        prt_amd._check(prt_amd.lib().prt_hip_download(tr._ctx, img.ctypes.data_as(C.c_void_p), 0, 0, W - 1, H - 1), "download")
        st = tr.stats()
        key = (hashlib.sha256(img.tobytes()).hexdigest(), st["raysTraced"], st["occludedTraced"])
        if first is None:
            first = key
        elif key != first:
            bad += 1
            print(name, "run", i, "DIFFERS", key[1:], first[1:], flush=True)
    print(name, reps, "runs,", "all identical" if bad == 0 else "see above", first[1], "rays", flush=True)
    tr.close()
print("stress finished,", bad, "differing runs", flush=True)
sys.exit(1 if bad else 0)
