"""Kernel time of one rank's share of the C3 frame for 1, 2, 4, 8 ranks (tiles dealt round-robin): how much efficiency a
smaller launch loses.  Diagnostic only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
W, H, spp, depth = 1920, 1080, 64, 8
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
full = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for r in range(n if n <= 2 else 2):
        for i in range(2):
            tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, rank=r, nranks=n)
            st = tr.stats()
        worst = max(worst, st["kernelMs"])
    if full is None: full = worst
    print(f"nranks {n}: {worst:.1f} ms per rank -> speed-up {full / worst:.2f}x of {n}", flush=True)
