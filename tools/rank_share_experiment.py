"""Kernel time of one rank's share of the C3 frame for 1, 2, 4, 8 ranks (tiles dealt round-robin): how much efficiency a
smaller launch loses.  Diagnostic only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
if len(sys.argv) > 1:
    prt_amd.LIB_PATH = sys.argv[1]  # a tuning variant (with -DPRT_TUNING_ENV it honours PRT_SPREAD / PRT_ROWS)
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
if wl == "c3":
    W, H, spp, depth, kw = 1920, 1080, 64, 8, dict(tris=262000, seed=1)
elif wl == "c4":
    W, H, spp, depth, kw = 1920, 1080, 64, 14, dict(tris=2500000, seed=4)
else:
    W, H, spp, depth, kw = 3840, 2160, 64, 12, dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False)
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **kw)
print(f"{wl} {W}x{H} {spp} spp; library {prt_amd.loaded_source_sha16()}; PRT_SPREAD={os.environ.get('PRT_SPREAD', '(product rule)')}", flush=True)
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
