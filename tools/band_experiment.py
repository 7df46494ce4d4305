"""Does L2 locality matter?  Render C3 as 8 horizontal bands (all XCDs on one band at a time) and compare the summed kernel
time with the one-launch frame.  Diagnostic only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd

W, H, spp, depth = 1920, 1080, 64, 8
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)

def timed(rects):
    tr.stats()
    rays = 0; ms = 0.0
    out = []
    for (x0, y0, x1, y1) in rects:
        tr.render_async(x0, y0, x1, y1, spp, exposure=exposure)
        st = tr.stats()
        r = st["raysTraced"] + st["occludedTraced"]
        out.append((y0, y1, x0, x1, st["kernelMs"], r, r / st["kernelMs"] / 1e3))
        rays += r; ms += st["kernelMs"]
    return out, rays, ms

for name, rects in [("full", [(0, 0, W - 1, H - 1)]),
                    ("8 bands", [(0, k * 135, W - 1, k * 135 + 134) for k in range(8)]),
                    ("8 columns", [(k * 240, 0, k * 240 + 239, H - 1) for k in range(8)]),
                    ("32 blocks", [(i * 480, k * 135, i * 480 + 479, k * 135 + 134) for k in range(8) for i in range(4)])]:
    timed(rects[:1])
    out, rays, ms = timed(rects)
    print(f"{name}: {ms:.1f} ms, {rays/1e6:.1f} Mrays, {rays/ms/1e3:.1f} Mray/s")
    if len(out) <= 8:
        for o in out: print("   rows %d-%d cols %d-%d: %.1f ms %.0f rays %.1f Mray/s" % o)
