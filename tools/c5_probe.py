"""Why is the C5-class stand-in slow?  Rays, time, stack use.  Diagnostic."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
W, H = 1920, 1080
for kw, depth in ((dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False), 12), (dict(tris=5000000, seed=5), 12),
                  (dict(tris=262000, seed=1, emissive_fraction=0.1, light=False), 12)):
    scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **kw)
    tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
    tr.upload_scene(scene); tr.set_camera(camera)
    for i in range(2):
        tr.render_async(0, 0, W - 1, H - 1, 16, exposure=exposure)
        st = tr.stats()
    tr.render_async(0, 0, W - 1, H - 1, 16, exposure=exposure, count_traffic=True)
    ct = tr.stats()
    print(kw, "depth", depth, ": %.1f ms, %.1f Mrays -> %.0f Mray/s; per ray: %.1f box tests, %.1f tris, %.2f taps" % (
        st["kernelMs"], st["raysTraced"] / 1e6, st["raysTraced"] / st["kernelMs"] / 1e3, ct["nBox"] / ct["raysTraced"], ct["nTri"] / ct["raysTraced"],
        ct["nTap"] / ct["raysTraced"]), flush=True)
    tr.close()
