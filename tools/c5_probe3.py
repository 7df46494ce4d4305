import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
W, H, tris, spp = 3840, 2160, 5000000, 8
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=tris, seed=5, emissive_fraction=0.1, light=False)
tr = prt_amd.PathTracer(device=0, max_depth=12, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
for rect in ((0, 0, W - 1, 1079), (0, 1080, W - 1, H - 1), (0, 0, W - 1, 539), (0, 0, 1919, 1079), (0, 0, W - 1, H - 1), (0, 0, W - 1, 1023), (0, 0, W - 1, 1039)):
    for i in range(2):
        tr.render_async(*rect, spp, exposure=exposure)
        st = tr.stats()
    print(rect, ": %.1f ms, %.1f Mrays -> %.0f Mray/s" % (st["kernelMs"], st["raysTraced"] / 1e6, st["raysTraced"] / st["kernelMs"] / 1e3), flush=True)
