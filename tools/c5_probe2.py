import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
for (W, H, tris, spp) in ((3840, 2160, 262000, 8), (3840, 2160, 262000, 16), (1920, 1080, 262000, 8), (3840, 2160, 5000000, 8), (1920, 1080, 5000000, 8), (2560, 1440, 5000000, 8)):
    scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=tris, seed=5, emissive_fraction=0.1, light=False)
    tr = prt_amd.PathTracer(device=0, max_depth=12, seed=12345)
    tr.upload_scene(scene); tr.set_camera(camera)
    for i in range(2):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure)
        st = tr.stats()
    print((W, H, tris, spp), ": %.1f ms, %.1f Mrays -> %.0f Mray/s" % (st["kernelMs"], st["raysTraced"] / 1e6, st["raysTraced"] / st["kernelMs"] / 1e3), flush=True)
    tr.close()
