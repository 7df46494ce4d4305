"""One C3 frame (1920x1080, 64 spp, depth 8) for a rocprofv3 --pmc pass; prints the frame's kernel time.  Diagnostic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
if os.environ.get("PRT_LIB"):
    prt_amd.LIB_PATH = os.environ["PRT_LIB"]  # a tuning variant instead of the product library
W, H, spp, depth = 1920, 1080, 64, 8
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure)
st = tr.stats()
print("frame kernel ms", st["kernelMs"], "rays", st["raysTraced"], flush=True)
tr.close()
