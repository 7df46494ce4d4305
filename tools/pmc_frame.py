"""One frame of a BASELINE workload for a rocprofv3 --pmc pass; prints the frame's kernel time.  Diagnostic.
PRT_PMC_WORKLOAD = c3 (default: 1920x1080, 64 spp, depth 8) | c4 (2.5 M triangles, 1080p, 256 spp, depth 14) |
c5share (5 M triangles, 4K, 1024 spp, depth 12: rank 3 of 8, what one GPU of the node renders)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
if os.environ.get("PRT_LIB"):
    prt_amd.LIB_PATH = os.environ["PRT_LIB"]  # a tuning variant instead of the product library
which = os.environ.get("PRT_PMC_WORKLOAD", "c3")
kw = {}
if which == "c3":
    name, W, H, spp, depth, args = "c3_sponza_standin", 1920, 1080, 64, 8, dict(tris=262000, seed=1)
elif which == "c4":
    name, W, H, spp, depth, args = "c4_sanmiguel_standin", 1920, 1080, 256, 14, dict(tris=2500000, seed=4)
elif which == "c5share":
    name, W, H, spp, depth, args = "c5_zeroday_standin", 3840, 2160, 1024, 12, dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False)
    kw = dict(rank=3, nranks=8)
else:
    raise SystemExit("PRT_PMC_WORKLOAD: c3 | c4 | c5share")
spp = int(os.environ.get("PRT_PMC_SPP", spp))
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **args)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, **kw)
st = tr.stats()
print("library source_sha16", prt_amd.loaded_source_sha16(), flush=True)
print("frame kernel ms", st["kernelMs"], "rays", st["raysTraced"], "workload", name, f"{W}x{H},{spp}spp,depth{depth}" + (",rank3of8" if kw else ""), flush=True)
tr.close()
