"""Wave-time shares of the frame kernel's roles (shade passes, the four traversals, idle) on one C3 frame and on a 1/8 share,
from the -DPRT_PROFILE build (s_memtime stamps around the role calls; never a timed build).  Prints the library's report.
usage: role_profile.py <libprt_hip built with -DPRT_PROFILE>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
prt_amd.LIB_PATH = sys.argv[1]
wl = os.environ.get("PRT_BENCH_WORKLOAD", "c3")  # c3 | c4 (64 spp) | c5 (128 spp; "rank 1 of 8" is what a GPU of the node renders)
if wl == "c3":
    W, H, spp, depth, kw = 1920, 1080, 64, 8, dict(tris=262000, seed=1)
elif wl == "c4":
    W, H, spp, depth, kw = 1920, 1080, 64, 14, dict(tris=2500000, seed=4)
else:
    W, H, spp, depth, kw = 3840, 2160, 128, 12, dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False)
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **kw)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
for label, kw in ((("full frame", {}),) if wl != "c5" else ()) + (("rank 1 of 8", dict(rank=1, nranks=8)),):
    for i in range(1 if wl != "c3" else 2):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, **kw)
        sys.stderr.write(f"--- {label}, run {i}\n"); sys.stderr.flush()
        st = tr.stats()
    print(label, st["kernelMs"], "ms", flush=True)
tr.close()
