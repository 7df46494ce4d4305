"""Host time spent enqueueing a frame vs its GPU time.  Diagnostic only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
W, H, spp, depth = 1920, 1080, 64, 8
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
for n in (1, 2, 4, 8):
    for i in range(3):
        t0 = time.perf_counter()
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, rank=0, nranks=n)
        t1 = time.perf_counter()
        st = tr.stats()
        t2 = time.perf_counter()
    print(f"nranks {n}: enqueue {1e3*(t1-t0):.1f} ms, until done {1e3*(t2-t0):.1f} ms, kernel events {st['kernelMs']:.1f} ms", flush=True)
