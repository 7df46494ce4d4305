import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import prt_amd
    prt_amd.LIB_PATH = sys.argv[1]
    out = []
    for name, setup, kw, W, H, spp, depth in (("c2", prt_amd.setup_bunny_standin, dict(tris=69451), 1024, 1024, 64, 4), ("c3", prt_amd.setup_atrium_standin, dict(tris=262000, seed=1), 1920, 1080, 64, 8)):
        scene, camera, exposure = setup(W, H, **kw)
        tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
        tr.upload_scene(scene); tr.set_camera(camera)
        ms = []
        for i in range(3):
            tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure)
            ms.append(tr.stats()["kernelMs"])
        out.append(f"{name} {min(ms):.1f} ms")
        tr.close()
    print(os.path.basename(sys.argv[1]), "; ".join(out), flush=True)
else:
    var = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'prt_amd', 'lib', 'var')
    for r in range(3):
        for f in sorted(os.listdir(var)):
            subprocess.call([sys.executable, __file__, os.path.join(var, f)])
