"""C3 kernel time per library variant in prt_amd/lib/var/ (tools/build_variants.py) and GPU_MAX_HW_QUEUES setting, for the whole
frame and for one rank's share of 8.  usage: matrix_bench.py 4,16   Diagnostic only."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import prt_amd
    prt_amd.LIB_PATH = sys.argv[2]
    W, H, spp, depth = 1920, 1080, 64, 8
    scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
    tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
    tr.upload_scene(scene); tr.set_camera(camera)
    out = []
    for n in (1, 8):
        ms = []
        for i in range(3):
            tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, rank=0, nranks=n)
            ms.append(tr.stats()["kernelMs"])
        out.append(f"1/{n}: {min(ms):.1f} ms")
    print(sys.argv[3], "  ".join(out), flush=True)
    tr.close()
else:
    var = os.path.join(root, "prt_amd", "lib", "var")
    for f in sorted(os.listdir(var)):
        for q in sys.argv[1].split(","):
            env = dict(os.environ); env["GPU_MAX_HW_QUEUES"] = q
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", os.path.join(var, f), f"{f} hwq={q}"], env=env)
