"""Time the C3 frame under several settings of a tuning environment variable (one child process each).  Diagnostic only.
usage: env_bench.py NAME v1 v2 ..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import prt_amd
    W, H, spp, depth = 1920, 1080, 64, 8
    scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
    tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
    tr.upload_scene(scene); tr.set_camera(camera)
    ms = []
    for i in range(3):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure)
        st = tr.stats()
        ms.append(st["kernelMs"])
    print(f"{sys.argv[2]}: {min(ms):.1f} ms  ({st['raysTraced'] / min(ms) / 1e3:.0f} Mray/s)", flush=True)
    tr.close()
else:
    name = sys.argv[1]
    for v in sys.argv[2:]:
        env = dict(os.environ); env[name] = v
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", f"{name}={v}"], env=env)
