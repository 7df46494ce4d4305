"""Time the C3 frame (or, PRT_BENCH_WORKLOAD=c4 | c5, a shortened frame of those scenes) with every library in prt_amd/lib/var/ (one child process each).  Diagnostic only."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, root)
    import prt_amd
    prt_amd.LIB_PATH = sys.argv[1]
    wl = os.environ.get("PRT_BENCH_WORKLOAD", "c3")  # c3 | c4 (2.5 M triangles, at PRT_BENCH_SPP, default 64) | c5 (5 M triangles, 4K, rank 3 of 8 at PRT_BENCH_SPP, default 128)
    if wl == "c3":
        W, H, spp, depth, kw = 1920, 1080, 64, 8, dict(tris=262000, seed=1)
    elif wl == "c4":
        W, H, spp, depth, kw = 1920, 1080, int(os.environ.get("PRT_BENCH_SPP", "64")), 14, dict(tris=2500000, seed=4)
    else:
        W, H, spp, depth, kw = 3840, 2160, int(os.environ.get("PRT_BENCH_SPP", "128")), 12, dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False)
    scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **kw)
    tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
    tr.upload_scene(scene); tr.set_camera(camera)
    ms = []
    for i in range(2):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, **(dict(rank=3, nranks=8) if wl == "c5" else {}))
        st = tr.stats()
        ms.append(st["kernelMs"])
    share = []
    for i in range(2):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, rank=1, nranks=8)
        share.append(tr.stats()["kernelMs"])
    print(f"{os.path.basename(sys.argv[1])} {os.environ.get('PRT_FRAME_BPC', '')}: {min(ms):.1f} ms  ({st['raysTraced'] / min(ms) / 1e3:.0f} Mray/s); 1/8 share {min(share):.1f} ms", flush=True)
    tr.close()
else:
    # every library several times, interleaved (one child process per run): frame times wander by 2-3 % between processes
    var = os.path.join(root, "prt_amd", "lib", "var")
    rounds = int(os.environ.get("PRT_BENCH_ROUNDS", "3"))
    for r in range(rounds):
        for f in sorted(os.listdir(var)):
            subprocess.call([sys.executable, os.path.abspath(__file__), os.path.join(var, f)])
