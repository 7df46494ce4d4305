"""What does a rank's share cost as a function of WHERE its pixels lie?  Times, on one GPU: the whole frame, one rank's share under the
product's tile-to-rank rule, and the eight contiguous bands of 1/8 of the image's tile rows (whose sum is the whole frame again).
If the bands sum to about the whole frame while eight shares sum to far more, the rule costs coherence.  Diagnostic only.
usage: share_mapping_experiment.py c3|c4|c5 [spp]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import prt_amd
wl = sys.argv[1]
W, H, spp, depth, kw = {"c3": (1920, 1080, 64, 8, dict(tris=262000, seed=1)), "c4": (1920, 1080, 64, 14, dict(tris=2500000, seed=4)),
                        "c5": (3840, 2160, 128, 12, dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False))}[wl]
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **kw)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
def run(rect=None, **k):
    x0, y0, x1, y1 = rect or (0, 0, W - 1, H - 1)
    best = None
    for _ in range(2):
        tr.render_async(x0, y0, x1, y1, spp, exposure=exposure, **k)
        st = tr.stats()
        if best is None or st["kernelMs"] < best[0]:
            best = (st["kernelMs"], st["raysTraced"])
    return best
full = run()
print(f"{wl} {W}x{H} {spp} spp: whole frame {full[0]:.1f} ms, {full[1] / full[0] / 1e3:.0f} Mray/s", flush=True)
tot = 0.0
for r in range(8):
    ms, rays = run(rank=r, nranks=8)
    tot += ms
    print(f"  share of rank {r} of 8: {ms:.1f} ms, {rays / ms / 1e3:.0f} Mray/s", flush=True)
print(f"  eight shares: {tot:.1f} ms = {tot / full[0]:.3f} x the whole frame; slowest rank x 8 = {0:.1f}", flush=True)
rows = (H + 15) // 16
tot = 0.0
worst = 0.0
for b in range(8):
    y0, y1 = (rows * b // 8) * 16, min(H, (rows * (b + 1) // 8) * 16) - 1
    ms, rays = run((0, y0, W - 1, y1))
    tot += ms
    worst = max(worst, ms)
    print(f"  band rows {y0}..{y1}: {ms:.1f} ms, {rays / ms / 1e3:.0f} Mray/s", flush=True)
print(f"  eight bands: {tot:.1f} ms = {tot / full[0]:.3f} x the whole frame; the slowest band {worst:.1f} ms", flush=True)
tr.close()
