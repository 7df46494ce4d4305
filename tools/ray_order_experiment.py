"""What could a ray-coherence stage buy AT MOST?    One batch of incoherent scatter rays of a BASELINE scene --
third-generation bounce rays: camera rays -> hit points -> random directions off the surface, three times over -- is traced by the
row-level rays kernel (the frame kernel's own stepper: persistent lanes, cooperative leaf rounds; consecutive rays of the batch go to
one wave, consecutive waves to one workgroup) in several ORDERS of the same rays:
    natural   the order a path tracer produces them in (pixel order)
    shuffled  a random permutation
    octant    sorted by the direction's octant                       (what rings per octant inside a workgroup can reach at best)
    morton    sorted by a 30-bit Morton code of the origin           (a global sort: every ray of the launch takes part)
    both      octant, then Morton
Results are identical in every order (checked).  The ratio natural / sorted time is the ceiling of ANY reordering stage in front of the
traversal, before its own cost.   usage: ray_order_experiment.py c3|c4|c5 [rays]     (test build of the library)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
want = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
if wl == "c3":
    W, H, kw = 1920, 1080, dict(tris=262000, seed=1)
elif wl == "c4":
    W, H, kw = 1920, 1080, dict(tris=2500000, seed=4)
else:
    W, H, kw = 3840, 2160, dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False)
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **kw)
tr = prt_amd.PathTracer(device=0, test_entry_points=True)
tr.upload_scene(scene); tr.set_camera(camera)
rng = np.random.default_rng(7)
d = camera.desc
pos, fwd, up, right = (np.array(list(v), dtype=np.float32) for v in (d.pos, d.dir, d.up, d.right))
# camera rays of a sub-grid of the image (camera.cpp:46-56 without the jitter)
n0 = want * 2
gx = int((n0 * W / H) ** 0.5) // 8 * 8
gy = n0 // gx
xs, ys = np.meshgrid((np.arange(gx) + 0.5) / gx, (np.arange(gy) + 0.5) / gy)
nx = 2.0 * (xs.ravel() - 0.5) * 0.6 * (W / H)
ny = -2.0 * (ys.ravel() - 0.5) * 0.6
dirs = nx[:, None] * right + ny[:, None] * up + fwd
dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
org = np.broadcast_to(pos, dirs.shape).astype(np.float32).copy()
FAR = 1e5


def trace(o, dd):
    n = len(o) // 8 * 8
    h = tr.trace_rays(0, o[:n], dd[:n], FAR)
    ms = tr.stats()["kernelMs"]
    return h, ms, n


for gen in range(3):
    h, ms, n = trace(org, dirs)
    hit = h["t"] != -1
    print(f"generation {gen}: {n} rays, {int(hit.sum())} hits, {ms:.1f} ms = {n / ms / 1e3:.0f} Mray/s", flush=True)
    o, dd, t = org[:n][hit], dirs[:n][hit], h["t"][hit]
    p = o + (t[:, None] - 1e-3) * dd
    nd = rng.normal(size=p.shape).astype(np.float32)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    flip = (nd * dd).sum(axis=1) > 0           # leave on the side the ray came from
    nd[flip] = -nd[flip]
    org, dirs = p.astype(np.float32), nd.astype(np.float32)
n = min(len(org), want) // 8 * 8
org, dirs = org[:n], dirs[:n]
lo, hi = org.min(axis=0), org.max(axis=0)
q = np.clip(((org - lo) / np.maximum(hi - lo, 1e-20) * 1024).astype(np.uint64), 0, 1023)


def spread(v):
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
octant = ((dirs[:, 0] < 0).astype(np.uint64) | ((dirs[:, 1] < 0).astype(np.uint64) << 1) | ((dirs[:, 2] < 0).astype(np.uint64) << 2))
orders = {
    "natural": np.arange(n),
    "shuffled": rng.permutation(n),
    "octant": np.argsort(octant, kind="stable"),
    "morton": np.argsort(morton, kind="stable"),
    "both (octant, morton)": np.argsort((octant << 30) | morton, kind="stable"),
    "both (morton 15 bits, octant)": np.argsort(((morton >> 15) << 3) | octant, kind="stable"),
}
if n > (1 << 22):  # a large batch (the throughput regime: dozens of rays per lane): the orders that matter
    orders = {k: v for k, v in orders.items() if k in ("natural", "shuffled", "morton", "both (morton 15 bits, octant)")}
print(f"{wl}: {n} third-generation scatter rays; library {prt_amd.loaded_source_sha16() if hasattr(prt_amd, 'loaded_source_sha16') else ''}", flush=True)
base_hits, base_ms = None, None
for name, perm in orders.items():
    best = 1e30
    for rep in range(2):
        h, ms, _ = trace(org[perm], dirs[perm])
        best = min(best, ms)
    back = np.empty_like(h)
    back[perm] = h
    if base_hits is None:
        base_hits, base_ms = back, best
    same = back.tobytes() == base_hits.tobytes()
    print(f"  {name:32s} {best:8.2f} ms = {n / best / 1e3:7.0f} Mray/s   x{base_ms / best:5.2f} of natural   results identical: {same}", flush=True)
tr.close()
