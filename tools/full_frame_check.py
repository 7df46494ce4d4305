"""A BASELINE workload rendered WHOLE by the GPU and by the oracle (all host cores), compared bit for bit: image, ray count,
occlusion-ray count.  The oracle takes minutes at these sizes (C3: 1.3 G rays), which is why the suite checks tiles and properties
instead; this is the one-off whole-frame statement.  usage: full_frame_check.py c2|c3|c4|c5 [rank nranks [firstRow lastRow]]   (a row band of the oracle's part, for runs the box's time limit would cut)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
T.oracle().orc_set_anyhit_accounting(1)
which = sys.argv[1]
rank, nranks = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)
band = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else None
cfg = {"c2": ("setup_bunny_standin", dict(tris=69451, seed=1), 1024, 1024, 64, 14),
       "c3": ("setup_atrium_standin", dict(tris=262000, seed=1), 1920, 1080, 64, 8),
       "c4": ("setup_atrium_standin", dict(tris=2500000, seed=4), 1920, 1080, 256, 14),
       "c5": ("setup_atrium_standin", dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False), 3840, 2160, 1024, 12)}[which]
setup, kw, W, H, spp, depth = cfg
exposure = 64.0 if which == "c5" else 1.0
scene, camera, _ = getattr(prt_amd, setup)(W, H, **kw)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
t0 = time.time()
tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, rank=rank, nranks=nranks)
import ctypes as C
gpu = np.zeros((H, W, 3), dtype=np.float32)
prt_amd._check(prt_amd.lib().prt_hip_download(tr._ctx, gpu.ctypes.data_as(C.c_void_p), 0, 0, W - 1, H - 1), "download")
st = tr.stats()
print(f"{which} rank {rank}/{nranks}: GPU {st['raysTraced'] / 1e9:.3f} G rays in {st['kernelMs']:.0f} ms", flush=True)
own = prt_amd.owned_pixel_mask(W, H, rank, nranks)
if band:
    own[:band[0]] = False
    own[band[1] + 1:] = False
s = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure))
cores = len(os.sched_getaffinity(0))
t0 = time.time()
if nranks == 1:  # in bands of 64 rows, so that a long run keeps reporting
    ref = np.zeros((H, W, 3), dtype=np.float32)
    rays = occl = 0
    for y0 in range(band[0] if band else 0, (band[1] + 1) if band else H, 64):
        y1 = min((band[1] + 1) if band else H, y0 + 64) - 1
        crop, ost = s.render_rect((0, y0, W - 1, y1), spp, max_depth=depth, threads=cores, stats=True)
        ref[y0:y1 + 1] = crop
        rays += ost["raysTraced"]; occl += ost["occludedTraced"]
        print(f"  oracle rows {y0}..{y1}, {time.time() - t0:.0f} s", flush=True)
else:  # the oracle renders the 16x16 tiles this rank owns, one tile row at a time
    ref = np.zeros((H, W, 3), dtype=np.float32)
    rays = occl = 0
    tiles_x = (W + 15) // 16
    for ty in range((band[0] // 16) if band else 0, ((band[1] + 16) // 16) if band else (H + 15) // 16):
        for tx in range(tiles_x):
            if (ty * tiles_x + tx) % nranks != rank:
                continue
            x0, y0, x1, y1 = tx * 16, ty * 16, min(W, tx * 16 + 16) - 1, min(H, ty * 16 + 16) - 1
            crop, ost = s.render_rect((x0, y0, x1, y1), spp, max_depth=depth, threads=cores, stats=True)
            ref[y0:y1 + 1, x0:x1 + 1] = crop
            rays += ost["raysTraced"]; occl += ost["occludedTraced"]
        if ty % 8 == 0:
            print(f"  oracle tile row {ty}, {time.time() - t0:.0f} s", flush=True)
print(f"oracle: {rays / 1e9:.3f} G rays in {time.time() - t0:.0f} s on {cores} threads", flush=True)
same = np.array_equal(gpu[own].view(np.uint32), np.asarray(ref)[own].view(np.uint32))
print(f"{which} rank {rank}/{nranks} {W}x{H} {spp} spp depth {depth}: image {'EQUAL' if same else 'DIFFERENT'} over {int(own.sum())} pixels; "
      f"rays {st['raysTraced']} / {rays}, occlusion rays {st['occludedTraced']} / {occl}", flush=True)
if band:
    print(f"  (rows {band[0]}..{band[1]} only: the ray counts of a band are the oracle's alone)", flush=True)
    sys.exit(0 if same else 1)
sys.exit(0 if same and st["raysTraced"] == rays and st["occludedTraced"] == occl else 1)
