"""The constant a launch pays for its last pixels' chains: kernel time of the whole frame and of one rank's
1-in-8 share with the workgroups' pools capped at PRT_ROWS rows of 64 pixel groups (the product: 16, or fewer when the launch is
small).  A pixel is samples/8 packets x (1 + depth) rounds strictly in sequence; with N groups in flight and a throughput of R groups
per second every pixel spends N / R in flight, and when the row cursor runs dry the pools drain for about that long at a falling
occupancy: T(share) = a + work / rate with a ~ N / (2 R).  Smaller pools shorten `a` and cost steady-state rate.
Needs a library built with -DPRT_TUNING_ENV (tools/build_variants.py): usage  pool_rows_experiment.py LIB c3|c4|c5 [spp [rows,rows,...]]   (PRT_SPREAD=0|1 overrides the spread-rows rule)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd
prt_amd.LIB_PATH = sys.argv[1]
wl = sys.argv[2]
if wl == "c3":
    W, H, spp, depth, kw = 1920, 1080, 64, 8, dict(tris=262000, seed=1)
elif wl == "c4":
    W, H, spp, depth, kw = 1920, 1080, 64, 14, dict(tris=2500000, seed=4)
else:
    W, H, spp, depth, kw = 3840, 2160, 128, 12, dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False)
if len(sys.argv) > 3:
    spp = int(sys.argv[3])
rows_list = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (16, 12, 8, 6, 4, 3, 2)
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, **kw)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)


def t(rank, nranks, reps=2):
    best, rays = 1e30, 0
    for _ in range(reps):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, rank=rank, nranks=nranks)
        st = tr.stats()
        best, rays = min(best, st["kernelMs"]), st["raysTraced"]
    return best, rays


print(f"{wl} {W}x{H} {spp} spp depth {depth}; library {prt_amd.loaded_source_sha16()}; PRT_SPREAD={os.environ.get('PRT_SPREAD', '(product rule)')}", flush=True)
for rows in rows_list:
    os.environ["PRT_ROWS"] = str(rows)
    full, fr = (t(0, 1, 1) if wl == "c5" and spp > 128 else t(0, 1)) if wl != "c5" or rows in (16, 8, 4) else (float("nan"), 0)
    share, sr = t(3, 8)
    # T(share) = a + work / rate, T(full) = a + 8 work / rate (the shares are equal to ~1 %) => a = (8 T(share) - T(full)) / 7
    a = (8 * share - full) / 7 if full == full else float("nan")
    print(f"rows {rows:2d}: frame {full:9.1f} ms ({fr / max(full, 1e-9) / 1e3:6.0f} Mray/s)  1/8 share {share:8.1f} ms ({sr / share / 1e3:6.0f} Mray/s)  "
          f"=> a = {a:7.1f} ms, 8 ranks projected {full / share:4.2f}x of the rows-{rows} frame", flush=True)
tr.close()
