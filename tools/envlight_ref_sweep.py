"""InfiniteAreaLight::create + sample: the oracle against the reference's compiled light.cpp (oracle/_ref/ref_path envlight; build container
only) on random float environment maps -- 2..48 x 2..24 texels, black rows (never the last) and columns, texels of 1e-3 .. 1e4, a very
bright texel anywhere including (0, 0) -- and 4096 draws each: uniform ones, draws equal to CDF entries and one ulp either side, 0 and
the largest float below 1 (kept below the last vertical entry: past it the reference reads beyond its table).  Both CDF tables,
directions and colours must be identical.  usage: envlight_ref_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
if T.ref_binary("ref_path") is None:
    sys.exit("oracle/_ref/ref_path is not built (needs /root/reference)")
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(95000 + seed)
    w, h = int(rng.integers(2, 49)), int(rng.integers(2, 25))  # (a one-row map always runs the reference off its vertical table)
    env = np.ones((h, w, 4), dtype=np.float32)
    env[..., :3] = np.exp(rng.uniform(np.log(1e-3), np.log(1e1), (h, w, 3))).astype(np.float32)
    if seed % 2 and h > 1:
        env[rng.integers(0, h - 1), :, :3] = 0.0
    if seed % 3 == 0:
        env[:, rng.integers(0, w), :3] = 0.0
    if seed % 4 == 0:
        env[0 if seed % 8 == 0 else rng.integers(0, h), 0 if seed % 8 == 0 else rng.integers(0, w), :3] = float(rng.choice([300.0, 1e4]))
    scene, camera, _ = prt_amd.setup_cornell_box(16, 16)
    scene.set_infinite_area_light(env)
    desc = T.scene_desc_from_product(scene, camera, 1.0)
    s = T.OracleScene(desc)
    vp, hp = s.env_tables()
    last = vp[-1]
    if not np.isfinite(last):
        continue
    u = rng.random((4096, 2)).astype(np.float32)
    k = 256
    ent = np.concatenate([vp[np.isfinite(vp)], hp[np.isfinite(hp)]])
    pick = ent[rng.integers(0, len(ent), 3 * k)]
    u[:k, 1] = pick[:k]; u[k:2 * k, 1] = np.nextafter(pick[k:2 * k], np.float32(-1)); u[2 * k:3 * k, 1] = np.nextafter(pick[2 * k:3 * k], np.float32(2))
    pick = ent[rng.integers(0, len(ent), 3 * k)]
    u[3 * k:4 * k, 0] = pick[:k]; u[4 * k:5 * k, 0] = np.nextafter(pick[k:2 * k], np.float32(-1)); u[5 * k:6 * k, 0] = np.nextafter(pick[2 * k:3 * k], np.float32(2))
    u[6 * k:6 * k + 8] = [[0, 0], [0, 0.5], [0.5, 0], [np.nextafter(np.float32(1), np.float32(0)), 0.25], [0.25, 0], [0.75, 0.1], [1e-30, 1e-30], [0.5, 0.5]]
    u = np.clip(u, 0.0, np.nextafter(np.float32(1), np.float32(0))).astype(np.float32)
    u[:, 1] = np.minimum(u[:, 1], np.nextafter(last, np.float32(-1)))  # stay on the vertical table
    u[:, 1] = np.maximum(u[:, 1], 0.0)
    rvp, rhp, rdir, rcol = T.ref_envlight(desc, u)
    odir, ocol = s.env_sample(u)
    def same(a, b):
        a, b = np.asarray(a), np.asarray(b)
        n = np.isnan(b)
        return np.array_equal(np.isnan(a), n) and np.array_equal(a[~n].view(np.uint32), b[~n].view(np.uint32))
    ok = same(vp, rvp) and same(hp, rhp) and same(odir, rdir) and same(ocol, rcol)
    if not ok:
        bad += 1
        print("seed", seed, (w, h), "MISMATCH tables", same(vp, rvp), same(hp, rhp), "dir", int((odir.view(np.uint32) != rdir.view(np.uint32)).any(1).sum()),
              "colour", int((ocol.view(np.uint32) != rcol.view(np.uint32)).any(1).sum()), flush=True)
    if seed % 50 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
