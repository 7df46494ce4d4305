"""The ORACLE's whole frame of a BASELINE configuration as per-tile SHA-256 digests: the fixture that lets the GPU suite compare a
whole frame at the configuration's own sample count in every run (the oracle needs minutes to an hour of all cores for one of
these frames, the suite must not).

    whole_frame_digests.py c2|c3|c4 OUT.npz      -> sha[tilesY, tilesX, 32] of the 16x16 tiles (edge tiles are what is left of them),
                                                    the oracle's ray and occlusion-ray counts, and the parameters of the frame

The scene comes from the product's generator (as in the GPU tests that use the fixture), the pixels from oracle/prt_oracle.c.
Config 5 is too large for this (44 G rays): tools/c5_split_check.py covers it by tile rows."""
import hashlib
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T

CONFIGS = {  # setup, kwargs, W, H, spp, depth -- as the GPU tests test_full_size_c{2,3,4}_* render them
    "c2": ("setup_bunny_standin", {}, 1024, 1024, 64, 14),
    "c3": ("setup_atrium_standin", dict(tris=262000, seed=1), 1920, 1080, 64, 8),
    "c4": ("setup_atrium_standin", dict(tris=2500000, seed=4), 1920, 1080, 256, 14),
}
TILE = 16


def tile_digests(img):
    H, W, _ = img.shape
    ty, tx = (H + TILE - 1) // TILE, (W + TILE - 1) // TILE
    sha = np.zeros((ty, tx, 32), dtype=np.uint8)
    for r in range(ty):
        for c in range(tx):
            t = np.ascontiguousarray(img[r * TILE:(r + 1) * TILE, c * TILE:(c + 1) * TILE], dtype=np.float32)
            sha[r, c] = np.frombuffer(hashlib.sha256(t.view(np.uint32).tobytes()).digest(), dtype=np.uint8)
    return sha


def main():
    which, out = sys.argv[1], sys.argv[2]
    setup, kw, W, H, spp, depth = CONFIGS[which]
    scene, camera, exposure = getattr(prt_amd, setup)(W, H, **kw)
    T.oracle().orc_set_anyhit_accounting(1)
    s = T.OracleScene(T.scene_desc_from_product(scene, camera, exposure))
    threads = int(os.environ.get("PRT_ORACLE_THREADS", "0")) or len(os.sched_getaffinity(0))
    img = np.zeros((H, W, 3), dtype=np.float32)
    rays = occl = 0
    t0 = time.time()
    for y0 in range(0, H, 64):  # bands, so that a long run keeps reporting
        y1 = min(H, y0 + 64) - 1
        crop, st = s.render_rect((0, y0, W - 1, y1), spp, max_depth=depth, threads=threads, stats=True)
        img[y0:y1 + 1] = crop
        rays += st["raysTraced"]
        occl += st["occludedTraced"]
        print(f"{which}: rows {y0}..{y1}, {rays / 1e6:.0f} M rays so far, {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(out, sha=tile_digests(img), width=W, height=H, spp=spp, max_depth=depth, exposure=np.float32(exposure), seed=12345,
                        rays=np.uint64(rays), occluded=np.uint64(occl))
    print(f"{which} {W}x{H} {spp} spp depth {depth}: {rays} rays, {occl} occlusion rays in {time.time() - t0:.0f} s on {threads} threads -> {out}", flush=True)


if __name__ == "__main__":
    main()
