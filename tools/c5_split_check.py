"""BASELINE config 5 (3840x2160, 1024 spp, depth 12, 5 M triangles) compared WHOLE-FRAME-WISE in two halves that need not run
on the same machine: the oracle needs hours of CPU for this frame (44 G rays), a GPU box is held for minutes.

    c5_split_check.py gpu OUT.npz                  on the GPU box: the whole frame in one launch; per 16x16 tile the SHA-256 of its
                                                   float pixels, its float64 channel sums, and the launch's ray count
    c5_split_check.py fixture IN.npz OUT.npz CPU.log...  the digests of the tile rows the `cpu` runs reported as equal in EVERY tile -- the
                                                   oracle's output for those rows (equal SHA-256 = equal bytes) -- as the
                                                   fixture of the GPU suite (tests/golden/c5_tile_rows.npz)
    c5_split_check.py cpu IN.npz MINUTES [ROW0 [SKIP.npz]]  (SKIP.npz: a fixture whose rows are done already and are left out)
    c5_split_check.py cpu IN.npz MINUTES [ROW0]    anywhere: the oracle (all threads given by PRT_ORACLE_THREADS, default all cores)
                                                   renders tile rows from ROW0 (default: the middle of the image) outwards for
                                                   MINUTES and compares every finished tile with the GPU's digest; one line per
                                                   tile row, a summary at the end (exit 1 on any differing tile)

The scene comes from the product's own generator on both sides (same seed, same library), the oracle from oracle/prt_oracle.c."""
import hashlib
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd

W, H, SPP, DEPTH, EXPOSURE, TILE = 3840, 2160, 1024, 12, 64.0, 16
KW = dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False)


def tile_digests(img, y0, y1):
    """(sha[rows, tilesX, 32] uint8, sums[rows, tilesX, 3] float64) of the 16x16 tiles of pixel rows y0..y1 (tile-aligned)."""
    tx = (W + TILE - 1) // TILE
    rows = (y1 + 1 - y0) // TILE
    sha = np.zeros((rows, tx, 32), dtype=np.uint8)
    sums = np.zeros((rows, tx, 3), dtype=np.float64)
    for r in range(rows):
        for c in range(tx):
            t = np.ascontiguousarray(img[y0 + r * TILE:y0 + (r + 1) * TILE, c * TILE:(c + 1) * TILE])
            sha[r, c] = np.frombuffer(hashlib.sha256(t.view(np.uint32).tobytes()).digest(), dtype=np.uint8)
            sums[r, c] = t.astype(np.float64).sum(axis=(0, 1))
    return sha, sums


def main():
    mode = sys.argv[1]
    if mode == "fixture":
        import re
        z = np.load(sys.argv[2])
        tx = (W + TILE - 1) // TILE
        text = "\n".join(open(f).read() for f in sys.argv[4:])
        assert "DIFFERENT" not in re.sub(r"\b0 DIFFERENT", "", text), "a run reported differing tiles"
        rows = sorted({int(m.group(1)) for m in re.finditer(r"^tile row (\d+) .*: (\d+) of (\d+) tiles equal", text, re.M)
                       if m.group(2) == m.group(3) == str(tx)})
        np.savez_compressed(sys.argv[3], rows=np.array(rows, dtype=np.int32), sha=z["sha"][rows], width=W, height=H, spp=SPP, max_depth=DEPTH,
                            exposure=np.float32(EXPOSURE), seed=12345)
        print(f"{len(rows)} tile rows ({len(rows) * tx} tiles, {100.0 * len(rows) * TILE / H:.1f} % of the frame): rows {rows}")
        return 0
    scene, camera, _ = prt_amd.setup_atrium_standin(W, H, **KW)
    if mode == "gpu":
        import ctypes as C
        tr = prt_amd.PathTracer(device=0, max_depth=DEPTH, seed=12345)
        tr.upload_scene(scene)
        tr.set_camera(camera)
        tr.render_async(0, 0, W - 1, H - 1, SPP, exposure=EXPOSURE)
        img = np.zeros((H, W, 3), dtype=np.float32)
        prt_amd._check(prt_amd.lib().prt_hip_download(tr._ctx, img.ctypes.data_as(C.c_void_p), 0, 0, W - 1, H - 1), "download")
        st = tr.stats()
        print(f"c5 whole frame on one GPU: {st['raysTraced']} rays in {st['kernelMs']:.0f} ms = {st['raysTraced'] / st['kernelMs'] / 1e3:.1f} Mray/s; "
              f"library {prt_amd.loaded_source_sha16()}", flush=True)
        sha, sums = tile_digests(img, 0, H - 1)
        np.savez_compressed(sys.argv[2], sha=sha, sums=sums, rays=np.uint64(st["raysTraced"]), kernel_ms=np.float64(st["kernelMs"]),
                            source_sha16=np.array(prt_amd.loaded_source_sha16()))
        tr.close()
        return 0
    import prt_testlib as T
    z = np.load(sys.argv[2])
    minutes = float(sys.argv[3])
    tiles_y = H // TILE
    row0 = int(sys.argv[4]) if len(sys.argv) > 4 else tiles_y // 2
    threads = int(os.environ.get("PRT_ORACLE_THREADS", "0")) or len(os.sched_getaffinity(0))
    T.oracle().orc_set_anyhit_accounting(1)
    s = T.OracleScene(T.scene_desc_from_product(scene, camera, EXPOSURE))
    # tile rows row0, row0 + 1, row0 - 1, row0 + 2, ... : the middle of the image (the atrium's floor, cards and lamps) first
    order = [row0]
    for d in range(1, tiles_y):
        for r in (row0 + d, row0 - d):
            if 0 <= r < tiles_y:
                order.append(r)
    if len(sys.argv) > 5:
        have = set(int(r) for r in np.load(sys.argv[5])["rows"])
        order = [r for r in order if r not in have]
    t0 = time.time()
    equal = differing = 0
    rays = 0
    done_rows = []
    for r in order:
        if time.time() - t0 > minutes * 60.0:
            break
        y0, y1 = r * TILE, r * TILE + TILE - 1
        crop, ost = s.render_rect((0, y0, W - 1, y1), SPP, max_depth=DEPTH, threads=threads, stats=True)
        band = np.zeros((TILE, W, 3), dtype=np.float32)
        band[:] = crop
        sha, _ = tile_digests(_place(band, y0), y0, y1)
        same = (sha[0] == z["sha"][r]).all(axis=1)
        equal += int(same.sum())
        differing += int((~same).sum())
        rays += ost["raysTraced"]
        done_rows.append(r)
        print(f"tile row {r} (pixel rows {y0}..{y1}): {int(same.sum())} of {len(same)} tiles equal, {ost['raysTraced'] / 1e6:.0f} M rays, "
              f"{time.time() - t0:.0f} s elapsed", flush=True)
    px = len(done_rows) * TILE * W
    print(f"c5 oracle vs GPU digests (library {str(z['source_sha16'])}): tile rows {sorted(done_rows)[:1]}..{sorted(done_rows)[-1:]} ({len(done_rows)} rows, {px} pixels = "
          f"{100.0 * px / (W * H):.1f} % of the frame, {rays / 1e9:.2f} G oracle rays on {threads} threads in {time.time() - t0:.0f} s): "
          f"{equal} tiles EQUAL, {differing} DIFFERENT", flush=True)
    return 1 if differing else 0


def _place(band, y0):
    """A full-height view whose rows y0.. are `band` (only those rows are read by tile_digests)."""
    class V:
        def __getitem__(self, key):
            ys, xs = key
            return band[ys.start - y0:ys.stop - y0, xs]
    return V()


if __name__ == "__main__":
    sys.exit(main())
