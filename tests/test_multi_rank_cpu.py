"""world_size-2 (and 3) test of the multi-GPU host path on CPU (gloo): tile ownership + the image gather.  The kernels
cannot run here, so each rank fills its own tiles with the ORACLE's pixels (the checker standing in for the GPU), exactly as
bench.py's ranks fill theirs with prt_hip_render(rank, nranks), and poisons every pixel it does not own; the owned tiles are
packed tile-major, sent to rank 0 and de-interleaved there -- the data movement of prt_hip_gather_rccl (prt_gather.hip)
on host tensors.  Rank 0 must end up with the single-rank image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import prt_amd
    import prt_testlib as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp = 72, 40, 8  # not multiples of the tile size
    scene = T.OracleScene(T.cornell_scene(W, H, with_teapot=False))
    own = prt_amd.owned_pixel_mask(W, H, rank, world)
    fb = np.zeros((H, W, 3), dtype=np.float32)
    for ty in range(0, H, 16):
        for tx in range(0, W, 16):
            if own[ty, tx]:
                x1, y1 = min(tx + 15, W - 1), min(ty + 15, H - 1)
                crop, _ = scene.trace_block(tx, ty, x1, y1, spp)
                fb[ty:y1 + 1, tx:x1 + 1] = crop
    fb[~own] = np.float32(-7.0)  # nothing a rank does not own may reach the image: the gather moves owned tiles only
    payload = prt_amd.pack_tiles(fb, rank, world).nbytes
    assert payload == len(prt_amd.owned_tile_ids(W, H, rank, world)) * 16 * 16 * 12  # 1/world of the (tile-padded) image
    t = torch.from_numpy(prt_amd.gather_image_host(fb, rank, world, dst=0))
    # every pixel has exactly one owner
    cover = torch.from_numpy(own.astype(np.int32))
    dist.all_reduce(cover)
    assert (cover == 1).all()
    if rank == 0:
        full, _ = scene.render(spp)
        np.save(out_path, np.stack([t.numpy(), full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharding_and_gather(tmp_path, world):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got, ref = np.load(out)
    assert got.tobytes() == ref.tobytes()


def test_owned_pixel_mask_partitions_the_image():
    import prt_amd
    for (w, h, n) in ((1920, 1080, 8), (100, 52, 3), (16, 16, 2), (17, 1, 4)):
        total = sum(prt_amd.owned_pixel_mask(w, h, r, n).astype(np.int64) for r in range(n))
        assert (total == 1).all()
        counts = [int(prt_amd.owned_pixel_mask(w, h, r, n).sum()) for r in range(n)]
        if w * h >= 16 * 16 * n * 4:
            assert max(counts) - min(counts) <= 4 * 256  # round-robin keeps the shares within a few tiles


def test_pack_and_unpack_are_inverse_and_tile_major():
    import prt_amd
    rng = np.random.default_rng(3)
    for (w, h, n) in ((72, 40, 3), (1920, 1080, 8), (16, 16, 2), (33, 17, 5)):
        img = rng.random((h, w, 3), dtype=np.float32)
        out = np.full_like(img, -1)
        for r in range(n):
            p = prt_amd.pack_tiles(img, r, n)
            assert p.shape[0] == len(prt_amd.owned_tile_ids(w, h, r, n))
            prt_amd.unpack_tiles(out, p, r, n)
        assert out.tobytes() == img.tobytes()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts torch.distributed.run as a CHILD:
    the two ranks see RANK / WORLD_SIZE / a 127.0.0.1 rendezvous (probe mode stops them before a device is needed), and without
    the probe the run fails with "need 2 devices" -- here, where there are none -- not with a usage message."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    probe = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, PRT_BENCH_LAUNCH_PROBE="1"),
                           capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0, probe.stderr[-2000:]
    seen = sorted((json.loads(line) for line in probe.stdout.splitlines() if line.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in seen] == [0, 1] and all(d["world"] == 2 and d["gpus"] == 2 for d in seen)
    assert all(d["master"].startswith("127.0.0.1:") for d in seen) and seen[0]["master"] == seen[1]["master"]
    if torch.cuda.device_count() < 2:
        real = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
        assert real.returncode != 0 and real.stdout.strip() == ""
        assert "need 2 devices" in real.stderr and "launch N > 1 with" not in real.stderr
