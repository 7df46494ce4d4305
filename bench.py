#!/usr/bin/env python3
"""bench.py -- Mray/s of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over one batch: rendering the whole image of the workload
(every pixel, every sample, every bounce) with the scene resident in HBM.  The default workload is
the configuration BASELINE.json's metric is quoted on -- "Mray/s at 1080p/64spp": config C3, a
Sponza-class scene at 1920x1080, 64 spp, depth cap 8 on one MI355X (the reference's sponza.obj is not
shipped; a seeded procedural stand-in of the same triangle count, alpha-masked textures and a bump map is
used).  With N > 1 the image's 16x16 tiles are dealt round-robin to the ranks (every rank holds
the whole scene, no data-path collective) and the only exchange is the final image gather over RCCL, inside the library
(prt_hip_gather_rccl: each rank sends the tiles it owns -- 1/N of the image -- to rank 0).

value = rays (the reference's own `raysTraced` definition: primary samples + occlusion rays + scatter rays,
path_tracer.cpp:62,219,242,276) of the whole job / wall time of the K steps, max over ranks.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

WORKLOADS = {
    # name: (setup function name, kwargs, width, height, spp, maxDepth, description)
    "c3_sponza_standin": ("setup_atrium_standin", dict(tris=262000, seed=1), 1920, 1080, 64, 8,
                          "BASELINE config 3: Sponza-class stand-in (262k tris, alpha-masked + bump textures, directional light), "
                          "1920x1080, 64 spp, depth 8"),
    "c2_bunny_standin": ("setup_bunny_standin", dict(tris=69451, seed=1), 1024, 1024, 64, 14,
                         "BASELINE config 2: Cornell box + bunny-class stand-in (69k tris) + directional light, 1024x1024, 64 spp"),
    "c1_cornell_teapot": ("setup_cornell_box", dict(teapot="fixture"), 512, 512, 16, 4,
                          "BASELINE config 1: Cornell box + teapot (the reference's teapot.obj as vertex arrays, tests/golden/teapot_mesh.npz), 512x512, 16 spp, depth 4"),
    # the two 8-GPU configurations of BASELINE.json (parity-test cases; selectable, never the default)
    "c4_sanmiguel_standin": ("setup_atrium_standin", dict(tris=2500000, seed=4), 1920, 1080, 256, 14,
                             "BASELINE config 4: San-Miguel-class stand-in (2.5 M tris, alpha-masked foliage cards, bump map, directional light), "
                             "1920x1080, 256 spp"),
    "c5_zeroday_standin": ("setup_atrium_standin", dict(tris=5000000, seed=5, emissive_fraction=0.1, light=False), 3840, 2160, 1024, 12,
                           "BASELINE config 5: Zero-Day-class stand-in (5 M tris, 10 % emissive triangles, no directional light), "
                           "3840x2160, 1024 spp, depth 12"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak


def algorithmic_bytes(st):
    """DESIGN.md "Algorithmic bytes": 32 B per node record whose box is tested, 40 B per triangle tested
    (36 B positions + 4 B remap index), 112 B per surface fetch, 16 B per bilinear tap, 12 B per pixel written."""
    return 32 * st["nBox"] + 40 * st["nTri"] + 112 * st["nHit"] + 16 * st["nTap"] + 12 * st["nPx"]


def algorithmic_split(st):
    """The algorithmic bytes by role of the frame kernel: the four traversals (32 B per box test, 40 B per triangle test, 16 B
    per alpha-test tap) and the shade passes (112 B per surface fetch, 16 B per shading tap, 12 B per pixel)."""
    names = ("trace primary packets", "trace scatter rays", "trace packet occlusion rays", "trace single occlusion rays")
    out = {n: 32 * st["modeBox"][m] + 40 * st["modeTri"][m] + 16 * st["modeTap"][m] for m, n in enumerate(names)}
    out["shade"] = 112 * st["nHit"] + 16 * (st["nTap"] - sum(st["modeTap"])) + 12 * st["nPx"]
    return out


# Measured ceiling of dependent 64-byte record gathers from beyond the L2 at this kernel's occupancy (8 waves per SIMD), by the size of
# the table gathered from: tools/rec_gather.hip, profiles/r03_rec_gather.txt (G records per second of the whole chip)
# and profiles/r04_rec_gather_sizes.txt (the sizes in between: the Infinity Cache stops helping random gathers well below its 256 MB)
GATHER_ROOF = ((16.8e6, 80.5, "a 16 MB table: inside the Infinity Cache"), (33.6e6, 68.5, "a 32 MB table"), (67.2e6, 60.6, "a 64 MB table"),
               (134.3e6, 57.2, "a 128 MB table"), (268.5e6, 55.4, "a 256 MB table"), (float("inf"), 52.9, "a 2 GB table: DRAM row activations"))


def binding_roof(ev, kernel_ms, tree_bytes):
    """The roofs that can bind the frame kernel, each as a MEASURED fraction <= 1 from the committed counter summary `ev`,
    and the one that binds = the largest.  The contract's algorithmic figure (bytes of the reference's events / time / 8 TB/s) is not
    among them: most of those bytes are served by LDS, L1 and L2, and it exceeds 1 on the headline configuration."""
    d, sec = ev["derived"], kernel_ms * 1e-3
    raw = ev["counters_raw"]
    # everything below is per second of THIS run's kernel time: the counters are event counts of one frame (deterministic work)
    clock = raw["GRBM_GUI_ACTIVE"] / 8 / (ev.get("kernel_ms_under_profiler_median", kernel_ms) * 1e-3)
    valu_busy = raw["SQ_INSTS_VALU"] * 3.07 / 1024.0 / (clock * sec)
    lanes = d["lanes_active_per_valu_instruction"]
    miss_rate = raw["TCC_MISS_sum"] / sec / 1e9
    roof_g, roof_note = next((g, n) for lim, g, n in GATHER_ROOF if tree_bytes <= lim)
    roofs = {
        "valu_useful": {"frac": min(1.0, valu_busy) * lanes / 64.0,
                        "formula": "SQ_INSTS_VALU x 3.07 cycles / 1024 SIMDs / frame cycles (vector ALU busy) x SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64 (lanes that do work)",
                        "valu_busy_frac": valu_busy, "lanes_per_valu": lanes},
        "l2_miss_rate": {"frac": miss_rate / roof_g, "formula": f"TCC_MISS_sum / kernel time / {roof_g} G per second (dependent 64-byte gathers from {roof_note}; profiles/r03_rec_gather.txt)",
                         "achieved_G_per_s": miss_rate, "roof_G_per_s": roof_g, "l2_misses_per_ray": d["l2_misses_per_ray"]},
        "hbm_bytes": {"frac": ev["traffic_bytes_per_frame"] / sec / 1e9 / HBM_PEAK_GBS, "formula": "(FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s",
                      "achieved_GBps": ev["traffic_bytes_per_frame"] / sec / 1e9},
    }
    name = max(roofs, key=lambda k: roofs[k]["frac"])
    return name, roofs


def counter_evidence(workload, world, overridden):
    """The committed counter summary of this workload (profiles/*_counters.json: separate rocprofv3 --pmc passes, one counter
    block each, summarised by tools/pmc_evidence.py), or None.  It is only used when it was taken with the kernel sources this
    process runs (source_sha16) on the workload's own size: the HBM-side bytes of a launch cannot be measured from inside
    the process, and a stale file must not pass for a measurement."""
    if world != 1 or overridden:
        return None
    import glob
    import prt_amd
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("source_sha16") == prt_amd.loaded_source_sha16() and "rank" not in d.get("frame", ""):
            d["file"] = "profiles/" + os.path.basename(f)
            best = d
    return best


def host_cores():
    """Cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box hands a
    1-GPU job a share of the host, not all of os.cpu_count())."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("PRT_BENCH_CPU_THREADS")
    return int(env) if env else n


def cpu_baseline(scene, camera, exposure, spp, max_depth, seed, budget_seconds, gpu_render_cap14):
    """The CPU path timed on this box's host cores (all of them).

    kind "reference+glue": the reference's own compiled code (oracle/_ref/ref_path: the reference's path_tracer / bvh / scene /
    camera / triangle / light objects, AVX2 8-wide, built from its sources by oracle/Makefile, linked with oracle/ref_glue.cpp
    for the four translation units that cannot be built here -- surface fetch and texture taps are the oracle's scalar C)
    on the WHOLE frame of the workload, what the reference's main.cpp:120-183 times.  Its depth cap is the literal 14
    (path_tracer.cpp:124), so the GPU renders the same whole frame once more at cap 14 (`gpu_render_cap14`): ONE ratio, identical
    pixels, identical cap -- and the two images are compared bit for bit while they are there.  Rays are the GPU launch's count
    (deterministic; the parity suite requires it equal to the oracle's and the compiled reference's).  Only when a probe says
    the whole frame would not fit `budget_seconds` is a centred window traced instead (both sides on the window).
    kind "port": oracle/prt_oracle.c (scalar C, OpenMP over tiles) on a window, when the reference binary is not there."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import prt_testlib as T
    desc = T.scene_desc_from_product(scene, camera, exposure)
    W, H = camera.width, camera.height
    cores = host_cores()

    def window(frac):
        w, h = max(16, int(W * frac) // 16 * 16), max(16, int(H * frac) // 16 * 16)
        x0, y0 = (W - w) // 2 // 16 * 16, (H - h) // 2 // 16 * 16
        return x0, y0, min(W - 1, x0 + w - 1), min(H - 1, y0 + h - 1)

    if T.ref_binary("ref_path") is None:
        s = T.OracleScene(desc)

        def run_port(rect, stats):
            t0 = time.time()
            _, st = s.render_rect(rect, spp, max_depth=max_depth, seed=seed, threads=cores, stats=stats)
            return time.time() - t0, st["raysTraced"]

        probe = window(0.06)
        sec, _ = run_port(probe, False)
        rect = window(min(1.0, 0.06 * (min(budget_seconds, 30.0) / max(sec, 1e-3)) ** 0.5))
        _, rays = run_port(rect, True)        # counting pass (not timed)
        port_sec, _ = run_port(rect, False)   # timed pass without counters
        return dict(value=rays / port_sec / 1e6, unit="Mray/s", cores=cores, kind="port",
                    sample=f"window x{rect[0]}..{rect[2]} y{rect[1]}..{rect[3]} of the {W}x{H} image at {spp} spp, depth cap {max_depth}: "
                           f"{rays} rays in {port_sec:.2f} s on {cores} threads (oracle port; oracle/_ref/ref_path is not built)")
    probe = window(0.08)
    _, pst = T.ref_render(desc, spp, probe, seed=seed, threads=cores, stats=False)
    px = lambda r: (r[2] - r[0] + 1) * (r[3] - r[1] + 1)  # noqa: E731
    est = pst["seconds"] * W * H / px(probe)
    rect = (0, 0, W - 1, H - 1) if est <= budget_seconds else window(min(1.0, 0.08 * (budget_seconds / max(pst["seconds"], 1e-3)) ** 0.5))
    whole = rect == (0, 0, W - 1, H - 1)
    cpu_img, rst = T.ref_render(desc, spp, rect, seed=seed, threads=cores, stats=False)
    gpu_img, gst = gpu_render_cap14(rect)
    same = cpu_img.view(np.uint32) == gpu_img.view(np.uint32)
    both_nan = np.isnan(cpu_img) & np.isnan(gpu_img)  # (a NaN's sign and payload differ between SSE and the GPU)
    differing = int((~(same | both_nan)).any(axis=2).sum())
    rays14 = gst["raysTraced"]
    cpu_rate, gpu_rate = rays14 / rst["seconds"] / 1e6, rays14 / gst["kernelMs"] / 1e3
    where = f"the whole {W}x{H} frame" if whole else f"window x{rect[0]}..{rect[2]} y{rect[1]}..{rect[3]} of the {W}x{H} image (the whole frame was estimated at {est:.0f} s, budget {budget_seconds:.0f} s)"
    return dict(value=cpu_rate, unit="Mray/s", cores=cores, kind="reference+glue",
                sample=f"{where} at {spp} spp, depth cap 14 (the reference's literal): {rays14} rays in {rst['seconds']:.2f} s on {cores} threads by "
                       "oracle/_ref/ref_path (the reference's compiled path_tracer/bvh/scene/camera/triangle objects + oracle/ref_glue.cpp for surface fetch and texture taps)",
                same_cap={"max_depth": 14, "pixels": px(rect), "rays": int(rays14), "gpu_value": gpu_rate, "gpu_ms": gst["kernelMs"], "cpu_value": cpu_rate,
                          "cpu_seconds": rst["seconds"], "gpu_over_cpu": gpu_rate / cpu_rate,
                          "note": "GPU and CPU trace the same pixels at the same depth cap with the same per-pixel seeds"},
                image_check=("GPU image == compiled reference's image, bit for bit" if differing == 0 else f"MISMATCH: {differing} pixels differ") +
                            f" ({px(rect)} pixels)")


def spawn_ranks(n, argv):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <argv>` as a child and wait for it: one
    process per GPU, rendezvous on 127.0.0.1 at a port that is free now.  Returns the child's exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3_sponza_standin", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--max-depth", type=int, default=0)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=120.0, help="budget of the CPU leg: the whole frame when a probe says it fits, else a window")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  A CHILD process (never an exec: this pool
        # forbids replacing a process, and nothing here has touched the GPU yet anyway); its stdout -- rank 0's one JSON line --
        # passes through and its exit code is ours.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout carries ONE JSON line: whatever libraries print there while the run lasts (RCCL's version banner, for one) goes to
    # stderr, and the line is written to the real stdout at the end
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (before anything initialises HIP: RCCL needs dmabuf IPC on this pool)
    import torch
    import torch.distributed as dist
    import prt_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world  # (a launcher's world size wins over the flag)
    if os.environ.get("PRT_BENCH_LAUNCH_PROBE"):
        # the launcher test (tests/test_multi_rank_cpu.py): say what this rank saw and stop before a device is needed
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "gpus": args.gpus,
                          "master": os.environ.get("MASTER_ADDR", "") + ":" + os.environ.get("MASTER_PORT", "")}), file=real_stdout, flush=True)
        return
    # (device_count() does not initialise the GPU on this image; is_available() does)
    have = torch.cuda.device_count()
    if have < world or local_rank >= have:
        raise SystemExit(f"bench.py: need {world} devices for --gpus {world}, this node shows {have}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run the RCCL path is exercised even with one rank
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))  # RCCL on ROCm

    setup, kw, W, H, spp, depth, describe = WORKLOADS[args.workload]
    W, H = args.width or W, args.height or H
    spp, depth = args.spp or spp, args.max_depth or depth
    # the library travels prebuilt; should it need a rebuild, one rank does it and the others wait
    if not use_dist or local_rank == 0:
        prt_amd.build()
    if use_dist:
        dist.barrier()
    kw = dict(kw)
    if kw.pop("teapot", None) == "fixture":  # C1's teapot: the committed vertex arrays of the reference's teapot.obj (main.cpp:28-50)
        z = np.load(os.path.join(ROOT, "tests", "golden", "teapot_mesh.npz"))
        mat = prt_amd.Material.make(diffuse=(0.9, 0.9, 0.9), reflection=prt_amd.Material.SPECULAR)
        idx = np.ascontiguousarray(z["indices"], dtype=np.uint32).reshape(-1, 3)
        teapot = prt_amd.Mesh.from_arrays(idx, z["positions"].astype(np.float32), np.zeros(len(idx), dtype=np.uint32),
                                          np.frombuffer(bytes(mat), dtype=prt_amd.MATERIAL_DTYPE), texcoords=z["texcoords"])
        teapot.transform(0.005, (-0.5, 0.0, 0.5))
        kw["teapot_mesh"] = teapot
    scene, camera, exposure = getattr(prt_amd, setup)(W, H, **kw)
    tracer = prt_amd.PathTracer(device=local_rank, max_depth=depth, seed=args.seed)
    tracer.upload_scene(scene)
    tracer.set_camera(camera)

    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")  # this rank's framebuffer (rank 0's receives the gathered image)
    if use_dist:
        # the library's own RCCL communicator: rank 0 makes the id, its 128 bytes travel over torch.distributed
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(prt_amd.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        tracer.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
    # a non-default torch stream: the kernel is launched on it through the C-ABI, the RCCL gather is ordered behind
    # it, and the C-ABI's HIP events are recorded on it
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def step():
        tracer.render_async(0, 0, W - 1, H - 1, spp, d_rgb=fb.data_ptr(), stream=stream, exposure=exposure, rank=rank, nranks=world)
        if use_dist:
            # the only exchange: every rank's owned tiles (25 MB / N at 1080p) to rank 0 over RCCL, de-interleaved there
            tracer.gather_rccl(d_rgb=fb.data_ptr(), root=0, stream=stream)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    tracer.stats()  # drop the warm-up launches' events
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    st = tracer.stats()
    t_all = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rays = torch.tensor([st["raysTraced"], st["occludedTraced"]], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(t_all.item())
    rays_per_step, occl_per_step = float(rays[0].item()), float(rays[1].item())
    kernel_ms = st["kernelMsSum"] / max(1, st["kernelLaunches"])

    out = None
    gather_check = None
    if rank == 0 and world > 1:
        # untimed: the image the ranks' tiles were gathered into must equal what one GPU renders alone, bit for bit
        gathered = fb.clone()
        alone = torch.zeros_like(fb)
        tracer.render_async(0, 0, W - 1, H - 1, spp, d_rgb=alone.data_ptr(), stream=stream, exposure=exposure, rank=0, nranks=1)
        torch.cuda.synchronize()
        tracer.stats()
        same = bool(torch.equal(gathered.view(torch.int32), alone.view(torch.int32)))
        gather_check = f"the image gathered from {world} ranks equals the one-GPU image bit for bit" if same else \
            f"MISMATCH: {int((gathered.view(torch.int32) != alone.view(torch.int32)).any(dim=2).sum().item())} pixels of the gathered image differ from the one-GPU image"
        del gathered, alone
    if rank == 0:
        # algorithmic bytes of THIS rank's launch from an untimed counting launch (deterministic event counts)
        tracer.render_async(0, 0, W - 1, H - 1, spp, d_rgb=fb.data_ptr(), stream=stream, exposure=exposure, rank=rank, nranks=world,
                            count_traffic=True)
        torch.cuda.synchronize()
        ct = tracer.stats()
        B = algorithmic_bytes(ct)
        achieved = B / (kernel_ms * 1e-3) / 1e9
        ev = counter_evidence(args.workload, world, bool(args.spp or args.width or args.height or args.max_depth))
        traffic_bytes = ev["traffic_bytes_per_frame"] if ev else None
        traffic_rate = traffic_bytes / (kernel_ms * 1e-3) / 1e9 if ev else None
        value = rays_per_step * args.steps / dt / 1e6
        name, cus = tracer.device_info()
        # bytes of the tree the traversals gather from (64-byte records of the internal nodes -- (nodes - 1) / 2 of a binary tree -- and
        # 36-byte triangle slots, ~1.12 slots per triangle with the padded leaf blocks): which of the measured gather ceilings applies
        sd = scene.describe().contents
        tree_bytes = sum((sd.meshes[i].nodeCount - 1) // 2 * 64 + sd.meshes[i].primCount * 36 * 1.12 for i in range(sd.meshCount))
        bind, roofs = binding_roof(ev, kernel_ms, tree_bytes) if ev else (None, None)
        out = {
            "metric": "Mray/s (primary+secondary) at 1080p/64spp; 1/2/4/8-GPU scaling",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "description": describe, "width": W, "height": H, "spp": spp, "max_depth": depth,
                       "seed": args.seed, "rays_per_step": int(rays_per_step), "occlusion_rays_per_step": int(occl_per_step),
                       "sharding": f"16x16 tiles round-robin over {world} rank(s), scene replicated, RCCL send/recv image gather of {tracer.gather_payload_bytes()} B per rank" if world > 1
                       else "one GPU", "device": name, "compute_units": cus, **({"gather_check": gather_check} if gather_check else {})},
            # `achieved` / `frac` are the contract's ALGORITHMIC figure (layout-independent bytes of the events the
            # reference's algorithm performs, most of them served by L1/L2) -- not a claim that DRAM is busy.  What DRAM and the
            # memory pipeline really do is beside it: `traffic` / `hbm_frac` (fabric-side counters) and `limiter` (from the same file).
            # `frac` = the roof that BINDS, a measured fraction <= 1 named by `frac_of` (the largest of `roofs`: useful share of the vector
            # ALU, L2-miss rate against the measured gather ceiling, fabric bytes against 8 TB/s); `achieved` / `algorithmic_frac` keep
            # SURVEY 8d's contract figure, which is not a fraction of anything (it exceeds 1 on this configuration)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": roofs[bind]["frac"] if ev else None, "frac_of": (bind + ": " + roofs[bind]["formula"]) if ev else None, "roofs": roofs,
                         "tree_bytes": int(tree_bytes),
                         "algorithmic_frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic_rate, "hbm_frac": (traffic_rate / HBM_PEAK_GBS) if traffic_rate else None,
                         "traffic_bytes_per_launch": traffic_bytes, "traffic_source": ev["file"] if ev else None,
                         "traffic_source_sha16": ev["source_sha16"] if ev else None, "source_sha16": prt_amd.loaded_source_sha16(),
                         # what the same counter file says about the two other roofs of this kernel: the vector ALU ...
                         "valu_busy_frac": ev["derived"].get("valu_busy_frac (SQ_INSTS_VALU x 3.07 cycles / 1024 SIMDs / frame cycles)") if ev else None,
                         "lanes_per_valu": ev["derived"]["lanes_active_per_valu_instruction"] if ev else None,
                         # ... and the rate of L2 misses, the ceiling of trees larger than the caches (dependent 64-byte gathers from beyond
                         # the caches: 53-56 G per second on this chip, a whole 128-byte line for the price of one sector:
                         # profiles/r03_rec_gather.txt; the large scenes run at 50-52 G misses per second)
                         "l2_misses_per_ray": ev["derived"]["l2_misses_per_ray"] if ev else None,
                         "limiter": ("dependent record gathers (a node or a leaf per round trip) at 8 waves per SIMD, with the vector ALU as the second roof: waves wait "
                                     f"{ev['derived']['wave_time_waiting_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)']:.2f} of their time, the vector ALU is busy "
                                     f"{ev['derived'].get('valu_busy_frac (SQ_INSTS_VALU x 3.07 cycles / 1024 SIMDs / frame cycles)', float('nan')):.2f} of the frame at "
                                     f"{ev['derived']['lanes_active_per_valu_instruction']:.1f} of 64 lanes per vector instruction, the gather path "
                                     f"{ev['derived']['ta_busy_frac (TA_BUSY_avr / cycles of the frame)']:.2f}; L1 hit {ev['derived']['l1_hit_rate']:.2f}, L2 hit "
                                     f"{ev['derived']['l2_hit_rate']:.2f}, {ev['derived']['l2_misses_per_ray']:.1f} L2 misses per ray; `algorithmic_frac` counts the bytes of the "
                                     "reference's events, most of them served by LDS / L1 / L2 -- it may exceed 1; `frac` is the binding roof of `roofs`") if ev
                         else "see profiles/ (no counter summary taken with these kernel sources)",
                         "kernel": "frame_kernel: one persistent launch per frame (shade passes + four ray traversals as roles of its waves)",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": int(B),
                         "events_per_launch": {k: ct[k] for k in ("nBox", "nTri", "nHit", "nTap", "nPx")},
                         "algorithmic_bytes_by_role": algorithmic_split(ct)},
        }
        if not args.no_cpu_baseline and world == 1:
            def gpu_render_cap14(rect):
                tracer.max_depth = 14
                try:
                    img = tracer.trace_block(*rect, spp, exposure=exposure)
                    return img, tracer.last_stats
                finally:
                    tracer.max_depth = depth
            try:
                out["cpu_baseline"] = cpu_baseline(scene, camera, exposure, spp, depth, args.seed, args.cpu_seconds, gpu_render_cap14)
            except Exception as e:  # the baseline is a report, never the product
                out["cpu_baseline"] = {"value": None, "unit": "Mray/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), file=real_stdout, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    tracer.close()


if __name__ == "__main__":
    main()
