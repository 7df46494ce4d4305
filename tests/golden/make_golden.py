#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the COMPILED REFERENCE.

Run in the build container only (needs /root/reference and `make -C oracle`):
    python tests/golden/make_golden.py

Outputs (all small, data only -- no reference source text):
  cornell_box.npz      Cornell box arrays as SampleModels::getCornellBox(true) produces them
                       (dumped from the reference's compiled sample_models.cpp via `ref_path cornell`)
  teapot_mesh.npz      the reference's data/teapot/teapot.obj as arrays (fan triangulation 0-1-2 / 0-2-3;
                       tinyobjloader, whose rule the reference would use, is an empty submodule here)
  leaf_vectors.npz     ray/triangle/box inputs + outputs of the reference's intersectTriangle (SoA with both
                       kinds of swap flags, and scalar), the four BBox::intersect overloads, SoaRay/Ray::prepare
  bvh_cornell_teapot.npz  flattened BVH arrays of both meshes, scene bbox, radius (reference BvhBuildNode::build)
  rays_cornell_teapot.npz 4096 rays -> reference Scene::intersect / Scene::occluded, single and packet
  camera_packets.npz   Camera::GenerateJitteredRayPacket + Random for a few pixels/states
  radiance_c1_crop.npz per-pixel radiance of a 64x64 crop of config C1 (512x512, 16 spp, depth cap 14) and of
                       Cornell-only 128x128x16spp, with ray counts (reference PathTracer::TraceBlock, hybrid link:
                       see oracle/ref_glue.cpp for which leaf functions are the oracle's)
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # the repository root: prt_amd's host code builds the stand-in scenes
import prt_testlib as T  # noqa: E402

REF = "/root/reference"


def dump_cornell():
    with tempfile.TemporaryDirectory() as td:
        op = os.path.join(td, "c.bin")
        T.run_ref("ref_path", "cornell", op)
        buf = open(op, "rb").read()
    pc, vc, mc = struct.unpack_from("<3I", buf, 0)
    off = 12
    idx = np.frombuffer(buf, "<u4", pc * 3, off).reshape(pc, 3).copy(); off += pc * 12
    pos = np.frombuffer(buf, "<f4", vc * 3, off).reshape(vc, 3).copy(); off += vc * 12
    pm = np.frombuffer(buf, "<u4", pc, off).copy(); off += pc * 4
    mats = np.frombuffer(buf, T.MATERIAL_DTYPE, mc, off).copy()
    np.savez_compressed(os.path.join(HERE, "cornell_box.npz"), indices=idx, positions=pos, prim_material=pm,
                        materials=mats.view(np.uint8).reshape(mc, -1))
    print("cornell: prims", pc, "verts", vc, "materials", mc)


def dump_teapot():
    pos, tex, faces = [], [], []
    for line in open(os.path.join(REF, "data/teapot/teapot.obj")):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            pos.append([float(x) for x in p[1:4]])
        elif p[0] == "vt":
            tex.append([float(x) for x in p[1:3]])
        elif p[0] == "f":
            faces.append(p[1:])
    pos = np.asarray(pos, dtype=np.float64).astype(np.float32)
    tex = np.asarray(tex, dtype=np.float64).astype(np.float32)
    vtex = np.zeros((len(pos), 2), dtype=np.float32)
    tris = []

    def corner(tok):
        a = tok.split("/")
        v = int(a[0])
        v = v - 1 if v > 0 else len(pos) + v
        t = -1
        if len(a) > 1 and a[1]:
            t = int(a[1])
            t = t - 1 if t > 0 else len(tex) + t
        return v, t

    for f in faces:
        c = [corner(t) for t in f]
        for k in range(1, len(c) - 1):
            for v, t in (c[0], c[k], c[k + 1]):
                # mesh.cpp:272-286: per-vertex texcoord, last writer wins, v flipped
                vtex[v] = (tex[t][0], np.float32(1.0) - tex[t][1]) if t >= 0 else (0.0, 0.0)
            tris.append([c[0][0], c[k][0], c[k + 1][0]])
    tris = np.asarray(tris, dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "teapot_mesh.npz"), positions=pos, indices=tris, texcoords=vtex)
    print("teapot: verts", len(pos), "tris", len(tris))


def leaf_inputs(rng, n):
    """Random + adversarial ray/triangle/box records (22 floats each)."""
    rec = np.zeros((n, 22), dtype=np.float32)
    org = rng.uniform(-2, 2, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tri = rng.uniform(-1.5, 1.5, (n, 3, 3))
    lo = rng.uniform(-1.5, 1.0, (n, 3))
    hi = lo + rng.uniform(0.0, 1.5, (n, 3))
    maxT = rng.uniform(0.5, 10.0, n)
    q = n // 8
    # aim a share of the rays at the triangle interior / edges / vertices
    w = rng.dirichlet([1, 1, 1], q)
    tgt = (tri[:q] * w[:, :, None]).sum(1)
    d[:q] = tgt - org[:q]
    d[:q] /= np.linalg.norm(d[:q], axis=1, keepdims=True)
    e = rng.uniform(0, 1, q)
    tgt = tri[q:2 * q, 0] * e[:, None] + tri[q:2 * q, 1] * (1 - e[:, None])  # on an edge
    d[q:2 * q] = tgt - org[q:2 * q]
    d[q:2 * q] /= np.linalg.norm(d[q:2 * q], axis=1, keepdims=True)
    d[2 * q:3 * q] = tri[2 * q:3 * q, 2] - org[2 * q:3 * q]  # through a vertex
    d[2 * q:3 * q] /= np.linalg.norm(d[2 * q:3 * q], axis=1, keepdims=True)
    # axis-aligned directions (zero components -> inf invDir), origins on slab planes (0*inf = NaN)
    ax = rng.integers(0, 3, q)
    d[3 * q:4 * q] = 0
    d[3 * q:4 * q][np.arange(q), ax] = rng.choice([-1.0, 1.0], q)
    k = rng.integers(0, 3, q)
    org[3 * q:4 * q][np.arange(q), (ax + 1) % 3] = np.where(k == 0, lo[3 * q:4 * q][np.arange(q), (ax + 1) % 3],
                                                            np.where(k == 1, hi[3 * q:4 * q][np.arange(q), (ax + 1) % 3],
                                                                     org[3 * q:4 * q][np.arange(q), (ax + 1) % 3]))
    # two equal components (tie rules of the swap selection), negative-dominant directions
    d[4 * q:5 * q, 1] = d[4 * q:5 * q, 0]
    d[5 * q:6 * q] = -np.abs(d[5 * q:6 * q])
    # degenerate triangles (det == 0) and rays starting inside the box
    tri[6 * q:6 * q + q // 2, 2] = tri[6 * q:6 * q + q // 2, 1]
    org[6 * q + q // 2:7 * q] = (lo[6 * q + q // 2:7 * q] + hi[6 * q + q // 2:7 * q]) * 0.5
    rec[:, 0:3] = org
    rec[:, 3:6] = d
    rec[:, 6:15] = tri.reshape(n, 9)
    rec[:, 15:18] = lo
    rec[:, 18:21] = hi
    rec[:, 21] = maxT
    # the reference's own known answer (tests/tests.cpp:109-127) as record 0
    rec[0, 0:15] = [0.4, 0.4, -1.0, 0, 0, 1.0, 0, 0, 0, 1, 0, 0, 0, 1, 0]
    return rec


def dump_leaf():
    rng = np.random.default_rng(20261003)
    rec = leaf_inputs(rng, 8192)
    with tempfile.TemporaryDirectory() as td:
        ip, op = os.path.join(td, "i.bin"), os.path.join(td, "o.bin")
        rec.astype("<f4").tofile(ip)
        T.run_ref("ref_core", "leaf", ip, op)
        out = np.fromfile(op, "<f4").reshape(-1, 24)
    assert out[0, 0] == 1.0, out[0]  # tests.cpp:126: t == 1.0f exactly
    np.savez_compressed(os.path.join(HERE, "leaf_vectors.npz"), inputs=rec, outputs=out)
    print("leaf vectors:", len(rec), "tri hits", int((out[:, 0] != -1).sum()), "box hits", int(out[:, 13].sum()))


def scene_rays(rng, desc, n):
    """Rays through the Cornell+teapot scene: camera-like, interior scatter, grazing, teapot-aimed."""
    org = np.zeros((n, 3), dtype=np.float32)
    d = rng.normal(size=(n, 3))
    q = n // 4
    org[:q] = (0, 0.965, 2.6)
    d[:q] = np.stack([rng.uniform(-0.5, 0.5, q), rng.uniform(-0.5, 0.5, q), -np.ones(q)], 1)
    org[q:] = rng.uniform([-0.95, 0.05, -0.95], [0.95, 1.9, 0.95], (n - q, 3))
    tp = desc.meshes[1].positions
    tgt = tp[rng.integers(0, len(tp), q)]
    d[2 * q:3 * q] = tgt - org[2 * q:3 * q]
    d[3 * q:3 * q + q // 2, 1] *= 0.01  # grazing the floor/ceiling
    d[3 * q + q // 2:, rng.integers(0, 3)] = 0.0  # an exactly-zero component
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return org.astype(np.float32), d.astype(np.float32)


def dump_scene_vectors():
    desc = T.cornell_scene(512, 512, with_teapot=True)
    bv, sbox, radius = T.ref_bvh(desc)
    bt, _, _ = T.ref_bvh(desc, threaded=True)
    for a, b in zip(bv, bt):  # topology must not depend on the build's thread pool (bvh.cpp:157-163)
        assert a["nodes"].tobytes() == b["nodes"].tobytes() and (a["remap"] == b["remap"]).all()
    np.savez_compressed(os.path.join(HERE, "bvh_cornell_teapot.npz"),
                        nodes0=bv[0]["nodes"].view(np.uint8), remap0=bv[0]["remap"], bbox0=bv[0]["bbox"],
                        nodes1=bv[1]["nodes"].view(np.uint8), remap1=bv[1]["remap"], bbox1=bv[1]["bbox"],
                        leaf_counts=np.array([bv[0]["leaf_count"], bv[1]["leaf_count"]]), scene_bbox=sbox,
                        radius=np.float32(radius), teapot_normals=desc.meshes[1].normals)
    print("bvh: cornell", len(bv[0]["nodes"]), "nodes /", bv[0]["leaf_count"], "leaves; teapot", len(bv[1]["nodes"]),
          "nodes /", bv[1]["leaf_count"], "leaves; radius", radius)

    rng = np.random.default_rng(7)
    org, d = scene_rays(rng, desc, 4096)
    far = np.float32(2.0) * np.float32(radius)
    single, occ1, packet, occ8 = T.ref_rays(desc, org, d, far)
    np.savez_compressed(os.path.join(HERE, "rays_cornell_teapot.npz"), org=org, dir=d, max_t=far,
                        single=single.view(np.uint8), occluded_single=occ1, packet=packet.view(np.uint8),
                        occluded_packet=occ8)
    print("rays: single hits", int((single["t"] != -1).sum()), "packet hits", int((packet["t"] != -1).sum()),
          "single!=packet", int((single["t"] != packet["t"]).sum()), "occluded", int(occ1.sum()), int(occ8.sum()))

    with tempfile.TemporaryDirectory() as td:
        sp = os.path.join(td, "s.prts")
        desc.write_prts(sp)
        recs = []
        for (x, y, state) in [(0, 0, 1), (255, 255, 0x9E3779B9), (511, 0, 12345), (17, 400, 0xFFFFFFFF), (300, 511, 7)]:
            op = os.path.join(td, "o.bin")
            T.run_ref("ref_core", "camera", sp, x, y, state, op)
            recs.append((x, y, state, np.fromfile(op, "<f4")))
    np.savez_compressed(os.path.join(HERE, "camera_packets.npz"),
                        xys=np.array([(r[0], r[1], r[2]) for r in recs], dtype=np.uint32),
                        out=np.stack([r[3] for r in recs]))
    print("camera packets:", len(recs))


def dump_radiance():
    # C1 crop: 64x64 window over the teapot/short-box region, 16 spp, depth cap 14 (the reference's literal)
    desc = T.cornell_scene(512, 512, with_teapot=True)
    rect = (160, 300, 223, 363)
    rgb, st = T.ref_render(desc, 16, rect, seed=12345, stats=True)
    d2 = T.cornell_scene(128, 128, with_teapot=False)
    rgb2, st2 = T.ref_render(d2, 16, (0, 0, 127, 127), seed=12345, stats=True)
    print("C1 crop mean", rgb.reshape(-1, 3).mean(0), st)
    print("cornell-only 128^2 mean", rgb2.reshape(-1, 3).astype(np.float64).mean(0), st2)
    np.savez_compressed(os.path.join(HERE, "radiance_c1_crop.npz"), rect=np.array(rect), rgb=rgb,
                        rays=np.array([st["raysTraced"], st["occludedTraced"]], dtype=np.uint64),
                        cornell_only_rgb=rgb2,
                        cornell_only_rays=np.array([st2["raysTraced"], st2["occludedTraced"]], dtype=np.uint64))


def dump_env_light():
    # InfiniteAreaLight::create + sample of the compiled reference (light.cpp:30-128) on two seeded maps, and a Cornell-only
    # render lit by the first (pins the two extra generator draws per diffuse bounce, path_tracer.cpp:164-167)
    out = {}
    for name, (w, h, black) in {"sky": (64, 32, False), "black_rows": (48, 24, True)}.items():
        desc = T.cornell_scene(96, 96, with_teapot=False)
        desc.env = T.sky_env(w, h, black_rows=black)
        u = T.env_test_u(4096)
        vp, hp, d, c = T.ref_envlight(desc, u)
        out.update({f"{name}_size": np.array([w, h, int(black)]), f"{name}_vertical": vp, f"{name}_horizontal": hp,
                    f"{name}_dir": d, f"{name}_color": c})
    desc = T.cornell_scene(96, 96, with_teapot=False)
    desc.env = T.sky_env(64, 32)
    rect = (16, 16, 79, 79)
    rgb, st = T.ref_render(desc, 16, rect, seed=12345, stats=True)
    print("env-lit cornell crop mean", rgb.reshape(-1, 3).mean(0), st)
    np.savez_compressed(os.path.join(HERE, "env_light.npz"), rect=np.array(rect), rgb=rgb,
                        rays=np.array([st["raysTraced"], st["occludedTraced"]], dtype=np.uint64), **out)


def dump_gbuffer():
    # GbufferVisualizer::TraceBlock (gbuffer_visualizer.cpp:17-51, Camera::GenerateJitteredRay camera.cpp:12-33) of the compiled
    # reference on Cornell + teapot: diffuse colour and bump-mapped normal images
    desc = T.cornell_scene(96, 96, with_teapot=True)
    rect = (8, 8, 87, 87)
    out = {f"kind{k}": T.ref_gbuffer(desc, k, rect) for k in (0, 1, 2)}
    np.savez_compressed(os.path.join(HERE, "gbuffer.npz"), rect=np.array(rect), **out)
    print("gbuffer means", {k: float(v.mean()) for k, v in out.items()})


def dump_c1_checksums():
    # BASELINE config 1 in full (Cornell + teapot, 512x512, 16 spp, the reference's depth literal 14) through the compiled
    # reference: digests of the float image instead of the 3 MB image itself
    import hashlib
    desc = T.cornell_scene(512, 512, with_teapot=True)
    rgb, st = T.ref_render(desc, 16, (0, 0, 511, 511), seed=12345, stats=True)
    f64 = rgb.reshape(-1, 3).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "c1_full_checksums.npz"), sum=f64.sum(0), sumsq=(f64 * f64).sum(0),
                        sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(rgb, dtype="<f4").tobytes()).digest(), dtype=np.uint8),
                        rays=np.array([st["raysTraced"], st["occludedTraced"]], dtype=np.uint64),
                        row_sums=rgb.astype(np.float64).sum(axis=(1, 2)))
    print("C1 full image: sum", f64.sum(0), "rays", st)


def dump_scene_digests():
    # Whole images of a C2-class scene (Cornell + displaced-sphere stand-in + directional light: packet and single occlusion
    # rays) and a C3-class scene (textured atrium: alpha masks, bump map, vertex normals, depth 14) through the compiled
    # reference (surface / texture members forwarded to the oracle by ref_glue.cpp).  Scenes come from the product's host code.
    import hashlib
    import prt_amd
    out = {}
    for name, (setup, kw, w, h) in {"c2": ("setup_bunny_standin", dict(tris=20000), 192, 192),
                                    "c3": ("setup_atrium_standin", dict(tris=40000), 192, 108)}.items():
        scene, camera, exposure = getattr(prt_amd, setup)(w, h, **kw)
        desc = T.scene_desc_from_product(scene, camera, exposure)
        rgb, st = T.ref_render(desc, 16, (0, 0, w - 1, h - 1), seed=12345, stats=True)
        out[f"{name}_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(rgb, dtype="<f4").tobytes()).digest(), dtype=np.uint8)
        out[f"{name}_rays"] = np.array([st["raysTraced"], st["occludedTraced"]], dtype=np.uint64)
        out[f"{name}_row_sums"] = rgb.astype(np.float64).sum(axis=(1, 2))
        print(name, "mean", rgb.reshape(-1, 3).mean(0), st)
    np.savez_compressed(os.path.join(HERE, "scene_digests.npz"), **out)


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", T.ORACLE_DIR])
    dump_cornell()
    dump_teapot()
    dump_leaf()
    dump_scene_vectors()
    dump_radiance()
    dump_env_light()
    dump_gbuffer()
    dump_c1_checksums()
    dump_scene_digests()
