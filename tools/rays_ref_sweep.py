"""The oracle's four traversals against the REFERENCE's compiled Scene::intersect / Scene::occluded (oracle/_ref/ref_core; build
container only) on random scenes of 1-3 soup BVHs and 1024 rays each: random rays, axis-aligned directions (zero components: the
NaN slab products), origins on box planes and vertices, rays along edges and through vertices, far-away origins; several maxT.
Hits {t, i, j, k, primId, meshId} and occlusion flags must be identical, single and packet.  usage: rays_ref_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import prt_testlib as T
if T.ref_binary("ref_core") is None:
    sys.exit("oracle/_ref/ref_core is not built (needs /root/reference)")
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(60000 + seed)
    scene = prt_amd.Scene()
    verts = []
    for m in range(int(rng.integers(1, 4))):
        n = int(rng.integers(5, 400))
        centre = rng.uniform(-1, 1, size=(n, 1, 3))
        size = np.exp(rng.uniform(np.log(0.02), np.log(0.9), size=(n, 1, 1)))
        pos = (centre + size * rng.normal(size=(n, 3, 3)))
        if seed % 4 == 0:
            pos = np.round(pos * 4) / 4  # vertices on a coarse grid: coplanar faces, shared edges, boxes with equal planes
        pos = pos.astype(np.float32).reshape(-1, 3)
        verts.append(pos)
        idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
        mat = np.array([T.make_material(diffuse=(0.5, 0.5, 0.5))], dtype=T.MATERIAL_DTYPE)
        mesh = prt_amd.Mesh.from_arrays(idx, pos, np.zeros(n, dtype=np.uint32), mat.view(prt_amd.MATERIAL_DTYPE))
        mesh.calculate_bounds()
        scene.add(mesh)
    verts = np.concatenate(verts)
    camera = prt_amd.Camera().create((0, 0, 5), (0, 0, -1), 16, 16)
    desc = T.scene_desc_from_product(scene, camera, 1.0)
    N = 1024
    org = rng.uniform(-2.5, 2.5, (N, 3))
    d = rng.normal(size=(N, 3))
    k = N // 8
    d[:k] = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))             # axis-aligned
    d[k:2 * k, rng.integers(0, 3)] = 0.0                                                    # one zero component
    org[2 * k:3 * k] = verts[rng.integers(0, len(verts), k)]                                # origins on vertices
    tgt = verts[rng.integers(0, len(verts), k)]
    d[3 * k:4 * k] = tgt - org[3 * k:4 * k]                                                 # through vertices
    a, b = verts[rng.integers(0, len(verts) // 3, k) * 3], None
    e0 = verts[(rng.integers(0, len(verts) // 3, k)) * 3 + 1]
    org[4 * k:5 * k] = a - (e0 - a) * 2.0
    d[4 * k:5 * k] = e0 - a                                                                 # along directions of edges
    org[5 * k:6 * k] *= 1e4                                                                 # far away
    d[5 * k:6 * k] = -org[5 * k:6 * k] + rng.normal(size=(k, 3))
    d[np.all(d == 0, axis=1)] = (0, 0, 1)
    if seed % 2:
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
    org, d = org.astype(np.float32), d.astype(np.float32)
    max_t = float(rng.choice([1e5, 4.0, 2.0 * 3.7 - 0.0008]))
    rs, ro1, rp, ro8 = T.ref_rays(desc, org, d, max_t)
    s = T.OracleScene(desc)
    os_, oo1 = s.intersect_single(org, d, max_t)
    op, oo8 = s.intersect_packet(org, d, max_t)
    ok = rs.tobytes() == os_.tobytes() and rp.tobytes() == op.tobytes() and (ro1 == oo1).all() and (ro8 == oo8).all()
    if not ok:
        bad += 1
        ds = int((rs.view(np.uint8).reshape(N, -1) != os_.view(np.uint8).reshape(N, -1)).any(1).sum())
        dp = int((rp.view(np.uint8).reshape(N, -1) != op.view(np.uint8).reshape(N, -1)).any(1).sum())
        print("seed", seed, "MISMATCH single", ds, "packet", dp, "occluded", int((ro1 != oo1).sum()), int((ro8 != oo8).sum()), flush=True)
    if seed % 20 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
