"""The host BVH builder against the REFERENCE's own compiled Bvh::build (oracle/_ref/ref_core, built from /root/reference by
oracle/Makefile -- so this runs in the build container only) over the soup shapes of tools/bvh_sweep.py: node arrays, leaf order and
primRemapping must be identical.  Together with bvh_sweep.py (GPU builder == host builder) this pins the GPU builder to the
reference beyond the Cornell / teapot fixture.  usage: bvh_ref_sweep.py FIRST LAST"""
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
    rng = np.random.default_rng(40000 + seed)
    n = int(np.exp(rng.uniform(0, np.log(20000))))
    kind = int(rng.integers(0, 8))
    if kind == 0:
        pos = rng.uniform(-1, 1, (3 * n, 3))
    elif kind == 1:
        c = rng.uniform(-10, 10, (int(rng.integers(1, 6)), 3))
        pos = c[rng.integers(0, len(c), 3 * n)] + rng.normal(size=(3 * n, 3)) * np.exp(rng.uniform(-6, 0, (3 * n, 1)))
    elif kind == 2:
        g = int(np.ceil(n ** 0.5))
        q = np.stack(np.meshgrid(np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 2)[:n]
        base = np.concatenate([q, np.zeros((n, 1))], 1)
        pos = (base[:, None, :] + np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]])[None]).reshape(-1, 3)
    elif kind == 3:
        pos = rng.uniform(-1, 1, (3 * n, 3))
        pos[:, int(rng.integers(0, 3))] = rng.uniform(-1, 1)
    elif kind == 4:
        t = rng.uniform(-1, 1, (n, 1, 1))
        d = rng.normal(size=(1, 1, 3))
        tri = rng.normal(size=(n, 3, 3)) * 0.01
        tri -= tri.mean(axis=1, keepdims=True)
        pos = (t * d + tri).reshape(-1, 3)
    elif kind == 5:
        pos = rng.uniform(-1, 1, (3 * n, 3))
        k = max(1, n // 3)
        pos[:3 * k] = np.tile(pos[:3], (k, 1))
        pos[-3:] = pos[-1]
    elif kind == 6:
        pos = rng.uniform(-1, 1, (3 * n, 3)) * np.exp(rng.uniform(-20, 20))
        pos += rng.uniform(-1, 1, 3) * 1e6 * (seed % 2)
    else:
        tri = rng.normal(size=(n, 3, 3))
        tri -= tri.mean(axis=1, keepdims=True)
        pos = tri.reshape(-1, 3) * rng.uniform(0.1, 2.0)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    if seed % 3 == 0 and n > 4:
        idx = rng.integers(0, max(3, n), (n, 3)).astype(np.uint32)
    mat = np.array([T.make_material(diffuse=(0.5, 0.5, 0.5))], dtype=T.MATERIAL_DTYPE)
    mesh = prt_amd.Mesh.from_arrays(idx, pos, np.zeros(n, dtype=np.uint32), mat.view(prt_amd.MATERIAL_DTYPE))
    mesh.calculate_bounds()
    scene = prt_amd.Scene()
    scene.add(mesh)  # Bvh::build of the host classes
    camera = prt_amd.Camera().create((0, 0, 5), (0, 0, -1), 16, 16)
    host = scene.arrays()["meshes"][0]
    ref, _, _ = T.ref_bvh(T.scene_desc_from_product(scene, camera, 1.0))
    r = ref[0]
    ok = len(r["nodes"]) == len(host["nodes"]) and all((r["nodes"][f] == host["nodes"][f]).all() for f in ("primOrSecondNodeIndex", "primCount", "splitAxis")) \
        and np.array_equal(r["nodes"]["lower"], host["nodes"]["lower"]) and np.array_equal(r["nodes"]["upper"], host["nodes"]["upper"]) and (r["remap"] == host["remap"]).all()
    if ok:
        leaf = host["nodes"]["primCount"] != 0xF
        ok = (r["nodes"]["triVectorIndex"][leaf] == host["nodes"]["triVectorIndex"][leaf]).all()
    if not ok:
        bad += 1
        print("seed", seed, "kind", kind, "n", n, "MISMATCH", len(r["nodes"]), len(host["nodes"]), flush=True)
    if seed % 25 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
