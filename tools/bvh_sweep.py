"""Sweep of the GPU BVH build against the host builder (tests/test_gpu_parity.py::test_gpu_bvh_build_matches_host_builder_on_soups over
many more shapes): random sizes 1..20000, uniform / clustered / gridded / planar / collinear-centroid / duplicated / huge-and-tiny
coordinate soups with shared vertices.  Node arrays, leaf order and primRemapping must be identical.  usage: bvh_sweep.py FIRST LAST"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import prt_amd
import test_gpu_parity as G
tr = prt_amd.PathTracer(device=0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    rng = np.random.default_rng(40000 + seed)
    n = int(np.exp(rng.uniform(0, np.log(20000))))
    kind = int(rng.integers(0, 8))
    if kind == 0:
        pos = rng.uniform(-1, 1, (3 * n, 3))
    elif kind == 1:  # clusters of very different density
        c = rng.uniform(-10, 10, (int(rng.integers(1, 6)), 3))
        pos = c[rng.integers(0, len(c), 3 * n)] + rng.normal(size=(3 * n, 3)) * np.exp(rng.uniform(-6, 0, (3 * n, 1)))
    elif kind == 2:  # a regular grid: many equal centroid coordinates, ties in every bucket
        g = int(np.ceil(n ** 0.5))
        q = np.stack(np.meshgrid(np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 2)[:n]
        base = np.concatenate([q, np.zeros((n, 1))], 1)
        pos = (base[:, None, :] + np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]])[None]).reshape(-1, 3)
    elif kind == 3:  # planar: one axis has zero extent
        pos = rng.uniform(-1, 1, (3 * n, 3))
        pos[:, int(rng.integers(0, 3))] = rng.uniform(-1, 1)
    elif kind == 4:  # all centroids on a line
        t = rng.uniform(-1, 1, (n, 1, 1))
        d = rng.normal(size=(1, 1, 3))
        tri = rng.normal(size=(n, 3, 3)) * 0.01
        tri -= tri.mean(axis=1, keepdims=True)
        pos = (t * d + tri).reshape(-1, 3)
    elif kind == 5:  # duplicates and degenerate triangles
        pos = rng.uniform(-1, 1, (3 * n, 3))
        k = max(1, n // 3)
        pos[:3 * k] = np.tile(pos[:3], (k, 1))
        pos[-3:] = pos[-1]
    elif kind == 6:  # huge and tiny coordinates
        pos = rng.uniform(-1, 1, (3 * n, 3)) * np.exp(rng.uniform(-20, 20))
        pos += rng.uniform(-1, 1, 3) * 1e6 * (seed % 2)
    else:  # identical centroids everywhere (the fallback to the middle)
        tri = rng.normal(size=(n, 3, 3))
        tri -= tri.mean(axis=1, keepdims=True)
        pos = tri.reshape(-1, 3) * rng.uniform(0.1, 2.0)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    if seed % 3 == 0 and n > 4:  # shared vertices
        idx = rng.integers(0, max(3, n), (n, 3)).astype(np.uint32)
    try:
        nodes, remap, ms = tr.build_bvh(idx, pos)
        ref_nodes, ref_remap = G._host_bvh(idx, pos)
        G._assert_same_tree(nodes, remap, ref_nodes, ref_remap, f"seed {seed}")
    except AssertionError as e:
        bad += 1
        print("seed", seed, "kind", kind, "n", n, "MISMATCH", str(e)[:200], flush=True)
    if seed % 25 == 0:
        print("seed", seed, "done", flush=True)
print("sweep finished,", bad, "failures", flush=True)
sys.exit(1 if bad else 0)
