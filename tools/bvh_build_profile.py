"""Builds the BVH of the C4-class stand-in (2.5 M triangles) on the GPU three times, for a rocprofv3 --kernel-trace --stats run
(per-kernel times of prt_bvh_build.hip).  Diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prt_amd
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 2500000
scene, camera, exposure = prt_amd.setup_atrium_standin(64, 64, tris=tris, seed=1)  # (the scene's own BVH comes from the host builder)
m = scene.arrays()["meshes"][0]
idx, pos = m["indices"], m["positions"]
tr = prt_amd.PathTracer(device=0)
for i in range(3):
    nodes, remap, ms = tr.build_bvh(idx, pos)
    print(f"{len(idx)} triangles -> {len(nodes)} nodes in {ms:.2f} ms on the device", flush=True)
tr.close()
