import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import prt_amd, prt_testlib as T
tr = prt_amd.PathTracer()
scene, cam, exp = prt_amd.setup_atrium_standin(96, 54, tris=20000, alpha=False, bump=False, light=True)
tr.upload_scene(scene); tr.set_camera(cam)
desc = T.scene_desc_from_product(scene, cam, exp)
s = T.OracleScene(desc)
a = desc.product_arrays["meshes"][0]
nodes = a["nodes"]
print("nodes", len(nodes), "prims", len(a["indices"]))
rng = np.random.default_rng(3)
n = 8192
org = rng.uniform([-17, 0.2, -6.5], [17, 11, 6.5], (n, 3)).astype(np.float32)
org[: n // 2] = (-15.0, 4.0, 0.5)
d = rng.normal(size=(n, 3)); d[: n // 2, 0] = np.abs(d[: n // 2, 0]) * 3
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
far = float(np.float32(2.0) * np.float32(s.radius()))
o1, c1 = s.intersect_single(org, d, far)
o8, c8 = s.intersect_packet(org, d, far)
for mode, ref in ((0, o1), (1, o8)):
    got = tr.trace_rays(mode, org, d, far)
    bad = np.nonzero(got["t"].view(np.uint32) != ref["t"].view(np.uint32))[0]
    print("mode", mode, "mismatch t:", len(bad))
    for i in bad[:6]:
        print("  ray", i, org[i], d[i], "gpu", got[i], "oracle", ref[i])
g2 = tr.trace_rays(2, org, d, far)["t"]; g3 = tr.trace_rays(3, org, d, far)["t"]
print("occl single mismatches", int((g2 != c1).sum()), "packet", int((g3 != c8).sum()))
# tree depth / max stack
depth = 0; st=[(0,1)]
while st:
    i, dd = st.pop(); depth = max(depth, dd)
    if nodes["primCount"][i] == 0xF: st.append((i+1, dd+1)); st.append((int(nodes["primOrSecondNodeIndex"][i]), dd+1))
print("tree depth", depth)
