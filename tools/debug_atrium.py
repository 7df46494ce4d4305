import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import prt_amd, prt_testlib as T
tr = prt_amd.PathTracer()
for alpha in (False, True):
    for bump in (False, True):
        for light in (False, True):
            for depth in (1, 8):
                scene, cam, exp = prt_amd.setup_atrium_standin(96, 54, tris=20000, alpha=alpha, bump=bump, light=light)
                tr.upload_scene(scene); tr.set_camera(cam)
                rgb = tr.render(8, max_depth=depth, count_traffic=True)
                st = tr.stats()
                ref, ost = T.OracleScene(T.scene_desc_from_product(scene, cam, exp)).render(8, max_depth=depth)
                nd = int((rgb.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
                md = float(np.nanmax(np.abs(rgb - ref)))
                keys = ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap")
                print(f"alpha={alpha} bump={bump} light={light} depth={depth}: differing px {nd}/{96*54} maxabs {md:.3g} " +
                      " ".join(f"{k}:{st[k]-ost[k]:+d}" for k in keys), flush=True)
