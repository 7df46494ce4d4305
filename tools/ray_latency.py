"""Latency of one long scatter ray traced alone (64 copies in one wave): microseconds per traversal step.  Diagnostic."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, prt_amd, prt_testlib as T
W, H = 1920, 1080
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
desc = T.scene_desc_from_product(scene, camera, exposure)
osc = T.OracleScene(desc)
L = T.oracle()
rng = np.random.default_rng(1)
n = 6000
cam = np.array(camera.pos_arg, dtype=np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True); d[:, 2] = -np.abs(d[:, 2])
h = T.OrcHit()
pts = []
for r in range(n):
    L.orc_intersect_single(osc.scene, T.fp(cam), T.fp(d[r]), 1e5, C.byref(h), None)
    if h.t > 0: pts.append(cam + np.float32(h.t * 0.999) * d[r])
pts = np.array(pts, dtype=np.float32)
d2 = rng.normal(size=(len(pts), 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
steps = np.zeros(len(pts), dtype=np.int64); taps = np.zeros(len(pts), dtype=np.int64)
for r in range(len(pts)):
    st = T.OrcStats()
    L.orc_intersect_single(osc.scene, T.fp(pts[r]), T.fp(d2[r]), 1e5, C.byref(h), C.byref(st))
    steps[r] = st.nBox // 2 + (st.nTri + 1) // 2; taps[r] = st.nTap
tr = prt_amd.PathTracer(device=0, max_depth=8, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
order = np.argsort(steps)
for which in (order[-1], order[-2], order[len(order) // 2], order[len(order) // 10]):
    o = np.repeat(pts[which][None], 64, 0); dd = np.repeat(d2[which][None], 64, 0)
    best = 1e9
    base = 1e9
    for i in range(5):
        t0 = time.perf_counter(); tr.trace_rays(0, o, dd, 1e5); best = min(best, time.perf_counter() - t0)
        # a ray that misses everything: the call's fixed cost
        t0 = time.perf_counter(); tr.trace_rays(0, o + 1e6, dd, 1e5); base = min(base, time.perf_counter() - t0)
    print(f"ray with {steps[which]} steps ({taps[which]} alpha taps): call {best*1e6:.0f} us, empty call {base*1e6:.0f} us -> {(best-base)*1e6/max(1,steps[which]):.2f} us/step", flush=True)
