"""Frame-to-frame variation of the C3 kernel time (clock behaviour, stream choice).  Diagnostic only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    fb = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
    ts = torch.cuda.Stream()
else:
    torch = None
import prt_amd
W, H, spp, depth = 1920, 1080, 64, 8
scene, camera, exposure = prt_amd.setup_atrium_standin(W, H, tris=262000, seed=1)
tr = prt_amd.PathTracer(device=0, max_depth=depth, seed=12345)
tr.upload_scene(scene); tr.set_camera(camera)
for i in range(3):
    tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure)
    st = tr.stats()
    print(f"ctx stream frame {i}: {st['kernelMs']:.1f} ms", flush=True)
if torch:
    for i in range(3):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, d_rgb=fb.data_ptr(), stream=ts.cuda_stream)
        torch.cuda.synchronize()
        st = tr.stats()
        print(f"torch stream frame {i}: {st['kernelMs']:.1f} ms", flush=True)
    for i in range(3):
        tr.render_async(0, 0, W - 1, H - 1, spp, exposure=exposure, d_rgb=fb.data_ptr())
        st = tr.stats()
        print(f"ctx stream, torch fb, frame {i}: {st['kernelMs']:.1f} ms", flush=True)
os.system("cat /proc/%d/maps | grep -o '/[^ ]*libamdhip64[^ ]*' | sort -u" % os.getpid())
