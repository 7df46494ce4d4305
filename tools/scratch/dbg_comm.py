import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import prt_amd
L = prt_amd.lib()
t = prt_amd.PathTracer(device=0, max_depth=4)
uid = prt_amd.comm_unique_id()
print("uid", uid[:16].hex(), flush=True)
buf = C.create_string_buffer(bytes(uid), 128)
rc = L.prt_hip_comm_init(t._ctx, buf, 0, 1)
print("init rc", rc, L.prt_hip_last_error(), flush=True)
n = C.c_uint64()
rc = L.prt_hip_gather_rccl(t._ctx, None, 0, None)
print("gather rc", rc, L.prt_hip_last_error(), flush=True)
