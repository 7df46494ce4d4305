"""Child process of test_rccl_gather_two_and_three_ranks_through_the_standin (GPU box only).

Runs prt_hip_gather_rccl with MORE THAN ONE rank on a one-GPU box: PRT_RCCL_LIB points the product at tests/fake_rccl.cpp, whose
ranks are the threads started here (one context per rank, all on device 0).  Every case requires the root's image to equal the
one-GPU render bit for bit.  A process of its own because the product binds its RCCL library once per process and the suite's
other tests bind the real one.  Prints "OK <cases>" on success."""
import ctypes as C
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prt_amd  # noqa: E402


def download(t, W, H):
    out = np.zeros((H, W, 3), dtype=np.float32)
    t._chk(t._L.prt_hip_download(t._ctx, out.ctypes.data_as(C.c_void_p), 0, 0, W - 1, H - 1), "download")
    return out


def in_threads(fns):
    """Run the callables concurrently (ctypes releases the GIL inside the library); re-raise the first exception."""
    errs = [None] * len(fns)

    def wrap(i):
        try:
            fns[i]()
        except BaseException as e:  # noqa: BLE001
            errs[i] = e

    th = [threading.Thread(target=wrap, args=(i,)) for i in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        assert not t.is_alive(), "a rank hangs in the gather"
    for e in errs:
        if e is not None:
            raise e


def main():
    fake = C.CDLL(os.environ["PRT_RCCL_LIB"])  # the same loaded object the product binds
    hip = C.CDLL("libamdhip64.so")
    fake.fake_rccl_counts.argtypes = [C.POINTER(C.c_uint64)] * 3
    cases = 0
    # (width, height, ranks, root, tile): ragged right/bottom tiles, a root that is not rank 0, more ranks than tiles, another tile size
    for W, H, n, root, tile in ((200, 120, 2, 0, 16), (333, 211, 3, 1, 16), (16, 16, 3, 0, 16), (97, 65, 3, 2, 8), (130, 70, 2, 1, 32)):
        scene, camera, exposure = prt_amd.setup_cornell_box(W, H)
        one = prt_amd.PathTracer(device=0, max_depth=4, seed=777)
        one.upload_scene(scene)
        one.set_camera(camera)
        whole = one.render(8, tile=tile)
        one.close()
        ranks = [prt_amd.PathTracer(device=0, max_depth=4, seed=777) for _ in range(n)]
        try:
            uid = prt_amd.comm_unique_id()
            for i, t in enumerate(ranks):
                t.upload_scene(scene)
                t.set_camera(camera)
                t.comm_init(uid, i, n)
            for rep in range(2):  # twice: the staging buffers of the first gather are reused by the second
                fbp = ranks[root]._L.prt_hip_framebuffer(ranks[root]._ctx)  # device pointer, or None before the first render
                if fbp:  # forget the first gather's image
                    assert hip.hipMemset(C.c_void_p(fbp), 0, C.c_size_t(W * H * 12)) == 0 and hip.hipDeviceSynchronize() == 0
                for i, t in enumerate(ranks):
                    t.render_async(0, 0, W - 1, H - 1, 8, rank=i, nranks=n, tile=tile)
                own = download(ranks[root], W, H)
                mask = prt_amd.owned_pixel_mask(W, H, root, n, tile)
                assert own[mask].tobytes() == whole[mask].tobytes()
                if n > 1 and (~mask).any():
                    assert own.tobytes() != whole.tobytes(), "the root must not hold the other ranks' pixels before the gather"
                s0, r0, b0 = C.c_uint64(), C.c_uint64(), C.c_uint64()
                fake.fake_rccl_counts(s0, r0, b0)
                in_threads([lambda t=t: t.gather_rccl(root=root) for t in ranks])
                got = download(ranks[root], W, H)
                assert got.tobytes() == whole.tobytes(), f"{W}x{H}, {n} ranks, root {root}, tile {tile}: gathered image differs from the one-GPU image"
                s1, r1, b1 = C.c_uint64(), C.c_uint64(), C.c_uint64()
                fake.fake_rccl_counts(s1, r1, b1)
                payload = sum(t.gather_payload_bytes() for i, t in enumerate(ranks) if i != root)
                assert b1.value - b0.value == payload, (b1.value - b0.value, payload)  # exactly the tiles the peers own, nothing else
                senders = sum(1 for i, t in enumerate(ranks) if i != root and t.gather_payload_bytes())
                assert s1.value - s0.value == senders and r1.value - r0.value == senders
            cases += 1
        finally:
            for t in ranks:
                t.close()

    # ---- a failing ncclSend / ncclRecv inside the group: the error comes back AND the group is closed again
    W, H = 96, 64
    scene, camera, exposure = prt_amd.setup_cornell_box(W, H)
    ranks = [prt_amd.PathTracer(device=0, max_depth=4, seed=777) for _ in range(2)]
    try:
        uid = prt_amd.comm_unique_id()
        for i, t in enumerate(ranks):
            t.upload_scene(scene)
            t.set_camera(camera)
            t.comm_init(uid, i, 2)
            t.render_async(0, 0, W - 1, H - 1, 8, rank=i, nranks=2)
        for who, what in ((1, 1), (0, 2)):  # the sender's ncclSend, the root's ncclRecv (this thread plays that rank alone)
            fake.fake_rccl_fail_next(what)
            try:
                ranks[who].gather_rccl(root=0)
                raise AssertionError("the injected RCCL error was swallowed")
            except prt_amd.PrtError as e:
                assert "internal error" in str(e), str(e)
            assert fake.fake_rccl_open_groups() == 0, "prt_hip_gather_rccl returned with the RCCL group still open"
            # RCCL leaves a communicator in an error state after a failed operation: the context refuses it from now on ...
            try:
                ranks[who].gather_rccl(root=0)
                raise AssertionError("a gather on a broken communicator was let through")
            except prt_amd.PrtError as e:
                assert "unusable after a failed" in str(e), str(e)
        # ... until every rank replaces it (the broken ones end by ncclCommAbort), and the new one works
        aborted0 = fake.fake_rccl_aborted_comms()
        uid = prt_amd.comm_unique_id()
        for i, t in enumerate(ranks):
            t.comm_init(uid, i, 2)
        assert fake.fake_rccl_aborted_comms() - aborted0 == 2
        in_threads([lambda t=t: t.gather_rccl(root=0) for t in ranks])
        one = prt_amd.PathTracer(device=0, max_depth=4, seed=777)
        one.upload_scene(scene)
        one.set_camera(camera)
        assert download(ranks[0], W, H).tobytes() == one.render(8).tobytes()
        one.close()
        cases += 1
    finally:
        for t in ranks:
            t.close()
    print("OK", cases)


if __name__ == "__main__":
    main()
