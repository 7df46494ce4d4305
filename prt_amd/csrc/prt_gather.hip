// prt_gather.hip -- the one exchange of the multi-GPU path: the image gather.
//
// Tiles are dealt round-robin (tile id % nranks, prt_render_params.rank/nranks), so a rank's pixels are scattered over the
// frame.  Each rank PACKS the tiles it owns into a tile-major buffer (tile q of rank r = tile id r + q*nranks; T*T pixels
// per tile, rows of the tile contiguous, pixels outside the image zero), the packed tiles travel, and the root UNPACKS
// them into its frame.  Payload per rank = (tiles owned) * T*T*12 bytes = 1/nranks of the image (3.1 MB at 1080p with 8
// ranks) instead of a full-frame reduction.
//   * prt_hip_gather_rccl  one process per GPU: grouped ncclSend / ncclRecv to the root, so every xGMI link into the
//                          root carries exactly one peer's tiles (no ring, no reduction).  RCCL is loaded with dlopen on
//                          first use -- a single-GPU host never needs librccl.
//   * prt_hip_gather       one process driving several contexts: the same pack and unpack kernels, device-to-device copies
//                          in between.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "prt_internal.h"

namespace {

// ---- tile-major packing.  One thread per pixel of a packed tile.
struct PackArgs {
    float* frame;      // W*H*3
    float* packed;     // tiles * T*T*3
    uint32_t W, H, T, tilesX;
    uint32_t rank, nranks; // tiles rank, rank + nranks, ... of the image's tile grid
    uint32_t tiles;        // tiles this rank owns
};

__global__ void pack_tiles_kernel(PackArgs A)
{
    const uint32_t T2 = A.T * A.T;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)A.tiles * T2) return;
    const uint32_t q = (uint32_t)(i / T2), p = (uint32_t)(i % T2);
    const uint32_t tile = A.rank + q * A.nranks;
    const uint32_t x = (tile % A.tilesX) * A.T + p % A.T, y = (tile / A.tilesX) * A.T + p / A.T;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (x < A.W && y < A.H) {
        const float* px = A.frame + ((size_t)x + (size_t)y * A.W) * 3;
        r = px[0]; g = px[1]; b = px[2];
    }
    float* o = A.packed + i * 3;
    o[0] = r; o[1] = g; o[2] = b;
}

__global__ void unpack_tiles_kernel(PackArgs A)
{
    const uint32_t T2 = A.T * A.T;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)A.tiles * T2) return;
    const uint32_t q = (uint32_t)(i / T2), p = (uint32_t)(i % T2);
    const uint32_t tile = A.rank + q * A.nranks;
    const uint32_t x = (tile % A.tilesX) * A.T + p % A.T, y = (tile / A.tilesX) * A.T + p / A.T;
    if (x >= A.W || y >= A.H) return;
    const float* o = A.packed + i * 3;
    float* px = A.frame + ((size_t)x + (size_t)y * A.W) * 3;
    px[0] = o[0]; px[1] = o[1]; px[2] = o[2];
}

uint32_t tiles_owned(uint32_t totalTiles, uint32_t rank, uint32_t nranks)
{
    return rank < totalTiles ? (totalTiles - rank + nranks - 1) / nranks : 0;
}

int ensure_floats(float** buf, size_t* have, size_t need)
{
    if (need <= *have) return PRT_HIP_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc((void**)buf, std::max<size_t>(need, 64) * sizeof(float)));
    *have = need;
    return PRT_HIP_OK;
}

void launch_pack(bool unpack, float* frame, float* packed, uint32_t W, uint32_t H, uint32_t T, uint32_t rank, uint32_t nranks, uint32_t tiles,
                 hipStream_t s)
{
    if (tiles == 0) return;
    PackArgs A{frame, packed, W, H, T, (W + T - 1) / T, rank, nranks, tiles};
    const uint64_t threads = (uint64_t)tiles * T * T;
    const dim3 grid((uint32_t)((threads + 255) / 256)), block(256);
    if (unpack) hipLaunchKernelGGL(unpack_tiles_kernel, grid, block, 0, s, A);
    else hipLaunchKernelGGL(pack_tiles_kernel, grid, block, 0, s, A);
}

// ---- RCCL through dlopen: the types are spelled out so that the product does not need rccl.h to build
typedef struct { char internal[128]; } rccl_unique_id; // ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40-43)
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(void**, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr; // optional: ends a communicator whose operations may never complete
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
const int kNcclFloat = 7; // ncclFloat32 (rccl.h ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7)

Rccl& rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // PRT_RCCL_LIB names the library to bind instead (read once, never written): a host with its own RCCL build, or the
        // test suite's stand-in whose "ranks" are threads of one process (tests/fake_rccl.cpp) -- the only way the > 1-rank
        // branch below can run on a one-GPU box, where RCCL itself refuses two ranks on one device.
        const char* override = getenv("PRT_RCCL_LIB");
        if (override && *override) {
            R.handle = dlopen(override, RTLD_NOW | RTLD_LOCAL);
            if (!R.handle) {
                const char* e = dlerror(); // (a second call returns NULL: the first one clears the message)
                R.why = std::string("PRT_RCCL_LIB=") + override + ": " + (e ? e : "?");
                return;
            }
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (R.handle) break;
            R.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!R.handle) {
            const char* e = dlerror();
            R.why = std::string("librccl not found: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(R.handle, n);
            if (!p && R.why.empty()) R.why = std::string("librccl lacks ") + n;
            return p;
        };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.CommAbort = (decltype(R.CommAbort))dlsym(R.handle, "ncclCommAbort");
        R.CommCount = (decltype(R.CommCount))sym("ncclCommCount");
        R.CommUserRank = (decltype(R.CommUserRank))sym("ncclCommUserRank");
        R.Send = (decltype(R.Send))sym("ncclSend");
        R.Recv = (decltype(R.Recv))sym("ncclRecv");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    });
    return R;
}

#define RCCL_TRY(R, call)                                                                                     \
    do {                                                                                                      \
        int r_ = (call);                                                                                      \
        if (r_ != 0) return prt_fail(PRT_HIP_ECOMM, std::string(#call) + ": " + (R).GetErrorString(r_));      \
    } while (0)

// Inside ncclGroupStart .. ncclGroupEnd: a failing call must not leave the group open (every later RCCL call of this thread
// would be queued into it and never run) -- close it, then report the FIRST error.  Closing the group also LAUNCHES whatever was
// queued before the failure (a root that queued k of n receives now waits for those k peers), and RCCL leaves a communicator
// in an error state after a failed operation: the context marks it broken, every later gather on it is refused, and
// prt_hip_comm_init / _adopt / _destroy end it with ncclCommAbort (which does not wait for operations that cannot complete).
#define RCCL_TRY_IN_GROUP(R, c, call)                                                                         \
    do {                                                                                                      \
        int r_ = (call);                                                                                      \
        if (r_ != 0) {                                                                                        \
            (void)(R).GroupEnd();                                                                             \
            (c)->commBroken = true;                                                                           \
            return prt_fail(PRT_HIP_ECOMM, std::string(#call) + ": " + (R).GetErrorString(r_));               \
        }                                                                                                     \
    } while (0)

// Ends the context's communicator if the context owns it: a broken one by ncclCommAbort where the library has it.
static void drop_comm(Rccl& R, prt_hip_ctx* c)
{
    if (c->comm && c->commOwned) {
        if (c->commBroken && R.CommAbort) (void)R.CommAbort(c->comm);
        else if (R.CommDestroy) (void)R.CommDestroy(c->comm);
    }
    c->comm = nullptr;
    c->commOwned = false;
    c->commBroken = false;
}

int need_rccl(Rccl** out)
{
    Rccl& R = rccl();
    if (!R.why.empty()) return prt_fail(PRT_HIP_ECOMM, R.why);
    *out = &R;
    return PRT_HIP_OK;
}

} // namespace

void prt_gather_release(prt_hip_ctx* c)
{
    if (c->packBuf) (void)hipFree(c->packBuf);
    if (c->stageBuf) (void)hipFree(c->stageBuf);
    c->packBuf = c->stageBuf = nullptr;
    c->packFloats = c->stageFloats = 0;
    if (c->comm) drop_comm(rccl(), c);
}

extern "C" {

int prt_hip_comm_unique_id(void* id)
{
    if (!id) return prt_fail(PRT_HIP_EINVAL, "id is NULL");
    Rccl* R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    rccl_unique_id u;
    RCCL_TRY(*R, R->GetUniqueId(&u));
    static_assert(sizeof(u) == PRT_HIP_COMM_ID_BYTES, "unique id size");
    memcpy(id, &u, sizeof(u));
    return PRT_HIP_OK;
}

int prt_hip_comm_init(prt_hip_ctx* c, const void* id, int rank, int nranks)
{
    if (!c || !id) return prt_fail(PRT_HIP_EINVAL, "NULL argument");
    if (nranks <= 0 || rank < 0 || rank >= nranks) return prt_fail(PRT_HIP_EINVAL, "rank outside 0..nranks-1");
    Rccl* R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    drop_comm(*R, c);
    rccl_unique_id u;
    memcpy(&u, id, sizeof(u));
    void* comm = nullptr;
    RCCL_TRY(*R, R->CommInitRank(&comm, nranks, u, rank));
    if (!comm) return prt_fail(PRT_HIP_ECOMM, "ncclCommInitRank succeeded but returned no communicator");
    c->comm = comm;
    c->commOwned = true;
    c->commRank = rank;
    c->commSize = nranks;
    return PRT_HIP_OK;
}

int prt_hip_comm_adopt(prt_hip_ctx* c, void* ncclComm)
{
    if (!c || !ncclComm) return prt_fail(PRT_HIP_EINVAL, "NULL argument");
    Rccl* R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    int n = 0, r = 0;
    RCCL_TRY(*R, R->CommCount(ncclComm, &n));
    RCCL_TRY(*R, R->CommUserRank(ncclComm, &r));
    drop_comm(*R, c);
    c->comm = ncclComm;
    c->commOwned = false;
    c->commRank = r;
    c->commSize = n;
    return PRT_HIP_OK;
}

int prt_hip_comm_destroy(prt_hip_ctx* c)
{
    if (!c) return prt_fail(PRT_HIP_EINVAL, "ctx is NULL");
    if (c->comm && c->commOwned) {
        Rccl* R;
        int rc = need_rccl(&R);
        if (rc) return rc;
        HIP_TRY(hipSetDevice(c->device));
        if (c->commBroken) {
            drop_comm(*R, c); // (no stream synchronisation first: an operation of a broken communicator may never complete)
        } else {
            HIP_TRY(hipStreamSynchronize(c->stream));
            RCCL_TRY(*R, R->CommDestroy(c->comm));
        }
    }
    c->comm = nullptr;
    c->commOwned = false;
    c->commBroken = false;
    c->commSize = 0;
    return PRT_HIP_OK;
}

int prt_hip_gather_rccl(prt_hip_ctx* c, float* d_rgb, int root, void* stream)
{
    if (!c) return prt_fail(PRT_HIP_EINVAL, "ctx is NULL");
    if (!c->comm) {
        char where[64];
        snprintf(where, sizeof(where), " (ctx %p, size %d)", (void*)c, c->commSize);
        return prt_fail(PRT_HIP_ESTATE, std::string("no communicator: call prt_hip_comm_init or prt_hip_comm_adopt first") + where);
    }
    if (c->commBroken)
        return prt_fail(PRT_HIP_ECOMM, "the communicator is unusable after a failed ncclSend / ncclRecv: call prt_hip_comm_init (or prt_hip_comm_adopt) again on every rank");
    if (root < 0 || root >= c->commSize) return prt_fail(PRT_HIP_EINVAL, "root outside the communicator");
    if (!c->haveCamera || c->lastTile == 0) return prt_fail(PRT_HIP_ESTATE, "nothing rendered yet");
    if (c->lastNranks != (uint32_t)c->commSize || c->lastRank != (uint32_t)c->commRank)
        return prt_fail(PRT_HIP_ESTATE, "the last render's rank/nranks must be the communicator's");
    Rccl* R;
    int rc = need_rccl(&R);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!d_rgb) d_rgb = c->fb;
    if (!d_rgb) return prt_fail(PRT_HIP_ESTATE, "no framebuffer");
    const uint32_t W = c->cam.width, H = c->cam.height, T = c->lastTile;
    const uint32_t totalTiles = ((W + T - 1) / T) * ((H + T - 1) / T);
    const uint32_t n = (uint32_t)c->commSize, me = (uint32_t)c->commRank;
    const size_t tileFloats = (size_t)T * T * 3;
    hipStream_t s = c->stream;
    hipStream_t caller = (stream && (hipStream_t)stream != c->stream) ? (hipStream_t)stream : nullptr;
    if (caller) {
        HIP_TRY(hipEventRecord(c->evIn, caller));
        HIP_TRY(hipStreamWaitEvent(s, c->evIn, 0));
    }
    if (n > 1) {
        if (me != (uint32_t)root) {
            const uint32_t mine = tiles_owned(totalTiles, me, n);
            if ((rc = ensure_floats(&c->packBuf, &c->packFloats, mine * tileFloats))) return rc;
            launch_pack(false, d_rgb, c->packBuf, W, H, T, me, n, mine, s);
            RCCL_TRY(*R, R->GroupStart());
            if (mine) RCCL_TRY_IN_GROUP(*R, c, R->Send(c->packBuf, mine * tileFloats, kNcclFloat, root, c->comm, s));
            RCCL_TRY(*R, R->GroupEnd());
        } else {
            // staged tiles of rank r start at off[r]; the root's own tiles are already where they belong
            std::vector<size_t> off(n + 1, 0);
            for (uint32_t r = 0; r < n; r++) off[r + 1] = off[r] + (r == me ? 0 : (size_t)tiles_owned(totalTiles, r, n) * tileFloats);
            if ((rc = ensure_floats(&c->stageBuf, &c->stageFloats, off[n]))) return rc;
            RCCL_TRY(*R, R->GroupStart());
            for (uint32_t r = 0; r < n; r++) {
                const size_t cnt = off[r + 1] - off[r];
                if (r != me && cnt) RCCL_TRY_IN_GROUP(*R, c, R->Recv(c->stageBuf + off[r], cnt, kNcclFloat, (int)r, c->comm, s));
            }
            RCCL_TRY(*R, R->GroupEnd());
            for (uint32_t r = 0; r < n; r++)
                if (r != me) launch_pack(true, d_rgb, c->stageBuf + off[r], W, H, T, r, n, tiles_owned(totalTiles, r, n), s);
        }
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return prt_fail(PRT_HIP_ELAUNCH, std::string("gather kernels: ") + hipGetErrorString(le));
    }
    if (caller) {
        HIP_TRY(hipEventRecord(c->evOut, s));
        HIP_TRY(hipStreamWaitEvent(caller, c->evOut, 0));
    }
    return PRT_HIP_OK;
}

int prt_hip_gather_payload_bytes(prt_hip_ctx* c, uint64_t* bytes)
{
    if (!c || !bytes) return prt_fail(PRT_HIP_EINVAL, "NULL argument");
    if (!c->haveCamera || c->lastTile == 0 || c->lastNranks == 0) return prt_fail(PRT_HIP_ESTATE, "nothing rendered yet");
    const uint32_t W = c->cam.width, H = c->cam.height, T = c->lastTile;
    const uint32_t totalTiles = ((W + T - 1) / T) * ((H + T - 1) / T);
    *bytes = (uint64_t)tiles_owned(totalTiles, c->lastRank, c->lastNranks) * T * T * 3 * sizeof(float);
    return PRT_HIP_OK;
}

int prt_hip_gather(prt_hip_ctx* const* ctxs, int n, float* rgb_host, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    if (!ctxs || n <= 0 || !rgb_host) return prt_fail(PRT_HIP_EINVAL, "bad argument");
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !ctxs[i]->fb) return prt_fail(PRT_HIP_ESTATE, "a context has nothing rendered into its framebuffer");
        if (ctxs[i]->lastNranks != (uint32_t)n || ctxs[i]->lastRank != (uint32_t)i)
            return prt_fail(PRT_HIP_ESTATE, "context i must have rendered with rank = i, nranks = n");
        if (ctxs[i]->cam.width != ctxs[0]->cam.width || ctxs[i]->cam.height != ctxs[0]->cam.height || ctxs[i]->lastTile != ctxs[0]->lastTile)
            return prt_fail(PRT_HIP_ESTATE, "contexts disagree on image size or tile size");
    }
    prt_hip_ctx* root = ctxs[0];
    const uint32_t W = root->cam.width, H = root->cam.height, T = root->lastTile;
    if (x1 < x0 || y1 < y0 || x1 >= W || y1 >= H) return prt_fail(PRT_HIP_EINVAL, "pixel rectangle outside the image");
    const uint32_t totalTiles = ((W + T - 1) / T) * ((H + T - 1) / T);
    const size_t tileFloats = (size_t)T * T * 3;
    int rc;
    // every other context packs its tiles; the packed tiles are copied device to device and unpacked into context 0's frame
    for (int i = 1; i < n; i++) {
        prt_hip_ctx* c = ctxs[i];
        const uint32_t mine = tiles_owned(totalTiles, (uint32_t)i, (uint32_t)n);
        if (mine == 0) continue;
        HIP_TRY(hipSetDevice(c->device));
        if ((rc = ensure_floats(&c->packBuf, &c->packFloats, mine * tileFloats))) return rc;
        launch_pack(false, c->fb, c->packBuf, W, H, T, (uint32_t)i, (uint32_t)n, mine, c->stream);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipSetDevice(root->device));
        if ((rc = ensure_floats(&root->stageBuf, &root->stageFloats, mine * tileFloats))) return rc;
        HIP_TRY(hipStreamSynchronize(root->stream)); // the previous unpack has consumed the staging buffer
        if (c->device == root->device) HIP_TRY(hipMemcpy(root->stageBuf, c->packBuf, mine * tileFloats * sizeof(float), hipMemcpyDeviceToDevice));
        else HIP_TRY(hipMemcpyPeer(root->stageBuf, root->device, c->packBuf, c->device, mine * tileFloats * sizeof(float)));
        launch_pack(true, root->fb, root->stageBuf, W, H, T, (uint32_t)i, (uint32_t)n, mine, root->stream);
    }
    HIP_TRY(hipSetDevice(root->device));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return prt_fail(PRT_HIP_ELAUNCH, std::string("gather kernels: ") + hipGetErrorString(le));
    HIP_TRY(hipStreamSynchronize(root->stream));
    for (int i = 0; i < n; i++) { // a launch that was cut short (watchdog, stack overflow) must not pass for an image
        HIP_TRY(hipSetDevice(ctxs[i]->device));
        if (i) HIP_TRY(hipStreamSynchronize(ctxs[i]->stream));
        if ((rc = prt_sticky_error(ctxs[i], false))) return rc;
    }
    HIP_TRY(hipSetDevice(root->device));
    const size_t rowBytes = (size_t)(x1 - x0 + 1) * 3 * sizeof(float), off = ((size_t)y0 * W + x0) * 3;
    HIP_TRY(hipMemcpy2D(rgb_host + off, (size_t)W * 3 * sizeof(float), root->fb + off, (size_t)W * 3 * sizeof(float), rowBytes, y1 - y0 + 1,
                        hipMemcpyDeviceToHost));
    return PRT_HIP_OK;
}

} // extern "C"
