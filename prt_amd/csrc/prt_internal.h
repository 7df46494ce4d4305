// prt_internal.h -- what the translation units of libprt_hip.so share: the context behind the C-ABI handle, the error
// helpers and the build constants of the wavefront pipeline.  Not installed; include/prt_hip.h is the interface.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/prt_hip.h"
#include "prt_device.h"

#define PRT_WORK_WORDS 512 // control words of a launch (frame kernel: prt_frame.h PRT_CTRL_CURSORS; claim cursor of the G-buffer kernel)
#define PRT_STICKY_WORDS 160 // behind the control words and NEVER cleared by a render: [0] error flags of any earlier launch (1 watchdog, 2 stack
                             // overflow), [2] + [8..135] the first watchdog reports; read and cleared by prt_hip_get_stats, read by download / gather
#define PRT_STAT_SHARDS 64 // copies of the statistics counters, summed on read-back
#ifdef PRT_PROFILE
#define PRT_STAT_STRIDE 128 // the profile build appends its lane-utilisation counters (words 32..127)
#else
#define PRT_STAT_STRIDE 32 // 64-bit words per copy (256 B apart)
#endif
#define PRT_TIMING_RING 32

// thread-local message behind prt_hip_last_error(); returns `code`
int prt_fail(int code, const std::string& msg);
const std::string& prt_last_error_string();

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return prt_fail(e_ == hipErrorOutOfMemory ? PRT_HIP_ENOMEM : PRT_HIP_ENODEVICE,             \
                            std::string(#expr) + ": " + hipGetErrorString(e_));                         \
    } while (0)

struct prt_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // Timing: a fixed ring of event pairs, one pair per render launch.  When the ring is full the oldest launches are folded
    // into accMs (they have long finished), so a caller that renders in a loop without reading the stats holds no more than
    // PRT_TIMING_RING pairs.
    hipEvent_t evT0[PRT_TIMING_RING] = {}, evT1[PRT_TIMING_RING] = {};
    uint32_t ringUsed = 0;
    double accMs = 0.0, lastMs = 0.0;
    uint64_t accLaunches = 0;
    int computeUnits = 0;
    std::string name;
    // scene
    bool haveScene = false, haveCamera = false;
    DevScene sc{};
    DevCamera cam{};
    std::vector<void*> sceneAllocs;
    // render resources
    float* fb = nullptr;
    size_t fbPixels = 0;
    uint32_t* work = nullptr; // Q_COUNT queue counters
    unsigned long long* counters = nullptr;
    hipEvent_t evIn = nullptr, evOut = nullptr; // order the pipeline against a caller's stream
    void* wfBuffer = nullptr; // the frame kernel's pool state and ray queues
    size_t wfBytes = 0;
    uint32_t* spill = nullptr;
    uint32_t spillThreads = 0;
    uint32_t lastRank = 0, lastNranks = 0, lastTile = 0; // of the last render (prt_hip_gather*)
    float* lastTarget = nullptr;                         // framebuffer of the last render (device)
    // image gather (prt_gather.hip): tile-major staging buffers and the RCCL communicator
    void* comm = nullptr;       // ncclComm_t
    bool commOwned = false;     // created by prt_hip_comm_init (destroyed with the context) or adopted from the host
    bool commBroken = false;    // a send / receive failed: the communicator is refused until it is replaced (prt_gather.hip)
    int commRank = 0, commSize = 0;
    float* packBuf = nullptr;   // this rank's tiles, tile-major
    size_t packFloats = 0;
    float* stageBuf = nullptr;  // root: the other ranks' tiles as received
    size_t stageFloats = 0;
    int frameBlocksPerCU = 0;  // resident blocks per CU of the frame kernel
    bool frameLaunched = false;
    void* frameArgs = nullptr;  // device copies of the frame kernel's argument block (ring)
    uint32_t frameArgSlot = 0;
    bool timed = false;
};


// prt_gather.hip: frees the staging buffers and an owned communicator
void prt_gather_release(prt_hip_ctx* c);
// prt_kernels.hip: 0, or the error code of a launch since the last prt_hip_get_stats whose image must not be trusted (the context's
// stream must be idle); `clear` consumes it (prt_hip_get_stats), download / gather only report it
int prt_sticky_error(prt_hip_ctx* c, bool clear);
