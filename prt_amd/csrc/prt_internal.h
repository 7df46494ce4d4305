// prt_internal.h -- what the translation units of libprt_hip.so share: the context behind the C-ABI handle, the error
// helpers and the build constants of the wavefront pipeline.  Not installed; include/prt_hip.h is the interface.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/prt_hip.h"
#include "prt_device.h"

#ifndef PRT_SHADE_WAVES
#define PRT_SHADE_WAVES 7 // waves per SIMD the shade kernel is compiled for (C3: 8: 514 ms, 7: 507, 6 and 5: 514, 4: 523)
#endif
#ifndef PRT_TRACE_BPC
#define PRT_TRACE_BPC 4 // persistent blocks per CU and trace kernel
#endif
#ifndef PRT_SIDE_STREAMS
#define PRT_SIDE_STREAMS 1 // side streams per pipeline for the three smaller trace kernels (1 or 3)
#endif
#ifndef PRT_PARTS
#define PRT_PARTS 2 // independent pipelines a pass is dealt to
#endif
#define PRT_WORK_WORDS 512 // control words per pipeline (wavefront pipeline: 4 + Q_COUNT * PRT_QSHARDS; frame kernel: PRT_CTRL_CURSORS)
#define PRT_STAT_SHARDS 64 // copies of the statistics counters, summed on read-back
#define PRT_STAT_STRIDE 32 // 64-bit words per copy (256 B apart)
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
    // PRT_PARTS independent pipelines (a pass dealt tile by tile) run side by side, each on a main stream (shade, scatter
    // trace) and a side stream (the other three trace kernels): stream, aux[0] | aux[1], aux[2] | ...  Two pipelines = four
    // streams = the four hardware queues a HIP process gets by default (measured on C3, whole frame / one rank's share of
    // 8: 2 pipelines 588 / 102 ms; 3: 590-660 / 110-138; 4: 595-657 / 106-135 depending on GPU_MAX_HW_QUEUES).
    hipStream_t aux[(1 + PRT_SIDE_STREAMS) * PRT_PARTS - 1] = {};
    hipEvent_t evFork[PRT_PARTS] = {}, evJoin[PRT_PARTS * PRT_SIDE_STREAMS] = {};
    hipEvent_t evStart = nullptr, evDone[PRT_PARTS] = {}; // pipelines 1.. against the main stream, per pass
    hipEvent_t evIn = nullptr, evOut = nullptr; // order the pipeline against a caller's stream
    void* wfBuffer = nullptr; // wavefront state + queues of one pass
    size_t wfBytes = 0;
    uint32_t wfGroups = 0;
    uint32_t* spill = nullptr;
    uint32_t spillThreads = 0;
    uint32_t lastRank = 0, lastNranks = 0, lastTile = 0; // of the last render (prt_hip_gather*)
    float* lastTarget = nullptr;                         // framebuffer of the last render (device)
    // image gather (prt_gather.hip): tile-major staging buffers and the RCCL communicator
    void* comm = nullptr;       // ncclComm_t
    bool commOwned = false;     // created by prt_hip_comm_init (destroyed with the context) or adopted from the host
    int commRank = 0, commSize = 0;
    float* packBuf = nullptr;   // this rank's tiles, tile-major
    size_t packFloats = 0;
    float* stageBuf = nullptr;  // root: the other ranks' tiles as received
    size_t stageFloats = 0;
    int blocksPerCU = 0;
    int frameBlocksPerCU = 0;  // resident blocks per CU of the frame kernel
    bool frameLaunched = false;
    void* frameArgs = nullptr;  // device copies of the frame kernel's argument block (ring)
    uint32_t frameArgSlot = 0;
    bool timed = false;
};


void prt_gather_release(prt_hip_ctx* c); // prt_gather.hip: frees the staging buffers and an owned communicator
