// prt_frame.h -- the frame kernel: ONE persistent launch renders a pixel rectangle (or a rank's tile share of it).
//
// Scheduling (DESIGN.md "Kernels").  The reference traces a pixel as samples/8 packets that share one xorshift32 stream,
// each packet a chain of 1 + maxDepth (shade -> trace) rounds (path_tracer.cpp:57-75, 124-301): per pixel that is a strictly
// sequential chain of (samples/8)*(1+maxDepth)+1 rounds, but pixels are independent.  So every workgroup keeps its OWN pool
// of pixel groups in flight and walks each group along its chain at the group's own pace -- there is no grid-wide step and
// no host loop:
//
//   pool      PRT_POOL_CHUNKS rows of 64 pixel groups (a group = the 8 path slots of one pixel).  A row is claimed from a
//             global cursor (64 consecutive tile-major work items = a quarter of a 16x16 tile) when all its groups are done.
//   pending   one LDS word per group: rays of the group still in flight.  0 = ready for its next shade round.
//   queues    four ray queues per block (primary packet rays, scatter rays, packet / single occlusion rays): rings in HBM that
//             only this block touches, head and tail in LDS.  An entry names the owner slot; the ray is rebuilt from its state.
//   roles     every wave alternates: SHADE (PRT_SHADERS roles per block, an LDS lock each; a role owns every PRT_SHADERS-th row of the
//             pool and appends to the block's queues behind a reservation, published in reservation order) sweeps the pending words of
//             its rows, runs the bounce of up to 8
//             ready groups per pass (8 lanes per group, wave ballot / popcount compaction and prefix-counted RNG stepping as
//             before), emits the next rays and sets pending = rays emitted;  TRACE takes the fullest queue and walks its rays
//             with persistent lanes (a lane refills from the queue the moment its ray finishes); a finished ray's lane stores
//             the hit and decrements the group's pending word.
//
// Everything a block shares lives on ONE CU: the hand-offs are LDS atomics plus workgroup-scope release/acquire (global
// stores drained with s_waitcnt vmcnt(0) before the LDS word that publishes them; waves of a workgroup share the CU's L1), so
// no agent-scope fence, no cross-XCD traffic and no inter-workgroup dependency exists -- a block that starts late or runs
// alone still completes.  The only global words are the row cursor, the statistics and the watchdog flag.
#pragma once

#ifndef PRT_POOL_CHUNKS
#define PRT_POOL_CHUNKS 16 // rows of 64 pixel groups a block keeps in flight (per wave: one row, as with round 2's 4 rows for 4 waves)
#endif
#ifndef PRT_FRAME_WAVES
#define PRT_FRAME_WAVES 8 // waves per SIMD the frame kernel is compiled for: 64 VGPRs, 80 SGPRs, 2 workgroups of 1024 threads per CU (78 KB of LDS each)
#endif
#ifndef PRT_SHADE_MIN
#define PRT_SHADE_MIN 192u // ready groups (of the pool's 1024) that make a wave at a decision point take a shade role (48: 392-402 ms, 96: 390, 192: 386 on C3)
#endif
#ifndef PRT_HOT_LDS
#define PRT_HOT_LDS 1 // keep the PRT_HOT_NODES records nearest the roots in LDS (256 records = 16 KB per workgroup of 16 waves)
#endif
#ifndef PRT_TRACE_PRIO
#define PRT_TRACE_PRIO 1
#endif
#ifndef PRT_SHADE_PRIO
#define PRT_SHADE_PRIO 0 // issue priority of the wave that holds the shade role (it PRODUCES the rays the tracing waves wait for)
#endif
#ifndef PRT_TRACE_MIN
#define PRT_TRACE_MIN 0u // a wave at a decision point shades first when the fullest queue holds fewer rays than this and PRT_TRACE_READY groups are ready (0 = rule off)
#endif
#ifndef PRT_TRACE_READY
#define PRT_TRACE_READY 8u
#endif
#ifndef PRT_SHADERS
#define PRT_SHADERS 4 // shade roles per block (a power of two): each owns every PRT_SHADERS-th row of the pool; all append to the block's queues
#endif
#define PRT_SHADER_GROUPS (PRT_POOL_GROUPS / PRT_SHADERS)
#ifndef PRT_CLAIM
#define PRT_CLAIM 128u // queue entries a wave reserves at a time
#endif
// Which queue a wave at a decision point traces (round 4).  Rounds 2-3: the fullest -- so every wave that came free took the scatter queue
// and they ran it dry together while hundreds of occlusion and primary rays waited for a wave (profiles/r04_experiments.txt E).  Now
// the queue with the most rays PER WAVE that would then be on it: score = 1 + len * PRT_BALANCE / (waves on it * PRT_BALANCE_W +
// PRT_BALANCE); the waves spread over the queues in proportion to their lengths and a queue without a wave gets the next one.
// C3 370 -> 357 ms; weights 8 .. 64 alike; PRT_BALANCE 0 = the fullest queue.
#ifndef PRT_BALANCE
#define PRT_BALANCE 1
#endif
#ifndef PRT_BALANCE_W
#define PRT_BALANCE_W 16
#endif
#ifndef PRT_DRAIN_READY
#define PRT_DRAIN_READY 1024u // ready groups at which tracing waves stop refilling (so that one of them comes free to shade): the whole pool
#endif
#ifndef PRT_SHADE_INLINE
#define PRT_SHADE_INLINE __noinline__
#endif
#ifndef PRT_ROLE_INLINE
#define PRT_ROLE_INLINE __noinline__
#endif
#ifndef PRT_WATCHDOG_TICKS
#define PRT_WATCHDOG_TICKS 200000000ull // a wave that has found no work for this long gives up: 2 s of s_memrealtime (the constant 100 MHz counter: the
                                        // deadline does not move with the shader clock or with how long an idle turn takes under load)
#endif
#define PRT_CTRL_CURSORS 256u // ctrl words: [256 + 32 * band] the 8 row cursors; behind them (PRT_WORK_WORDS..) the sticky error words: flag, watchdog reports
#define PRT_CHUNK 64u
#define PRT_POOL_GROUPS (PRT_POOL_CHUNKS * PRT_CHUNK)
#define PRT_POOL_SLOTS (PRT_POOL_GROUPS * 8u) // capacity of a block's ray queue per mode: one ray per slot and mode at most
static_assert((PRT_POOL_SLOTS & (PRT_POOL_SLOTS - 1u)) == 0u, "the ray rings index with a mask: PRT_POOL_CHUNKS must be a power of two");
#define PEND_DONE 0xffffffffu
#define PRT_NONE 0xffffffffu

struct FrameArgs {
    DevScene sc;
    DevCamera cam;
    prt_render_params p;
    uint32_t x0, y0, x1, y1;
    uint32_t tilesXImage;
    uint32_t rtx0, rty0, rtnx, rtny;
    uint32_t fullWidth, firstOwned;
    uint32_t totalWork;   // pixel groups (tile-major work items) of the launch
    uint32_t totalChunks; // rows of 64 work items
    uint32_t rowsPerBlock; // rows a block may hold at a time (<= PRT_POOL_CHUNKS; fewer when the launch is small)
    uint32_t spreadRows;   // rows are 8 clusters of 8 work items spread over the launch instead of 64 consecutive items
    float* rgb;
    unsigned long long* counters; // PRT_STAT_SHARDS copies of: rays, occl, nBox, nTri, nHit, nTap, nPx, overflow
    uint32_t* ctrl;               // row cursors, watchdog flag and report (PRT_CTRL_CURSORS)
    // per pool group (block * PRT_POOL_GROUPS + i)
    uint32_t* gRng;
    uint32_t* gInfo;  // packet | depth << 8 | alive << 16 | phase << 20 | alive at depth 0 << 24
    uint32_t* gPixel; // x | y << 16, 0xffffffff = no pixel (outside the rectangle / not this rank's tile)
    float4* gColor;
    // per pool slot (8 per group)
    float4* S0; // pos.xyz, bits(material of the surface the path stands on)
    float4* S1; // shading normal xyz, bits(slot flags)
    float4* S2; // direction of the ray in flight (primary or scatter) xyz
    float4* S3; // beta.xyz
    float4* S4; // result.xyz -- belongs to the SLOT, not the path
    float4* S5; // environment light only: lightDir[slot]
    float4* S6; //                         lightIntensity[slot]
    float4* hitA;   // t i j k
    uint2* hitB;    // primId meshId
    uint32_t* occl; // 1 = occluded
    uint32_t* qE;   // [block][Q_COUNT][PRT_POOL_SLOTS]: owner slot within the block | reverseBits << 26 | lightSet << 29
    uint32_t* spill;
    uint32_t spillStride;
};

struct __attribute__((aligned(16))) BlockState { // LDS, one per workgroup
    // one stack column per thread: PRT_STACK_LDS (12) references, or 6 (reference, entry distance) pairs for the packet traversal
    uint32_t stack[PRT_STACK_LDS * PRT_BLOCK];
    float hot[PRT_HOT_LDS ? PRT_HOT_NODES * 16 : 4]; // DevScene::hotNodes (16-byte aligned: read with ds_read_b128)
    uint32_t coop[(PRT_BLOCK / 64) * PRT_COOP_STRIDE]; // per wave: the pair table of the cooperative leaf rounds (prt_device.h)
    uint32_t pending[PRT_POOL_GROUPS];
    uint32_t readyList[PRT_POOL_GROUPS]; // the shade role's work list of one sweep
    uint32_t chunkLive[PRT_POOL_CHUNKS]; // groups of the row that are not done; 0 = the row can take new work
    uint32_t qTail[Q_COUNT], qHead[Q_COUNT];
    uint32_t qWaves[Q_COUNT];      // waves tracing each queue right now (PRT_BALANCE: the decision point shares the waves out by queue length)
    uint32_t qRes[Q_COUNT];        // PRT_SHADERS shade roles: entries RESERVED behind the tail (published to qTail in reservation order)
    uint32_t lock[PRT_SHADERS];    // shade role(s)
    uint32_t ready[PRT_SHADERS];   // groups with pending == 0, per shade role (a hint for the role decision, not a correctness word)
    uint32_t live;      // groups in the pool that are not done
    uint32_t exhausted; // the global cursor has no more rows
    uint32_t abort;
    uint32_t exited;    // waves that have left the role loop: the last one writes the block's statistics out
    uint32_t overflow;  // a traversal needed more than 64 stack entries
    uint32_t rays, occl, px;                        // stats.h:10-16
    unsigned long long nBox, nTri, nHit, nTap;      // counting build only
};
typedef __attribute__((address_space(3))) BlockState* BlockLds;

// The block's LDS: one object, reached from the kernel and from the role functions alike.
__device__ __forceinline__ BlockLds block_lds()
{
    __shared__ BlockState blockState;
    return (BlockLds)&blockState;
}

struct WaveStats {
    uint32_t rays, occl, px;
};

// LDS words are reached through address-space-3 pointers (ds_* instructions with constant offsets); a generic reference to
// the block's LDS would turn every access into a flat operation.
typedef __attribute__((address_space(3))) uint32_t lds_w;
__device__ __forceinline__ uint32_t lds_ld(const lds_w* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_ld_acq(const lds_w* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(lds_w* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st_rel(lds_w* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_add(lds_w* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_sub(lds_w* p, uint32_t v) { return __hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ bool lds_cas(lds_w* p, uint32_t expect, uint32_t v)
{
    return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// every global store of this wave has completed and is visible to the other waves of the workgroup (same CU, same L1)
__device__ __forceinline__ void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ __forceinline__ uint32_t bcast0(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint32_t block_ready(BlockLds B)
{
    uint32_t r = 0;
#pragma unroll
    for (int h = 0; h < PRT_SHADERS; h++) r += lds_ld(&B->ready[h]);
    return r;
}
__device__ __forceinline__ bool block_lock_free(BlockLds B)
{
    bool f = false;
#pragma unroll
    for (int h = 0; h < PRT_SHADERS; h++) f = f || lds_ld(&B->lock[h]) == 0u;
    return f;
}

// Pixel of work item w (tile-major order, row-major inside a tile, main.cpp:132-138); 0xffffffff when the item lies outside
// the rectangle or in a tile another rank owns.
__device__ __forceinline__ uint32_t work_item_pixel(const FrameArgs& A, uint32_t w)
{
    const uint32_t tile = A.p.tileSize, tile2 = tile * tile;
    const uint32_t tq = w / tile2, pix = w - tq * tile2;
    uint32_t gt;
    bool ok = true;
    if (A.fullWidth) {
        gt = A.firstOwned + tq * A.p.nranks;
    } else {
        uint32_t qx = tq % A.rtnx, qy = tq / A.rtnx;
        gt = (A.rty0 + qy) * A.tilesXImage + (A.rtx0 + qx);
        ok = (gt % A.p.nranks) == A.p.rank;
    }
    uint32_t tx = gt % A.tilesXImage, ty = gt / A.tilesXImage;
    uint32_t x = tx * tile + pix % tile, y = ty * tile + pix / tile;
    if (x < A.x0 || x > A.x1 || y < A.y0 || y > A.y1) ok = false;
    return ok ? (x | (y << 16)) : 0xffffffffu;
}

// The roles are real (not inlined) functions, so that each gets the registers of a kernel of its own size.  They reach the
// launch's arguments through the kernel-argument segment, whose address the kernel hands them as a plain integer (a callable
// function has no kernel-argument pointer of its own): rebuilt from readfirstlane halves it is a uniform pointer into constant
// memory, so fields arrive by scalar loads where they are used, and the data pointers in them are dereferenced as global
// addresses (gld / gst / nt_*) with a scalar base -- the code a kernel with by-value arguments gets.
typedef const __attribute__((address_space(4))) FrameArgs* FrameKernargs;
__device__ __forceinline__ uint64_t frame_kernarg_bits() { return (uint64_t)(const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ const FrameArgs& frame_args(uint64_t bits)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32));
    return *(const FrameArgs*)(FrameKernargs)(((uint64_t)hi << 32) | lo);
}

// Counting build: a role call's event counts go to the block's 64-bit sums.
__device__ __forceinline__ void block_count_traffic(BlockLds B, const Traffic& tr)
{
    unsigned long long b = tr.nBox, t = tr.nTri, h = tr.nHit, p = tr.nTap;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        b += (unsigned long long)__shfl_xor((long long)b, o, 64);
        t += (unsigned long long)__shfl_xor((long long)t, o, 64);
        h += (unsigned long long)__shfl_xor((long long)h, o, 64);
        p += (unsigned long long)__shfl_xor((long long)p, o, 64);
    }
    if ((threadIdx.x & 63u) == 0u) {
        typedef __attribute__((address_space(3))) unsigned long long lds_q;
        if (b) __hip_atomic_fetch_add((lds_q*)&B->nBox, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t) __hip_atomic_fetch_add((lds_q*)&B->nTri, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (h) __hip_atomic_fetch_add((lds_q*)&B->nHit, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p) __hip_atomic_fetch_add((lds_q*)&B->nTap, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ---------------------------------------------------------------------------------------------------------------- shade
// One shade round of up to 8 pixel groups (lanes 8j..8j+7 = the 8 path slots of group j).  P = the group's index into the
// pool state, PRT_NONE for lanes without a group.  Consumes the hits of the group's last rays, runs the bounce of
// path_tracer.cpp:124-293 and appends the next rays to the block's queues (`tails` = the queue tails, owned by the caller,
// who holds the shade lock and publishes them).  Returns, in every lane of the group, the group's new pending word: the
// number of rays emitted, or PEND_DONE when the pixel has been written.
template <bool COUNT, bool ENV>
__device__ __noinline__ uint32_t shade_pass(uint64_t kargs, uint32_t P, uint32_t poolLocal)
{
    // a function of its own: its registers are those of a shade kernel, whatever the caller keeps live around it
    const FrameArgs& A = frame_args(kargs);
    const BlockLds B = block_lds();
    uint32_t* blockQ = A.qE + (size_t)blockIdx.x * Q_COUNT * PRT_POOL_SLOTS;
    WaveStats ws{0, 0, 0};
    Traffic tr{};
    const uint32_t lane = threadIdx.x & 63u, slot = lane & 7u, gbase = lane & ~7u;
    const bool inRange = P != PRT_NONE;
    const uint32_t g = inRange ? P : 0u;
    const uint32_t gs = g * 8u + slot;
    const DevScene& sc = A.sc;
    const DevCamera& cam = A.cam;
    const uint32_t samples = A.p.samples, maxDepth = A.p.maxDepth, rrDepth = A.p.rrDepth, packets = samples / 8u;
    const float kPi = 3.14159265358979323846f;
    const float kFar = 2.0f * sc.radius; // path_tracer.cpp:192
    const uint32_t lowerMask = (1u << slot) - 1u;

    uint32_t info = inRange ? gld(&A.gInfo[g]) : ((uint32_t)PH_DONE << 20);
    uint32_t phase = (info >> 20) & 0xfu, pk = info & 0xffu, depth = (info >> 8) & 0xffu, alive = (info >> 16) & 0xfu;
    uint32_t alive0 = (info >> 24) & 0xfu; // slots that have held a path in this packet: the others' result is still 0
    const uint32_t aliveAtEntry = (phase == PH_WAIT_BOUNCE) ? alive : 0u;
    uint32_t rng = 0, pixel = 0xffffffffu;
    Vec3 color = mk3(0, 0, 0);
    if (inRange) {
        rng = gld(&A.gRng[g]);
        pixel = gld(&A.gPixel[g]);
        float4 c = gld4(&A.gColor[g]);
        color = mk3(c.x, c.y, c.z);
    }
    const uint32_t x = pixel & 0xffffu, y = pixel >> 16;

    // slot state
    Vec3 pos = mk3(0, 0, 0), rayDir = mk3(0, 0, 0), normal = mk3(0, 0, 0), beta = mk3(1, 1, 1), result = mk3(0, 0, 0), ndir = mk3(0, 0, 0);
    Surface props{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
    uint32_t material = 0, sflags = 0, lightSet = 0;
    Vec3 envL = mk3(0, 0, 0), envI = mk3(0, 0, 0); // ENV: this bounce's sampled light, when the slot samples one
    bool envSampled = false;

    bool needBounce = false, needEnd = false, needCamera = false;
    bool emitPrimary = false, emitShadow = false, emitScatter = false, shadowPacket = false;
    bool statPx = false, statSamples = false;
    uint32_t reverseBits = 0;

    if (phase == PH_START) {
        if (pixel == 0xffffffffu || packets == 0u) {
            phase = PH_DONE;
            if (pixel != 0xffffffffu && slot == 0) { // samples < 8: the reference still writes 0/samples
                uint32_t* px = (uint32_t*)(A.rgb + ((size_t)x + (size_t)y * cam.width) * 3);
                const uint32_t zero = asu(A.p.exposure * (0.0f / (float)samples));
                gst(px, zero);
                gst(px + 1, zero);
                gst(px + 2, zero);
                statPx = true;
                statSamples = true;
            }
        } else {
            needCamera = true;
            if (slot == 0) statSamples = true; // path_tracer.cpp:62
        }
    } else if (phase == PH_WAIT_PRIMARY) {
        // ---- ComputeRadiance set-up (path_tracer.cpp:81-120): hits gathered into slots 0..alive-1 in lane order
        float4 ha = nt_load4(&A.hitA[gs]);
        uint2 hb = gld2u(&A.hitB[gs]);
        float4 s2 = nt_load4(&A.S2[gs]); // the primary ray's direction
        Vec3 pdir = mk3(s2.x, s2.y, s2.z), porg = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
        DevHit h{ha.x, ha.y, ha.z, ha.w, hb.x, hb.y};
        bool isHit = h.t != -1.0f;
        Surf5 sv{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
        Vec3 snormal = mk3(0, 0, 0), spos = mk3(0, 0, 0);
        if (isHit) {
            Surface s;
            get_surface<COUNT>(sc, h, s, tr);
            sv = Surf5{s.normal, s.uv, s.mat, s.prim};
            snormal = sample_bump<COUNT>(sc, s.mat, s, tr);
            spos = add3(scale3(h.t, pdir), porg);
        }
        uint32_t hm = group_ballot(isHit, gbase);
        alive = __popc(hm);
        uint32_t src = gbase + ((slot < alive) ? nth_set(hm, slot) : slot);
        props.normal = sh3(sv.normal, src);
        props.uv = Vec2{shf(sv.uv.x, src), shf(sv.uv.y, src)};
        props.mat = shu(sv.mat, src);
        props.prim = shu(sv.prim, src);
        material = props.mat;
        normal = sh3(snormal, src);
        pos = sh3(spos, src);
        rayDir = sh3(pdir, src);
        beta = mk3(1.0f, 1.0f, 1.0f);
        result = mk3(0.0f, 0.0f, 0.0f);
        lightSet = 0;
        depth = 0;
        alive0 = alive;
        if (alive != 0u && depth < maxDepth) needBounce = true;
        else needEnd = true;
    } else if (phase == PH_WAIT_BOUNCE) {
        uint32_t pmat = 0;
        if (slot < alive) { // dead slots carry nothing
            float4 s0 = nt_load4(&A.S0[gs]), s1 = nt_load4(&A.S1[gs]), s2 = nt_load4(&A.S2[gs]), s3 = nt_load4(&A.S3[gs]);
            pos = mk3(s0.x, s0.y, s0.z);
            pmat = asu(s0.w);
            normal = mk3(s1.x, s1.y, s1.z);
            sflags = asu(s1.w);
            ndir = mk3(s2.x, s2.y, s2.z);
            beta = mk3(s3.x, s3.y, s3.z);
            float4 s4 = nt_load4(&A.S4[gs]);
            result = mk3(s4.x, s4.y, s4.z);
        }
        props.mat = pmat;
        lightSet = (sflags & SLOT_LIGHT_SET) ? 1u : 0u;
        // ---- light contribution of the previous bounce (path_tracer.cpp:226-231, 246-249)
        if (sflags & SLOT_HAS_SHADOW) {
            if (gld(&A.occl[gs]) == 0u) {
                Vec3 lightDir = mk3(0, 0, 0), lightInt = mk3(0, 0, 0);
                if (lightSet) {
                    if (ENV) {
                        float4 l5 = nt_load4(&A.S5[gs]), l6 = nt_load4(&A.S6[gs]);
                        lightDir = mk3(l5.x, l5.y, l5.z);
                        lightInt = mk3(l6.x, l6.y, l6.z);
                    } else {
                        lightDir = mk3(sc.lightDir[0], sc.lightDir[1], sc.lightDir[2]);
                        lightInt = mk3(sc.lightIntensity[0], sc.lightIntensity[1], sc.lightIntensity[2]);
                    }
                }
                Vec3 lr = div3s(scale3(std_max(dot3(lightDir, normal), 0.0f), lightInt), kPi);
                result = add3(result, mul3(beta, lr));
            }
        }
        // ---- scatter hits, ordered compaction into slot ci (path_tracer.cpp:281-293)
        bool hitNext = false;
        Surface ns{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
        Vec3 npos = mk3(0, 0, 0);
        if (sflags & SLOT_SURVIVE) {
            float4 ha = nt_load4(&A.hitA[gs]);
            uint2 hb = gld2u(&A.hitB[gs]);
            DevHit nh{ha.x, ha.y, ha.z, ha.w, hb.x, hb.y};
            if (nh.t != -1.0f) {
                hitNext = true;
                get_surface<COUNT>(sc, nh, ns, tr);
                npos = add3(scale3(nh.t, ndir), pos);
            }
        }
        uint32_t nm = group_ballot(hitNext, gbase);
        uint32_t nAlive = __popc(nm);
        if (nAlive == 0u) {
            needEnd = true; // path_tracer.cpp:295
        } else {
            uint32_t ci = __popc(nm & lowerMask);
            uint32_t smat = 0;
            Vec3 snorm = mk3(0, 0, 0);
            Surf5 nv{mk3(0, 0, 0), Vec2{0, 0}, 0, 0};
            if (hitNext) {
                // materials[ci] = props[i].material reads the slot's PREVIOUS surface unless ci == i (:286-288)
                smat = (ci == slot) ? ns.mat : props.mat;
                snorm = sample_bump<COUNT>(sc, smat, ns, tr);
                nv = Surf5{ns.normal, ns.uv, ns.mat, ns.prim};
            }
            // beta[ci] = beta[i]/(1-q) is only written under Russian roulette (:263); q is a function of beta
            Vec3 betaNew = beta;
            if (depth > rrDepth) {
                float q = std_max(0.05f, 1.0f - length3(beta));
                betaNew = div3s(beta, 1.0f - q);
            }
            uint32_t src2 = gbase + ((slot < nAlive) ? nth_set(nm, slot) : slot);
            props.normal = sh3(nv.normal, src2);
            props.uv = Vec2{shf(nv.uv.x, src2), shf(nv.uv.y, src2)};
            props.mat = shu(nv.mat, src2);
            props.prim = shu(nv.prim, src2);
            material = shu(smat, src2);
            normal = sh3(snorm, src2);
            pos = sh3(npos, src2);
            rayDir = sh3(ndir, src2);
            Vec3 bmoved = sh3(betaNew, src2);
            if (depth > rrDepth && slot < nAlive) beta = bmoved;
            alive = nAlive;
            depth++;
            if (depth < maxDepth) needBounce = true;
            else needEnd = true;
        }
    }

    if (needBounce) {
        // ---- one bounce (path_tracer.cpp:131-190 and the Russian roulette of :258-265)
        const bool active = slot < alive;
#ifdef PRT_PROFILE
        { // lanes of the pass that carry a path through its bounce (of 64)
            const unsigned long long carrying = __ballot(slot < alive); // (this branch is per group: the ballot is of the lanes in it)
            if (lane == (uint32_t)__builtin_ctzll(__ballot(true))) {
                unsigned long long* C = A.counters + (size_t)(blockIdx.x % PRT_STAT_SHARDS) * PRT_STAT_STRIDE;
                atomicAdd(&C[118], 1ull);
                atomicAdd(&C[119], (unsigned long long)__popcll(carrying));
            }
        }
#endif
        uint32_t rtype = 2u;
        if (active) {
            const float4* mp = sc.mats + PRT_MAT_STRIDE * (size_t)material;
            float4 m0 = gld4(mp), m1 = gld4(mp + 1);
            rtype = asu(m0.w);
            if (m1.x != 0.0f) result = add3(result, mul3(beta, mk3(m1.x, m1.y, m1.z))); // :137-139
        }
        const bool draws = active && (rtype == 0u || rtype == 1u);
        uint32_t dm = group_ballot(draws, gbase);
        uint32_t pre = 2u * __popc(dm & lowerMask), tot = 2u * __popc(dm);
        if (ENV) { // a diffuse slot draws two more for InfiniteAreaLight::sample (:164-167), after its r2, r1
            uint32_t em = group_ballot(active && rtype == 0u, gbase);
            pre += 2u * __popc(em & lowerMask);
            tot += 2u * __popc(em);
        }
        uint32_t s = rng, r2b = 0, r1b = 0, uxb = 0, uyb = 0;
        for (uint32_t j = 0; j < tot; j++) {
            s = xorshift32(s);
            if (j == pre) r2b = s;
            if (j == pre + 1u) r1b = s;
            if (ENV && j == pre + 2u) uxb = s;
            if (ENV && j == pre + 3u) uyb = s;
        }
        rng = s;
        Vec3 nextDir = mk3(0, 0, 0);
        bool wantLight = false;
        if (draws) {
            Vec3 dd = diffuse_dir(normal, rng_to_float(r2b), rng_to_float(r1b));
            if (rtype == 0u) {
                nextDir = dd;
                beta = mul3(beta, sample_diffuse<COUNT>(sc, material, props.uv, tr)); // :162
                if (ENV) { // :164-167
                    env_sample<COUNT>(sc, rng_to_float(uxb), rng_to_float(uyb), envL, envI, tr);
                    envSampled = true;
                    lightSet = 1u;
                    wantLight = true;
                } else if (sc.hasLight) { // :168-172
                    lightSet = 1u;
                    wantLight = true;
                }
            } else {
                Vec3 reflectDir = sub3(rayDir, scale3(dot3(normal, rayDir), scale3(2.0f, normal))); // :186
                nextDir = add3(scale3(0.9f, reflectDir), scale3(0.1f, dd));
            }
        }
        const bool directLighting = group_ballot(wantLight, gbase) != 0u;
        sflags = 0;
        if (directLighting && active) { // :196-252: every alive path gets an occlusion ray
            emitShadow = true; // the occlusion traversal builds org = pos + kFar*L, dir = -L from the slot's state
            shadowPacket = (alive & 0xfu) > 2u; // :198
            sflags |= SLOT_HAS_SHADOW;
            // (counted below: one ray, one occlusion ray)
        }
        bool survive = active;
        if (depth > rrDepth) { // one draw per alive slot, in slot order
            uint32_t s2 = rng, ub = 0;
            for (uint32_t j = 0; j < alive; j++) {
                s2 = xorshift32(s2);
                if (j == slot) ub = s2;
            }
            rng = s2;
            if (active) {
                float q = std_max(0.05f, 1.0f - length3(beta));
                if (rng_to_float(ub) < q) survive = false;
            }
        }
        if (survive) {
            ndir = normalize3(nextDir); // :267
            emitScatter = true;
            sflags |= SLOT_SURVIVE;
        }
        phase = PH_WAIT_BOUNCE;
    }

    if (needEnd) {
        // ---- sum of result[0..7] in slot order (path_tracer.cpp:303-307), color += (:71).  Slots whose path ended in an earlier
        // round left their result in memory.
        if (slot >= aliveAtEntry && slot < alive0 && phase == PH_WAIT_BOUNCE) {
            float4 s4 = nt_load4(&A.S4[gs]);
            result = mk3(s4.x, s4.y, s4.z);
        }
        Vec3 res = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t l = 0; l < 8; l++) res = add3(res, sh3(result, gbase + l));
        color = add3(color, res);
        pk++;
        if (pk < packets) {
            needCamera = true;
        } else {
            Vec3 c = scale3(A.p.exposure, div3s(color, (float)samples)); // path_tracer.cpp:28, image.cpp:45
            if (slot == 0) {
                uint32_t* px = (uint32_t*)(A.rgb + ((size_t)x + (size_t)y * cam.width) * 3);
                gst(px, asu(c.x));
                gst(px + 1, asu(c.y));
                gst(px + 2, asu(c.z));
                statPx = true;
            }
            phase = PH_DONE;
        }
    }

    if (needCamera) {
        DevRay pr;
        Vec3 avgDir;
        camera_packet(cam, rng, x, y, slot, gbase, pr, avgDir);
        reverseBits = (avgDir.x < 0.0f ? 1u : 0u) | (avgDir.y < 0.0f ? 2u : 0u) | (avgDir.z < 0.0f ? 4u : 0u);
        ndir = pr.dir;
        emitPrimary = true;
        phase = PH_WAIT_PRIMARY;
    }

    // ---- the rays of the next round go to the block's queues: lane rank by wave ballot + popcount behind a RESERVATION of the
    // pass's entries (PRT_SHADERS shade roles append to the same queues: qRes is an LDS atomic, the tails are published below in
    // reservation order)
    const bool want[Q_COUNT] = {emitPrimary, emitScatter, emitShadow && shadowPacket, emitShadow && !shadowPacket};
    const uint32_t owner = poolLocal * 8u + slot; // slot index inside the block's pool (< PRT_POOL_SLOTS)
    uint32_t emitted = 0, newTail[Q_COUNT], qBase[Q_COUNT];
#pragma unroll
    for (int q = 0; q < Q_COUNT; q++) {
        const unsigned long long mask = __ballot(want[q]);
        newTail[q] = 0;
        qBase[q] = 0;
        if (mask == 0ull) continue; // wave-uniform
        uint32_t tail;
        if (PRT_SHADERS > 1) { // another wave may append too: reserve the range (lane 0), publish below in reservation order
            uint32_t t = 0;
            if (lane == 0) t = lds_add(&B->qRes[q], (uint32_t)__popcll(mask));
            tail = bcast0(t);
        } else {
            tail = bcast0(lds_ld(&B->qTail[q]));
        }
        qBase[q] = tail;
        const uint32_t rank = mbcnt64(mask);
        if (want[q]) {
            const uint32_t idx = (tail + rank) & (PRT_POOL_SLOTS - 1u);
            const uint32_t bits = owner | (q == Q_PRIMARY ? (reverseBits << 26) : 0u) | (q >= Q_OCC_PACKET ? (lightSet << 29) : 0u);
            nt_store(&blockQ[q * PRT_POOL_SLOTS + idx], bits);
        }
        newTail[q] = tail + (uint32_t)__popcll(mask);
        emitted += (uint32_t)__popc(((uint32_t)(mask >> gbase)) & 0xffu); // of this lane's group
    }

    // ---- store state
    if (inRange && ((info >> 20) & 0xfu) != PH_DONE) {
        if (phase == PH_WAIT_BOUNCE && slot < alive) {
            nt_store4(&A.S0[gs], make_float4(pos.x, pos.y, pos.z, asf(props.mat)));
            nt_store4(&A.S1[gs], make_float4(normal.x, normal.y, normal.z, asf(sflags | (lightSet ? SLOT_LIGHT_SET : 0u))));
            nt_store4(&A.S2[gs], make_float4(ndir.x, ndir.y, ndir.z, 0.0f));
            nt_store4(&A.S3[gs], make_float4(beta.x, beta.y, beta.z, 0.0f));
            if (ENV && envSampled) {
                nt_store4(&A.S5[gs], make_float4(envL.x, envL.y, envL.z, 0.0f));
                nt_store4(&A.S6[gs], make_float4(envI.x, envI.y, envI.z, 0.0f));
            }
        }
        // a slot's result is stored while the slot is alive and once more in the round its path ends
        if (phase == PH_WAIT_BOUNCE && slot < (aliveAtEntry > alive ? aliveAtEntry : alive))
            nt_store4(&A.S4[gs], make_float4(result.x, result.y, result.z, 0.0f));
        if (phase == PH_WAIT_PRIMARY) nt_store4(&A.S2[gs], make_float4(ndir.x, ndir.y, ndir.z, 0.0f));
        if (slot == 0) {
            gst(&A.gInfo[g], (pk & 0xffu) | ((depth & 0xffu) << 8) | ((alive & 0xfu) << 16) | (phase << 20) | ((alive0 & 0xfu) << 24));
            gst(&A.gRng[g], rng);
            gst4(&A.gColor[g], make_float4(color.x, color.y, color.z, 0.0f));
        }
    }
    // ---- statistics (wave-uniform sums: stats.h:10-16; a scatter ray and a shadow ray each count as a ray, path_tracer.cpp:219, 276)
    {
        const uint32_t nShadow = (uint32_t)__popcll(__ballot(emitShadow)), nScatter = (uint32_t)__popcll(__ballot(emitScatter));
        ws.rays = nShadow + nScatter + samples * (uint32_t)__popcll(__ballot(statSamples));
        ws.occl = nShadow;
        ws.px = (uint32_t)__popcll(__ballot(statPx));
        if (lane == 0) {
            if (ws.rays) lds_add(&B->rays, ws.rays);
            if (ws.occl) lds_add(&B->occl, ws.occl);
            if (ws.px) lds_add(&B->px, ws.px);
        }
        if (COUNT) block_count_traffic(B, tr);
    }
    // ---- publish: state and queue entries have to be complete before the pending words, and those before the tails
    const uint32_t np = phase == PH_DONE ? PEND_DONE : emitted;
    wg_release();
    if (inRange && slot == 0u) lds_st(&B->pending[poolLocal], np);
#pragma unroll
    for (int q = 0; q < Q_COUNT; q++)
        if (newTail[q] != qBase[q] && lane == 0u) {
            if (PRT_SHADERS > 1) {
                // in reservation order: the tail moves over this pass's entries only when everything reserved before them is published
                // (the other shade role publishes its queues in the same ascending order, so the two waits cannot form a cycle);
                // bounded: a bug here must end as the watchdog's error, never as a hung GPU
                uint32_t spins = 0;
                while (lds_ld_acq(&B->qTail[q]) != qBase[q] && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1u << 24)) {
                    // the entries reserved before this pass's were never published: end the launch as the watchdog does (the host
                    // gets PRT_HIP_ELAUNCH, not an incomplete image with PRT_HIP_OK) and leave the tail where it is -- moving it
                    // would hand tracing waves the unwritten words in between
                    lds_st(&B->abort, 1u);
                    atomicOr(&A.ctrl[PRT_WORK_WORDS], 1u);
                    continue;
                }
            }
            lds_st_rel(&B->qTail[q], newTail[q]);
        }
    return np;
}

// ---------------------------------------------------------------------------------------------------------------- trace
// Persistent lanes over one of the block's ray queues.  A lane takes a new ray the moment its own finishes; the wave leaves
// when the queue is empty and its last ray is done.  A finished ray's result is stored to the owner slot and the owner
// group's pending word is decremented one loop turn later -- by then the step phase in between has waited for younger
// loads, so the store has completed (vector memory operations of a wave retire in order) and the release fence is free.
template <int MODE, bool COUNT>
__device__ PRT_ROLE_INLINE void trace_queue(uint64_t kargs)
{
    constexpr int NLDS = (MODE == PRT_MODE_PACKET) ? PRT_STACK_LDS_PACKET : PRT_STACK_LDS;
    const FrameArgs& A = frame_args(kargs);
    const BlockLds B = block_lds();
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t slotBase = blockIdx.x * PRT_POOL_SLOTS;
    const uint32_t* blockQ = A.qE + (size_t)blockIdx.x * Q_COUNT * PRT_POOL_SLOTS;
    const StackT<NLDS> st{(lds_u32*)&B->stack[tid], (lds_f32*)&B->stack[NLDS * PRT_BLOCK + tid], A.spill,
                          A.spillStride, PRT_HOT_LDS ? (const lds_f4*)&B->hot[0] : nullptr, (lds_u32*)&B->coop[(tid >> 6) * PRT_COOP_STRIDE]};
    const DevScene& sc = A.sc;
    const Vec3 camPos = mk3(A.cam.pos[0], A.cam.pos[1], A.cam.pos[2]);
    const Vec3 sceneLight = mk3(sc.lightDir[0], sc.lightDir[1], sc.lightDir[2]);
    const float kFar = 2.0f * sc.radius; // path_tracer.cpp:192
    Traffic tr{};
    uint32_t overflow = 0;
    // Every field is written before the loop: lanes that hold no ray keep these zeros (or their last ray's values), so nothing the
    // loop reads -- the cooperative leaf rounds hand EVERY lane's ray constants to ds_bpermute as source operands -- is ever
    // indeterminate (reading an indeterminate value is undefined behaviour, and the optimiser once used it)
    Tracer T{};
    T.ref = PRT_REF_NONE;
    T.sp = 0;
    T.m = 0;
    bool active = false;
    uint32_t owner = 0;      // pool slot (inside the block) of the lane's ray
    uint32_t sig = PRT_NONE; // pool group whose pending word this lane still has to decrement
    uint32_t held[PRT_CLAIM / 64u];     // queue entries the wave has claimed and not yet handed to a lane
    uint32_t heldNext = 0, heldEnd = 0; // which of them are left (wave-uniform)
#pragma unroll
    for (uint32_t j = 0; j < PRT_CLAIM / 64u; j++) held[j] = 0;
    __builtin_amdgcn_s_setprio(PRT_TRACE_PRIO); // tracing waves sit on dependent loads: they issue first, shading fills in
#ifdef PRT_PROFILE
    unsigned long long pTurns = 0, pLanes = 0, pClaims = 0, pEmptyClaims = 0, pRefills = 0, pRefillLanes = 0, pT0 = __builtin_amdgcn_s_memtime();
    unsigned long long pTRefill = 0, pTEnter = 0, pTStep = 0;
    unsigned long long pShortTurns = 0, pShortLanes = 0, pShortOthers = 0, pShortReady = 0, pShortLock = 0;
#endif
    for (;;) {
        // ---- 1. refill.  The wave takes PRT_CLAIM entries of the queue at a time and hands them to its lanes as they come
        // free: consecutive entries are rays of neighbouring pixels at the same bounce, and lanes that walk the same part of
        // the tree step together.  The entries are READ when they are claimed and wait in registers (held[]): the ring's
        // capacity argument -- at most one ray per slot and mode is out -- only holds for entries consumed in order.
#ifdef PRT_PROFILE
        const unsigned long long pTa = __builtin_amdgcn_s_memtime();
#endif
        const unsigned long long need = __ballot(!active);
        if (need) {
            const uint32_t k = (uint32_t)__popcll(need);
            const uint32_t avail = heldEnd - heldNext; // wave-uniform
            const uint32_t r = mbcnt64(need);
            uint32_t bits = 0;
            bool gotRay = false;
            if (avail != 0u) { // hand out what the wave holds: entry heldNext + r sits in lane (e & 63) of held[e >> 6]
                const uint32_t e = heldNext + r;
#pragma unroll
                for (uint32_t j = 0; j < PRT_CLAIM / 64u; j++) {
                    const uint32_t v = shu(held[j], e & 63u);
                    if ((e >> 6) == j) bits = v;
                }
                gotRay = r < avail;
            }
            if (avail < k) { // wave-uniform: the held entries are used up, take the next range
                uint32_t newBase = 0, newGot = 0;
                if (lane == 0 && !(block_ready(B) >= PRT_DRAIN_READY && block_lock_free(B))) {
                    for (int tries = 0; tries < 16; tries++) {
                        const uint32_t t = lds_ld_acq(&B->qTail[MODE]), h = lds_ld(&B->qHead[MODE]);
                        const int32_t queued = (int32_t)(t - h);
                        if (queued <= 0) break;
                        const uint32_t take = PRT_CLAIM < (uint32_t)queued ? PRT_CLAIM : (uint32_t)queued;
                        if (lds_cas(&B->qHead[MODE], h, h + take)) {
                            newBase = h;
                            newGot = take;
                            break;
                        }
                    }
                }
                newBase = bcast0(newBase);
                newGot = bcast0(newGot);
#pragma unroll
                for (uint32_t j = 0; j < PRT_CLAIM / 64u; j++) {
                    const uint32_t e = j * 64u + lane;
                    held[j] = e < newGot ? nt_load(&blockQ[MODE * PRT_POOL_SLOTS + ((newBase + e) & (PRT_POOL_SLOTS - 1u))]) : 0u;
                }
                const uint32_t want = k - avail, used = want < newGot ? want : newGot;
                {
                    const uint32_t e = r - avail; // meaningful for the lanes that take from the new range; every lane shuffles
                    uint32_t fresh = 0;
#pragma unroll
                    for (uint32_t j = 0; j < PRT_CLAIM / 64u; j++) {
                        const uint32_t v = shu(held[j], e & 63u);
                        if ((e >> 6) == j) fresh = v;
                    }
                    if (r >= avail && e < newGot) {
                        bits = fresh;
                        gotRay = true;
                    }
                }
                heldNext = used;
                heldEnd = newGot;
            } else {
                heldNext += k;
            }
#ifdef PRT_PROFILE
            {   // when lanes stay empty after the refill: where is the block's work?
                const uint32_t short_ = (uint32_t)__popcll(__ballot(!active && !gotRay));
                if (short_) {
                    uint32_t others = 0;
                    for (int q = 0; q < Q_COUNT; q++)
                        if (q != MODE) {
                            const int32_t d = (int32_t)(bcast0(lds_ld(&B->qTail[q])) - bcast0(lds_ld(&B->qHead[q])));
                            others += d > 0 ? (uint32_t)d : 0u;
                        }
                    pShortTurns++;
                    pShortLanes += short_;
                    pShortOthers += others;
                    pShortReady += bcast0(block_ready(B));
                    pShortLock += bcast0(block_lock_free(B) ? 0u : 1u);
                }
            }
            pClaims++;
            if (!__any(gotRay)) pEmptyClaims++;
            else {
                pRefills++;
                pRefillLanes += (unsigned long long)__popcll(__ballot(!active && gotRay));
            }
#endif
            if (!active && gotRay) {
                owner = bits & 0x3ffffffu;
                const uint32_t gs = slotBase + owner;
                Vec3 org, dir;
                float maxT;
                uint32_t rev = (bits >> 26) & 7u;
                if (MODE == PRT_MODE_PACKET) {
                    float4 s2 = nt_load4(&A.S2[gs]);
                    org = camPos;
                    dir = mk3(s2.x, s2.y, s2.z);
                    maxT = 100000.0f; // camera.cpp:64
                } else if (MODE == PRT_MODE_SINGLE) {
                    float4 s0 = nt_load4(&A.S0[gs]), s2 = nt_load4(&A.S2[gs]);
                    org = mk3(s0.x, s0.y, s0.z);
                    dir = mk3(s2.x, s2.y, s2.z);
                    maxT = kFar; // path_tracer.cpp:270
                } else {
                    float4 s0 = nt_load4(&A.S0[gs]);
                    Vec3 L = mk3(0.0f, 0.0f, 0.0f);
                    if ((bits >> 29) & 1u) {
                        if (sc.hasEnv) {
                            float4 l5 = nt_load4(&A.S5[gs]);
                            L = mk3(l5.x, l5.y, l5.z);
                        } else {
                            L = sceneLight;
                        }
                    }
                    org = add3(mk3(s0.x, s0.y, s0.z), scale3(kFar, L)); // path_tracer.cpp:210, 237
                    dir = mk3(-L.x, -L.y, -L.z);
                    maxT = kFar - 0.0008f; // :209, 236
                }
                tracer_begin<MODE>(T, org, dir, maxT, rev);
                // NaN rays: see trace_loop (answered at once in the timed build, walked in the counting build)
                if (!COUNT && !(org.x == org.x && org.y == org.y && org.z == org.z && dir.x == dir.x && dir.y == dir.y && dir.z == dir.z))
                    T.m = sc.bvhCount - 1u;
                active = true;
            }
        }
        // ---- 2. rays that finished in the previous turn: their result stores were issued before this turn's refill loads, which
        // the refill has waited for (vector memory operations of a wave retire in order), so the release below is free -> tell the group
        if (__any(sig != PRT_NONE)) {
            wg_release();
            if (sig != PRT_NONE) {
                if (lds_sub(&B->pending[sig], 1u) == 1u) lds_add(&B->ready[(sig / PRT_CHUNK) % PRT_SHADERS], 1u);
                sig = PRT_NONE;
            }
        }
#ifdef PRT_PROFILE
        const unsigned long long pTb = __builtin_amdgcn_s_memtime();
        pTRefill += pTb - pTa;
#endif
        if (!__any(active)) break;
#ifdef PRT_PROFILE
        pTurns++;
        pLanes += (unsigned long long)__popcll(__ballot(active));
#endif
        // ---- 3. new rays enter their first BVH (one that misses every root box is finished at once), 4. step, 5. rays that have
        // left their last BVH store their result and give the lane up NOW, so that the next turn's refill -- the next thing the
        // wave does -- hands it a new ray; retiring them behind the refill (as the first version of this loop did) left every
        // such lane empty for a whole step phase: 18 of 64 lanes in a scatter-ray node round (profiles/r03_experiments.txt).
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            if (active && T.ref == PRT_REF_NONE) {
                if (!tracer_next_bvh<MODE, COUNT>(sc, T, tr)) {
                    const uint32_t gs = slotBase + owner;
                    if (MODE == PRT_MODE_PACKET || MODE == PRT_MODE_SINGLE) {
                        if (T.hit.t == T.maxT) T.hit.t = -1.0f; // setMissForMaxT, scene.cpp:62
                        nt_store4(&A.hitA[gs], make_float4(T.hit.t, T.hit.i, T.hit.j, T.hit.k));
                        gst2u(&A.hitB[gs], make_uint2(T.hit.primId, T.hit.meshId));
                    } else {
                        gst(&A.occl[gs], T.occ ? 1u : 0u);
                    }
                    sig = owner >> 3;
                    active = false;
                }
            }
#ifdef PRT_PROFILE
            const unsigned long long pTc = __builtin_amdgcn_s_memtime();
            if (pass == 0) pTEnter += pTc - pTb;
#endif
            if (pass == 0)
                trace_step_phase<MODE, COUNT>(sc, T, active, st, tr, overflow, [&]() -> uint32_t {
                    const int32_t d = (int32_t)(bcast0(lds_ld(&B->qTail[MODE])) - bcast0(lds_ld(&B->qHead[MODE])));
                    return (heldEnd - heldNext) + (d > 0 ? (uint32_t)d : 0u);
                });
#ifdef PRT_PROFILE
            if (pass == 0) pTStep += __builtin_amdgcn_s_memtime() - pTc;
#endif
        }
    }
    if (overflow) lds_st(&B->overflow, 1u);
    if (COUNT) {
        block_count_traffic(B, tr);
        // the same events per traversal mode (bench.py splits the algorithmic bytes by them): words 16 + 3 * MODE of the counters
        unsigned long long b = tr.nBox, t = tr.nTri, p = tr.nTap;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            b += (unsigned long long)__shfl_xor((long long)b, o, 64);
            t += (unsigned long long)__shfl_xor((long long)t, o, 64);
            p += (unsigned long long)__shfl_xor((long long)p, o, 64);
        }
        if (lane == 0) {
            unsigned long long* C = A.counters + (size_t)(blockIdx.x % PRT_STAT_SHARDS) * PRT_STAT_STRIDE;
            if (b) atomicAdd(&C[16 + MODE * 3], b);
            if (t) atomicAdd(&C[17 + MODE * 3], t);
            if (p) atomicAdd(&C[18 + MODE * 3], p);
        }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef PRT_PROFILE
    unsigned long long pops = tr.pPops, deepPops = tr.pDeepPops; // per lane: summed over the wave by all its lanes
    unsigned long long directInt = tr.pDirectInt, popInt = tr.pPopInt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        directInt += (unsigned long long)__shfl_xor((long long)directInt, o, 64);
        popInt += (unsigned long long)__shfl_xor((long long)popInt, o, 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        pops += (unsigned long long)__shfl_xor((long long)pops, o, 64);
        deepPops += (unsigned long long)__shfl_xor((long long)deepPops, o, 64);
    }
    if (lane == 0) {
        unsigned long long* C = A.counters + (size_t)(blockIdx.x % PRT_STAT_SHARDS) * PRT_STAT_STRIDE;
        if (MODE != PRT_MODE_PACKET) { // (the packet traversal pops in a loop of its own and is not counted; word 39 is its slot)
            atomicAdd(&C[39 + MODE * 8], pops);
            atomicAdd(&C[39], deepPops);
        }
        atomicAdd(&C[32 + MODE * 8], tr.pNodeRounds);
        atomicAdd(&C[33 + MODE * 8], tr.pNodeLanes);
        atomicAdd(&C[34 + MODE * 8], tr.pLeafRounds);
        atomicAdd(&C[35 + MODE * 8], tr.pLeafLanes);
        atomicAdd(&C[36 + MODE * 8], tr.pTri2Lanes);
        atomicAdd(&C[112], pShortTurns);
        atomicAdd(&C[113], pShortLanes);
        atomicAdd(&C[114], pShortOthers);
        atomicAdd(&C[115], pShortReady);
        atomicAdd(&C[116], pShortLock);
        atomicAdd(&C[96 + MODE * 4], pTRefill >> 10);
        atomicAdd(&C[97 + MODE * 4], pTEnter >> 10);
        atomicAdd(&C[98 + MODE * 4], pTStep >> 10);
        atomicAdd(&C[64 + MODE * 8], tr.pNodeWaitLeaf);
        atomicAdd(&C[65 + MODE * 8], tr.pNodeDone);
        atomicAdd(&C[66 + MODE * 8], tr.pNodeNoRay);
        atomicAdd(&C[67 + MODE * 8], tr.pLeafWaitNode);
        atomicAdd(&C[68 + MODE * 8], tr.pLeafDone);
        atomicAdd(&C[69 + MODE * 8], tr.pLeafNoRay);
        atomicAdd(&C[70 + MODE * 8], tr.pLeafUpdates);
        atomicAdd(&C[71 + MODE * 8], tr.pNodeDistinct);
        atomicAdd(&C[122], tr.pAlphaRounds);
        atomicAdd(&C[123], tr.pAlphaCycles >> 10);
        atomicAdd(&C[120], directInt);
        atomicAdd(&C[121], popInt);
        atomicAdd(&C[37 + MODE * 8], pRefillLanes);
        atomicAdd(&C[38 + MODE * 8], pRefills);
        atomicAdd(&C[16 + MODE * 3], pTurns);
        atomicAdd(&C[17 + MODE * 3], pLanes);
        atomicAdd(&C[18 + MODE * 3], (__builtin_amdgcn_s_memtime() - pT0) >> 10);
        atomicAdd(&C[28], pClaims);
        atomicAdd(&C[29], pEmptyClaims);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------- roles
// Shade role (the caller holds the lock): give empty rows new work, then sweep the pending words and run the ready groups.
template <bool COUNT, bool ENV>
__device__ PRT_SHADE_INLINE bool shade_role(uint64_t kargs, uint32_t half)
{
    const FrameArgs& A = frame_args(kargs);
    const BlockLds B = block_lds();
    const uint32_t poolBase = blockIdx.x * PRT_POOL_GROUPS;
    const uint32_t lane = threadIdx.x & 63u;
    half = PRT_SHADERS > 1 ? bcast0(half) : 0u; // this role's rows: half, half + PRT_SHADERS, ... (interleaved: a launch too small to fill
                                                 // the pools gives every block its FIRST rows, and both roles must get some of them)
    lds_w* const readyList = (lds_w*)&B->readyList[half * PRT_SHADER_GROUPS];
    bool did = false;
    __builtin_amdgcn_s_setprio(PRT_SHADE_PRIO);
    // ---- rows whose groups are all done take the next 64 work items
    for (uint32_t row = half; row < A.rowsPerBlock; row += PRT_SHADERS) {
        if (bcast0(lds_ld(&B->chunkLive[row])) != 0u || bcast0(lds_ld(&B->exhausted)) != 0u) continue;
        // Rows come from 8 cursors, one per eighth of the work (a band of the image): a block takes rows from the band of
        // its own XCD first (blocks b and b + 8 share an XCD and its L2), so that the rays an XCD has in flight come from one
        // part of the image and its L2 keeps that part of the tree; a block whose band is used up helps with the others.
        uint32_t c = PRT_NONE;
        if (lane == 0) {
            const uint32_t per = (A.totalChunks + 7u) / 8u;
            for (uint32_t k = 0; k < 8u && c == PRT_NONE; k++) {
                const uint32_t band = (blockIdx.x + k) & 7u, first = band * per;
                const uint32_t count = first >= A.totalChunks ? 0u : (A.totalChunks - first < per ? A.totalChunks - first : per);
                if (gld(&A.ctrl[PRT_CTRL_CURSORS + 32u * band]) >= count) continue; // used up (the counter only grows)
                const uint32_t got = atomicAdd(&A.ctrl[PRT_CTRL_CURSORS + 32u * band], 1u);
                if (got < count) c = first + got;
            }
        }
        c = bcast0(c);
        if (c == PRT_NONE) {
            if (lane == 0) lds_st(&B->exhausted, 1u);
            break;
        }
        // Work items of row c: 64 consecutive ones (a quarter of a 16x16 tile) -- or, in a launch too small to keep every
        // block's pool full, 8 clusters of 8 consecutive items taken at equal distances over the whole launch, so that every
        // row (and with it every block) gets the same mix of cheap and expensive pixels.
        const uint32_t w = A.spreadRows ? ((c + (lane >> 3) * A.totalChunks) * 8u + (lane & 7u)) : c * PRT_CHUNK + lane;
        const bool valid = w < A.totalWork;
        const uint32_t P = poolBase + row * PRT_CHUNK + lane;
        if (valid) {
            const uint32_t pixel = work_item_pixel(A, w);
            gst(&A.gPixel[P], pixel);
            gst(&A.gRng[P], pixel != 0xffffffffu ? pixel_seed(pixel & 0xffffu, pixel >> 16, A.cam.width, A.p.seed) : 0u);
            gst(&A.gInfo[P], (uint32_t)PH_START << 20);
            gst4(&A.gColor[P], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        }
        lds_st(&B->pending[row * PRT_CHUNK + lane], valid ? 0u : PEND_DONE);
        const uint32_t n = (uint32_t)__popcll(__ballot(valid));
        if (lane == 0) {
            lds_st(&B->chunkLive[row], n);
            lds_add(&B->live, n);
            lds_add(&B->ready[half], n);
        }
        did = true;
        break; // one row per call: the blocks of a small launch all start at once, so its rows spread evenly over them
    }
    wg_release(); // the group headers above are read back below (other lanes of this wave) and by later shade rounds
    // ---- sweep: collect the ready groups of all rows (ballot + popcount rank into a list in LDS), then run them 8 per pass
    uint32_t n = 0;
    for (uint32_t row = half; row < PRT_POOL_CHUNKS; row += PRT_SHADERS) {
        if (bcast0(lds_ld(&B->chunkLive[row])) == 0u) continue;
        const uint32_t v = lds_ld_acq(&B->pending[row * PRT_CHUNK + lane]);
        const unsigned long long m = __ballot(v == 0u);
        if (v == 0u) lds_st(&readyList[n + mbcnt64(m)], row * PRT_CHUNK + lane);
        n += (uint32_t)__popcll(m);
    }
    for (uint32_t i = 0; i < n; i += 8u) {
        const uint32_t j = i + (lane >> 3);
        const bool has = j < n;
        const uint32_t local = has ? lds_ld(&readyList[j]) : 0u;
        const uint32_t np = shade_pass<COUNT, ENV>(kargs, has ? poolBase + local : PRT_NONE, local);
        // (shade_pass has published the groups' state, their pending words and the queue tails, in that order)
        const bool head = has && (lane & 7u) == 0u;
        const bool done = head && np == PEND_DONE;
        if (done) lds_sub(&B->chunkLive[local >> 6], 1u);
        const uint32_t taken = n - i < 8u ? n - i : 8u;
        const uint32_t nDone = (uint32_t)__popcll(__ballot(done)), nAgain = (uint32_t)__popcll(__ballot(head && np == 0u));
        if (lane == 0) {
            if (nDone) lds_sub(&B->live, nDone);
            lds_sub(&B->ready[half], taken - nAgain);
        }
#ifdef PRT_PROFILE
        if (lane == 0) {
            unsigned long long* C = A.counters + (size_t)(blockIdx.x % PRT_STAT_SHARDS) * PRT_STAT_STRIDE;
            atomicAdd(&C[30], 1ull);
            atomicAdd(&C[31], (unsigned long long)taken);
        }
#endif
        did = true;
    }
    __builtin_amdgcn_s_setprio(0);
    return did;
}

template <bool COUNT, bool ENV>
__global__ __launch_bounds__(PRT_BLOCK, PRT_FRAME_WAVES) void frame_kernel(const FrameArgs kernargs)
{
    const uint64_t kargs = frame_kernarg_bits();
    const FrameArgs& A = frame_args(kargs);
    const BlockLds B = block_lds();
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < PRT_POOL_GROUPS; i += PRT_BLOCK) B->pending[i] = PEND_DONE;
    for (uint32_t i = tid; i < (PRT_HOT_LDS ? PRT_HOT_NODES * 4u : 0u); i += PRT_BLOCK) {
        const float4 v = gld4(A.sc.hotNodes + i);
        B->hot[4 * i] = v.x;
        B->hot[4 * i + 1] = v.y;
        B->hot[4 * i + 2] = v.z;
        B->hot[4 * i + 3] = v.w;
    }
    if (tid < PRT_POOL_CHUNKS) B->chunkLive[tid] = 0;
    if (tid < Q_COUNT) {
        B->qTail[tid] = 0;
        B->qWaves[tid] = 0;
        B->qRes[tid] = 0;
        B->qHead[tid] = 0;
    }
    if (tid == 0) {
        for (int h = 0; h < PRT_SHADERS; h++) B->lock[h] = B->ready[h] = 0;
        B->live = 0;
        B->exhausted = 0;
        B->abort = 0;
        B->exited = 0;
        B->overflow = 0;
        B->rays = B->occl = B->px = 0;
        B->nBox = B->nTri = B->nHit = B->nTap = 0;
    }
    __syncthreads();
    uint32_t idle = 0;
    unsigned long long idleSince = 0; // s_memrealtime at the first of the current run of idle turns
#ifdef PRT_PROFILE
    unsigned long long tShade = 0, tTrace = 0, tIdle = 0, nShade = 0, nTrace = 0, t0 = __builtin_amdgcn_s_memtime(), tStart = t0;
#define PROF(acc, cnt)                                            \
    do {                                                          \
        unsigned long long t1_ = __builtin_amdgcn_s_memtime();    \
        acc += t1_ - t0;                                          \
        cnt;                                                      \
        t0 = t1_;                                                 \
    } while (0)
#else
#define PROF(acc, cnt)
#endif
    for (;;) {
        bool did = false;
        PROF(tIdle, (void)0);
        // ---- decision point: this wave holds no rays
        uint32_t total = 0, best = 0, bestLen = 0;
#pragma unroll
        for (int q = 0; q < Q_COUNT; q++) {
            const int32_t d = (int32_t)(lds_ld_acq(&B->qTail[q]) - lds_ld(&B->qHead[q]));
            const uint32_t len = d > 0 ? (uint32_t)d : 0u;
            total += len;
#if PRT_BALANCE
            // rays queued per wave that would then be on the queue: waves spread over the queues in proportion to their lengths instead of
            // all taking the fullest one and running it dry together
            const uint32_t score = len == 0u ? 0u : 1u + (len * PRT_BALANCE) / (lds_ld(&B->qWaves[q]) * PRT_BALANCE_W + PRT_BALANCE);
#else
            const uint32_t score = len;
#endif
            if (score > bestLen) {
                bestLen = score;
                best = (uint32_t)q;
            }
        }
        const uint32_t ready = block_ready(B), live = lds_ld(&B->live), exhausted = lds_ld(&B->exhausted);
        const bool roomForRows = exhausted == 0u && live + PRT_CHUNK <= A.rowsPerBlock * PRT_CHUNK;
        if (ready >= PRT_SHADE_MIN || (ready > 0u && total < 64u) || (roomForRows && total < 64u) || (roomForRows && ready + live == 0u) ||
            (PRT_TRACE_MIN && ready >= PRT_TRACE_READY && bestLen < PRT_TRACE_MIN)) {
            // which shade role: the one with the most ready groups -- or, with nothing ready, one that has a row to give new work to;
            // if it is taken, the next free one
            uint32_t got = 0, half = 0;
            if (lane == 0) {
                if (PRT_SHADERS > 1) {
                    uint32_t bestReady = 0, any = 0;
                    for (uint32_t h = 0; h < PRT_SHADERS; h++) {
                        const uint32_t r = lds_ld(&B->ready[h]);
                        any += r;
                        if (r > bestReady) {
                            bestReady = r;
                            half = h;
                        }
                    }
                    if (any == 0u)
                        for (uint32_t row = 0; row < A.rowsPerBlock; row++)
                            if (lds_ld(&B->chunkLive[row]) == 0u) {
                                half = row % PRT_SHADERS;
                                break;
                            }
                }
                for (uint32_t k = 0; k < PRT_SHADERS && !got; k++) {
                    const uint32_t h = (half + k) % PRT_SHADERS;
                    if (lds_cas(&B->lock[h], 0u, 1u)) {
                        got = 1u;
                        half = h;
                    }
                }
            }
            half = bcast0(half);
            if (bcast0(got)) {
                wg_acquire();
                did = shade_role<COUNT, ENV>(kargs, half);
                if (lane == 0) lds_st_rel(&B->lock[half], 0u);
                PROF(tShade, nShade++);
            }
        }
        if (!did && total > 0u) {
            best = bcast0(best);
            if (PRT_BALANCE && lane == 0) lds_add(&B->qWaves[best], 1u);
            if (best == Q_PRIMARY) trace_queue<PRT_MODE_PACKET, COUNT>(kargs);
            else if (best == Q_SCATTER) trace_queue<PRT_MODE_SINGLE, COUNT>(kargs);
            else if (best == Q_OCC_PACKET) trace_queue<PRT_MODE_OCC_PACKET, COUNT>(kargs);
            else trace_queue<PRT_MODE_OCC_SINGLE, COUNT>(kargs);
            if (PRT_BALANCE && lane == 0) lds_sub(&B->qWaves[best], 1u);
            did = true;
            PROF(tTrace, nTrace++);
        }
        if (did) {
            idle = 0;
            continue;
        }
        if (lds_ld(&B->exhausted) != 0u && lds_ld(&B->live) == 0u) break; // nothing left and nothing can arrive
        if (lds_ld(&B->abort) != 0u) break;
        __builtin_amdgcn_s_sleep(8);
        if (idle++ == 0u) idleSince = __builtin_amdgcn_s_memrealtime();
        // watchdog: a scheduling bug must end as an error code, never as a hung GPU (the clock is read every 256th idle turn)
        if ((idle & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - idleSince > PRT_WATCHDOG_TICKS) {
            // what the block looked like, for the host's error message (first 8 waves that give up)
            uint32_t stuck = 0, sum = 0;
            for (uint32_t i = lane; i < PRT_POOL_GROUPS; i += 64u) {
                const uint32_t v = lds_ld(&B->pending[i]);
                if (v != PEND_DONE && v != 0u) {
                    stuck++;
                    sum += v;
                }
            }
            stuck = wave_sum(stuck);
            sum = wave_sum(sum);
            if (lane == 0) {
                lds_st(&B->abort, 1u);
                uint32_t* S = A.ctrl + PRT_WORK_WORDS; // the sticky words: no render clears them (prt_sticky_error)
                atomicOr(&S[0], 1u);
                const uint32_t k = atomicAdd(&S[2], 1u);
                if (k < 8u) {
                    uint32_t* D = S + 8 + 16 * k;
                    D[0] = blockIdx.x; D[1] = tid >> 6; D[2] = block_ready(B); D[3] = lds_ld(&B->live); D[4] = lds_ld(&B->exhausted);
                    D[5] = block_lock_free(B) ? 0u : 1u; D[6] = stuck; D[7] = sum;
                    for (int q = 0; q < Q_COUNT; q++) {
                        D[8 + q] = lds_ld(&B->qTail[q]);
                        D[12 + q] = lds_ld(&B->qHead[q]);
                    }
                }
            }
            break;
        }
    }
    // ---- statistics: the last wave to leave adds the block's sums to one of PRT_STAT_SHARDS copies of the counters
    unsigned long long* C = A.counters + (size_t)(blockIdx.x % PRT_STAT_SHARDS) * PRT_STAT_STRIDE;
#ifdef PRT_PROFILE
    PROF(tIdle, (void)0);
    if (lane == 0) {
        atomicAdd(&C[8], tShade);
        atomicAdd(&C[9], tTrace);
        atomicAdd(&C[10], tIdle);
        atomicAdd(&C[11], nShade);
        atomicAdd(&C[12], nTrace);
        atomicAdd(&C[13], t0 - tStart);
        atomicAdd(&C[14], 1ull);
    }
#endif
    if (lane == 0 && lds_add(&B->exited, 1u) == PRT_BLOCK / 64u - 1u) {
        wg_acquire();
        const uint32_t r = lds_ld(&B->rays), o = lds_ld(&B->occl), px = lds_ld(&B->px);
        if (r) atomicAdd(&C[0], (unsigned long long)r);
        if (o) atomicAdd(&C[1], (unsigned long long)o);
        if (px) atomicAdd(&C[6], (unsigned long long)px);
        if (COUNT) {
            if (B->nBox) atomicAdd(&C[2], (unsigned long long)B->nBox);
            if (B->nTri) atomicAdd(&C[3], (unsigned long long)B->nTri);
            if (B->nHit) atomicAdd(&C[4], (unsigned long long)B->nHit);
            if (B->nTap) atomicAdd(&C[5], (unsigned long long)B->nTap);
        }
        if (lds_ld(&B->overflow)) {
            atomicAdd(&C[7], 1ull);
            atomicOr(&A.ctrl[PRT_WORK_WORDS], 2u); // sticky: survives the next render's clearing of the counters
        }
    }
}
