// prt_bvh_build.hip -- Bvh::build on the GPU: the reference's binned-SAH tree (bvh.cpp:31-171: 32 buckets x 3
// axes, leaves of <= 8 triangles) and its depth-first linearisation (bvh.cpp:230-299), producing the IDENTICAL node array,
// leaf order and primRemapping -- the traversal's tie-breaking depends on them.
//
// The reference recurses node by node; here a whole LEVEL of the tree is processed at a time (every step below is one launch
// over all triangles or all nodes of the level), which gives the same tree because nothing a node computes depends on
// another node of its level:
//   bounds     a node's box = min / max over its triangles' vertices (order-free; float atomics on order-preserving keys)
//   bins       per node, axis and bucket: triangle count and bounds.  The bucket of a triangle is the reference's expression
//              (32 * (centroid - lower) / extent, truncated, compared as unsigned), evaluated in its order without contraction
//   split      93 candidates per node, costs with the reference's operations; the first strict minimum in (axis, bucket)
//              order wins, NaN costs (an empty side: 0 * inf) never do -- bvh.cpp:98-125
//   partition  the reference's in-place loop (bvh.cpp:137-150) is a stable move of the "left" triangles to the front; its
//              effect on the right ones has a closed form (see scatter_kernel), so one segmented prefix sum of the left flags
//              (hipCUB) and one scatter reproduce the exact permutation
//   children   [start, mid] and [mid + 1, end], with the reference's fallback to the middle when a side would be empty
// and the linearisation is two passes over the levels (subtree sizes bottom-up, depth-first indices top-down).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

#include "prt_internal.h"

namespace {

constexpr uint32_t kBuckets = 32;          // bvh.cpp:64
constexpr uint32_t kLeafMax = 8;           // kMaxPrimCountInNode = kLaneCount, bvh.h:21
constexpr uint32_t kBinWords = 7;          // count + lower xyz + upper xyz
constexpr uint32_t kNone = 0xffffffffu;

// order-preserving key of a non-NaN float: unsigned compare of keys == float compare
__device__ __forceinline__ uint32_t fkey(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
#define KEY_MAX_FLT 0xff7fffffu // fkey(FLT_MAX): BBox::init().lower
#define KEY_LOWEST_FLT 0x00800000u // fkey(-FLT_MAX): BBox::init().upper

struct BuildArrays {
    uint32_t n;            // triangles
    const uint32_t* idx;   // 3 per triangle
    const float* pos;      // 3 per vertex
    uint32_t* remap;       // current permutation (leaf order in the making)
    uint32_t* remapNext;
    uint32_t* posNode;     // node that owns position i (deepest so far)
    uint32_t* flag;        // 1 = the triangle at position i goes left
    uint32_t* rank;        // inclusive count of left flags inside the node
    // per node (capacity 2n)
    uint32_t* nStart;
    uint32_t* nEnd;
    uint32_t* nChild;      // first child (second = +1), kNone for a leaf
    uint32_t* nAxis;
    uint32_t* nBox;        // 6 keys: lower xyz, upper xyz
    uint32_t* nSlot;       // index into the level's active list, kNone when the node is a leaf
    uint32_t* nSize;       // nodes in the subtree
    uint32_t* nLeaves;     // leaves in the subtree
    uint32_t* nIndex;      // depth-first index
    uint32_t* nLeafIndex;  // leaves before the subtree in depth-first order
    // per active node of the level
    uint32_t* active;      // node ids
    uint32_t* bins;        // [slot][3][32][7]
    uint32_t* firstRight;  // first position of the node whose triangle does not go left
    float* splitPos;
    uint32_t* counters;    // [0] nodes allocated, [1] active nodes of this level, [2] nodes created at this level's end
};

__device__ __forceinline__ float3 vtx(const BuildArrays& B, uint32_t prim, uint32_t j)
{
    const float* p = B.pos + 3 * (size_t)B.idx[3 * (size_t)prim + j];
    return make_float3(p[0], p[1], p[2]);
}

// centroid component as the reference forms it (bvh.cpp:78-85, 129-133): ((0 + v0) + v1) + v2, then (1/3) * sum
__device__ __forceinline__ float centroid(const BuildArrays& B, uint32_t prim, uint32_t dim)
{
    float t = 0.0f;
    for (uint32_t j = 0; j < 3; j++) {
        const float* p = B.pos + 3 * (size_t)B.idx[3 * (size_t)prim + j];
        t = t + p[dim];
    }
    return (1.0f / 3.0f) * t;
}

// std::fmin / std::fmax ignore a NaN operand (vecmath.cpp:46-79 merges with them): a NaN coordinate never reaches a bound
__device__ __forceinline__ void key_min(uint32_t* k, float v) { if (v == v) atomicMin(k, fkey(v)); }
__device__ __forceinline__ void key_max(uint32_t* k, float v) { if (v == v) atomicMax(k, fkey(v)); }

// ---- bounds of the nodes created at the previous step: node->bbox.merge(every vertex), bvh.cpp:45-49.  Near the root every
// wave's 64 triangles belong to one node: the wave reduces first and issues 6 atomics instead of 6 * 64 on the same words.
__global__ void bounds_kernel(BuildArrays B, uint32_t firstNew)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < B.n;
    const uint32_t node = in ? B.posNode[i] : kNone;
    const bool work = in && node >= firstNew;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (work) {
        const uint32_t prim = B.remap[i];
        for (uint32_t j = 0; j < 3; j++) {
            const float3 v = vtx(B, prim, j);
            lo[0] = fminf(lo[0], v.x); lo[1] = fminf(lo[1], v.y); lo[2] = fminf(lo[2], v.z);
            hi[0] = fmaxf(hi[0], v.x); hi[1] = fmaxf(hi[1], v.y); hi[2] = fmaxf(hi[2], v.z);
        }
    }
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
    const bool uniform = __all(node == first); // the whole wave (idle lanes included: their node is kNone) in one node
    if (uniform) {
        if (first == kNone || first < firstNew) return;
        for (int o = 32; o > 0; o >>= 1)
            for (int d = 0; d < 3; d++) {
                lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64));
                hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64));
            }
        if ((threadIdx.x & 63u) == 0u) {
            uint32_t* box = B.nBox + 6 * (size_t)first;
            for (int d = 0; d < 3; d++) {
                key_min(box + d, lo[d]);
                key_max(box + 3 + d, hi[d]);
            }
        }
    } else if (work) {
        uint32_t* box = B.nBox + 6 * (size_t)node;
        for (int d = 0; d < 3; d++) {
            key_min(box + d, lo[d]);
            key_max(box + 3 + d, hi[d]);
        }
    }
}

// ---- nodes with more than 8 triangles join the level's active list (bvh.cpp:52-57)
__global__ void classify_kernel(BuildArrays B, uint32_t firstNew, uint32_t nodeCount)
{
    const uint32_t node = firstNew + blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= nodeCount) return;
    const uint32_t count = B.nEnd[node] - B.nStart[node] + 1;
    if (count <= kLeafMax) {
        B.nSlot[node] = kNone;
        return;
    }
    const uint32_t slot = atomicAdd(&B.counters[1], 1u);
    B.nSlot[node] = slot;
    B.active[slot] = node;
    B.firstRight[slot] = kNone;
}

__global__ void clear_bins_kernel(BuildArrays B, uint32_t activeCount)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)activeCount * 3 * kBuckets) return;
    uint32_t* b = B.bins + i * kBinWords;
    b[0] = 0;
    b[1] = b[2] = b[3] = KEY_MAX_FLT;
    b[4] = b[5] = b[6] = KEY_LOWEST_FLT;
}

// cvttss2si (bvh.cpp:87: int32_t b = ...): NaN and out-of-range values give INT_MIN
__device__ __forceinline__ int32_t to_int(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return (int32_t)0x80000000u;
    return (int32_t)f;
}

// ---- bins (bvh.cpp:66-96): for each axis the triangle's bucket takes its count and bounds.  A block whose 256 triangles
// belong to one node (every block near the root) collects them in LDS and adds only its non-empty bins to the node's.
__global__ __launch_bounds__(256) void bin_kernel(BuildArrays B)
{
    __shared__ uint32_t sBins[3 * kBuckets * kBinWords];
    __shared__ uint32_t sNode[2]; // min and max node id of the block
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < B.n;
    const uint32_t node = in ? B.posNode[i] : kNone;
    if (threadIdx.x == 0) {
        sNode[0] = kNone;
        sNode[1] = 0;
    }
    for (uint32_t k = threadIdx.x; k < 3 * kBuckets; k += blockDim.x) {
        uint32_t* b = sBins + k * kBinWords;
        b[0] = 0;
        b[1] = b[2] = b[3] = KEY_MAX_FLT;
        b[4] = b[5] = b[6] = KEY_LOWEST_FLT;
    }
    __syncthreads();
    if (in) {
        atomicMin(&sNode[0], node);
        atomicMax(&sNode[1], node);
    }
    __syncthreads();
    const bool uniform = sNode[0] == sNode[1]; // (a partial last block is uniform when its real triangles are)
    const uint32_t slot = in ? B.nSlot[node] : kNone;
    if (slot != kNone) {
        const uint32_t prim = B.remap[i];
        const uint32_t* box = B.nBox + 6 * (size_t)node;
        float lo[3], hi[3], sum[3] = {0.0f, 0.0f, 0.0f};
        lo[0] = lo[1] = lo[2] = FLT_MAX;
        hi[0] = hi[1] = hi[2] = -FLT_MAX;
        for (uint32_t j = 0; j < 3; j++) {
            const float3 v = vtx(B, prim, j);
            const float c[3] = {v.x, v.y, v.z};
            for (int d = 0; d < 3; d++) {
                sum[d] = sum[d] + c[d];
                lo[d] = fminf(lo[d], c[d]);
                hi[d] = fmaxf(hi[d], c[d]);
            }
        }
        for (uint32_t dim = 0; dim < 3; dim++) {
            const float lower = funkey(box[dim]), upper = funkey(box[3 + dim]);
            const float extent = upper - lower;
            const float splitExtent = extent == 0.0f ? 0.0001f : extent; // bvh.cpp:67
            const float c = (1.0f / 3.0f) * sum[dim];
            int32_t b = to_int((float)kBuckets * (c - lower) / splitExtent);
            if ((uint32_t)b >= kBuckets) b = (int32_t)kBuckets - 1; // :88, an unsigned compare
            uint32_t* bin = uniform ? sBins + (dim * kBuckets + (uint32_t)b) * kBinWords
                                    : B.bins + (((size_t)slot * 3 + dim) * kBuckets + (uint32_t)b) * kBinWords;
            atomicAdd(bin, 1u);
            for (int d = 0; d < 3; d++) {
                key_min(bin + 1 + d, lo[d]);
                key_max(bin + 4 + d, hi[d]);
            }
        }
    }
    __syncthreads();
    if (uniform && sNode[0] != kNone) {
        const uint32_t bslot = B.nSlot[sNode[0]];
        if (bslot != kNone)
            for (uint32_t k = threadIdx.x; k < 3 * kBuckets; k += blockDim.x) {
                const uint32_t* sb = sBins + k * kBinWords;
                if (sb[0] == 0u) continue;
                uint32_t* gb = B.bins + ((size_t)bslot * 3 * kBuckets + k) * kBinWords;
                atomicAdd(gb, sb[0]);
                for (int d = 0; d < 3; d++) {
                    atomicMin(gb + 1 + d, sb[1 + d]);
                    atomicMax(gb + 4 + d, sb[4 + d]);
                }
            }
    }
}

struct HBox {
    float lx, ly, lz, ux, uy, uz;
};
__device__ __forceinline__ HBox hbox_init() { return HBox{FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX}; }
__device__ __forceinline__ void hbox_merge(HBox& a, const uint32_t* bin)
{
    a.lx = fminf(a.lx, funkey(bin[1])); a.ly = fminf(a.ly, funkey(bin[2])); a.lz = fminf(a.lz, funkey(bin[3]));
    a.ux = fmaxf(a.ux, funkey(bin[4])); a.uy = fmaxf(a.uy, funkey(bin[5])); a.uz = fmaxf(a.uz, funkey(bin[6]));
}
__device__ __forceinline__ float hbox_area(const HBox& b) // vecmath.cpp: 2 * (ex*ey + ey*ez + ez*ex)
{
    const float ex = b.ux - b.lx, ey = b.uy - b.ly, ez = b.uz - b.lz;
    return 2.0f * (ex * ey + ey * ez + ez * ex);
}

// ---- split (bvh.cpp:98-135): one wave per active node; candidate k = axis * 31 + bucket; the first strict minimum wins
__global__ __launch_bounds__(64) void split_kernel(BuildArrays B, uint32_t activeCount)
{
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= activeCount) return;
    const uint32_t node = B.active[slot];
    float best = FLT_MAX; // lowestCost
    uint32_t bestK = kNone;
    for (uint32_t k = lane; k < 3 * (kBuckets - 1); k += 64) {
        const uint32_t dim = k / (kBuckets - 1), i = k % (kBuckets - 1);
        const uint32_t* bins = B.bins + ((size_t)slot * 3 + dim) * kBuckets * kBinWords;
        uint32_t cl = 0, cr = 0;
        HBox left = hbox_init(), right = hbox_init();
        for (uint32_t j = 0; j <= i; j++) {
            cl += bins[j * kBinWords];
            hbox_merge(left, bins + j * kBinWords);
        }
        for (uint32_t j = i + 1; j < kBuckets; j++) {
            cr += bins[j * kBinWords];
            hbox_merge(right, bins + j * kBinWords);
        }
        const float cost = 0.125f + ((float)cl * hbox_area(left) + (float)cr * hbox_area(right));
        if (best > cost) { // lanes see their candidates in increasing k
            best = cost;
            bestK = k;
        }
    }
    // the winner over the wave: smallest cost, ties to the smallest candidate index (the reference's scan order)
    for (int o = 32; o > 0; o >>= 1) {
        const float oc = __shfl_xor(best, o, 64);
        const uint32_t ok = (uint32_t)__shfl_xor((int)bestK, o, 64);
        const bool take = ok != kNone && (bestK == kNone || oc < best || (oc == best && ok < bestK));
        if (take) {
            best = oc;
            bestK = ok;
        }
    }
    if (lane == 0) {
        const uint32_t dim = bestK == kNone ? 0u : bestK / (kBuckets - 1); // lowestDim = 0, lowestCostSplit = -1 when nothing won
        const int32_t split = bestK == kNone ? -1 : (int32_t)(bestK % (kBuckets - 1));
        const uint32_t* box = B.nBox + 6 * (size_t)node;
        const float lower = funkey(box[dim]), upper = funkey(box[3 + dim]);
        const float extent = upper - lower;
        const float splitExtent = extent == 0.0f ? 0.0001f : extent;
        B.splitPos[slot] = lower + (float)(split + 1) * splitExtent / (float)kBuckets; // :128
        B.nAxis[node] = dim;
    }
}

// ---- left flags (bvh.cpp:129-135) and the first triangle of each node that stays right
__global__ void flag_kernel(BuildArrays B)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    const uint32_t node = B.posNode[i];
    const uint32_t slot = B.nSlot[node];
    uint32_t f = 0;
    if (slot != kNone) f = centroid(B, B.remap[i], B.nAxis[node]) < B.splitPos[slot] ? 1u : 0u;
    B.flag[i] = f;
    // the first position of the node that stays right: one atomic per wave and node run instead of one per triangle
    const bool right = slot != kNone && !f;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prevNode = (uint32_t)__shfl_up((int)node, 1, 64);
    const bool prevRight = __shfl_up((int)right, 1, 64) != 0;
    if (right && (lane == 0 || prevNode != node || !prevRight)) atomicMin(&B.firstRight[slot], i);
}

// ---- the reference's partition loop (bvh.cpp:137-150) in closed form.  With c0 = the first position whose triangle stays
// right, the loop moves the m-th later left triangle to c0 + m - 1 (a stable move) by swapping it with whatever stands
// there; what stands there is the original triangle of that position if it stays right, else the triangle that an
// earlier swap left there.  So a position beyond the final cursor that held a left triangle receives the triangle found by
// following p = c0 + m - 1 while position p itself held a left triangle (then p = c0 + m(p) - 1, which is smaller).
__global__ void scatter_kernel(BuildArrays B)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    const uint32_t node = B.posNode[i];
    const uint32_t slot = B.nSlot[node];
    if (slot == kNone) {
        B.remapNext[i] = B.remap[i];
        return;
    }
    const uint32_t start = B.nStart[node], end = B.nEnd[node];
    const uint32_t c0 = B.firstRight[slot] == kNone ? end + 1 : B.firstRight[slot];
    const uint32_t lead = c0 - start;                       // left triangles in front of c0: they stay where they are
    const uint32_t total = B.rank[end];                     // all left triangles of the node (inclusive scan inside the node)
    const uint32_t cursor = c0 + (total - lead);            // first position of the right part
    if (i < c0) {
        B.remapNext[i] = B.remap[i];
    } else if (B.flag[i]) {
        const uint32_t m = B.rank[i] - lead;
        B.remapNext[c0 + m - 1] = B.remap[i];
        if (i >= cursor) {
            uint32_t p = c0 + m - 1;
            while (B.flag[p]) p = c0 + (B.rank[p] - lead) - 1;
            B.remapNext[i] = B.remap[p];
        }
    } else if (i >= cursor) {
        B.remapNext[i] = B.remap[i];
    }
}

// ---- children (bvh.cpp:152-170)
__global__ void children_kernel(BuildArrays B, uint32_t activeCount)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= activeCount) return;
    const uint32_t node = B.active[slot];
    const int32_t start = (int32_t)B.nStart[node], end = (int32_t)B.nEnd[node];
    const uint32_t c0 = B.firstRight[slot] == kNone ? (uint32_t)end + 1 : B.firstRight[slot];
    const uint32_t total = B.rank[end], lead = c0 - (uint32_t)start;
    int32_t mid = (int32_t)(c0 + (total - lead)) - 1;
    if (mid <= start || end <= mid) mid = (start + end) / 2;
    const uint32_t child = atomicAdd(&B.counters[0], 2u);
    B.nChild[node] = child;
    B.nStart[child] = (uint32_t)start;
    B.nEnd[child] = (uint32_t)mid;
    B.nStart[child + 1] = (uint32_t)mid + 1;
    B.nEnd[child + 1] = (uint32_t)end;
    for (uint32_t c = child; c < child + 2; c++) {
        B.nChild[c] = kNone;
        B.nAxis[c] = 0;
        B.nSlot[c] = kNone;
        uint32_t* box = B.nBox + 6 * (size_t)c;
        box[0] = box[1] = box[2] = KEY_MAX_FLT;
        box[3] = box[4] = box[5] = KEY_LOWEST_FLT;
    }
}

__global__ void assign_kernel(BuildArrays B)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    const uint32_t node = B.posNode[i];
    if (B.nSlot[node] == kNone) return;
    const uint32_t child = B.nChild[node];
    B.posNode[i] = i <= B.nEnd[child] ? child : child + 1;
}

// ---- linearisation (bvh.cpp:230-243): sizes bottom-up over the node ids of one level, indices top-down
__global__ void size_kernel(BuildArrays B, uint32_t first, uint32_t last)
{
    const uint32_t node = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= last) return;
    const uint32_t c = B.nChild[node];
    if (c == kNone) {
        B.nSize[node] = 1;
        B.nLeaves[node] = 1;
    } else {
        B.nSize[node] = 1 + B.nSize[c] + B.nSize[c + 1];
        B.nLeaves[node] = B.nLeaves[c] + B.nLeaves[c + 1];
    }
}

__global__ void index_kernel(BuildArrays B, uint32_t first, uint32_t last)
{
    const uint32_t node = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= last) return;
    const uint32_t c = B.nChild[node];
    if (c == kNone) return;
    B.nIndex[c] = B.nIndex[node] + 1;
    B.nLeafIndex[c] = B.nLeafIndex[node];
    B.nIndex[c + 1] = B.nIndex[node] + 1 + B.nSize[c];
    B.nLeafIndex[c + 1] = B.nLeafIndex[node] + B.nLeaves[c];
}

__global__ void emit_kernel(BuildArrays B, uint32_t nodeCount, prt_bvh_node* out)
{
    const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= nodeCount) return;
    prt_bvh_node o;
    const uint32_t* box = B.nBox + 6 * (size_t)node;
    for (int d = 0; d < 3; d++) {
        o.lower[d] = funkey(box[d]);
        o.upper[d] = funkey(box[3 + d]);
    }
    const uint32_t c = B.nChild[node];
    if (c == kNone) {
        o.primOrSecondNodeIndex = B.nStart[node];
        o.primCount = B.nEnd[node] - B.nStart[node] + 1;
        o.triVectorIndex = B.nLeafIndex[node];
        o.splitAxis = 0;
    } else {
        o.primOrSecondNodeIndex = B.nIndex[c + 1];
        o.primCount = 0xf; // LinearBvhNode::kInternalNode
        o.triVectorIndex = 0;
        o.splitAxis = B.nAxis[node];
    }
    out[B.nIndex[node]] = o;
}

__global__ void iota_kernel(uint32_t n, uint32_t* a, uint32_t* posNode)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        a[i] = i;
        posNode[i] = 0;
    }
}

struct DevMem {
    std::vector<void*> ptrs;
    ~DevMem()
    {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t get(T** out, size_t count)
    {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 16) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T*)p;
        return e;
    }
};

} // namespace

extern "C" int prt_hip_build_bvh(prt_hip_ctx* c, uint32_t primCount, const uint32_t* indices, uint32_t vertexCount, const float* positions,
                                 prt_bvh_node* nodes_out, uint32_t* nodeCount_out, uint32_t* primRemapping_out, double* buildMs)
{
    if (!c || !indices || !positions || !nodes_out || !nodeCount_out || !primRemapping_out) return prt_fail(PRT_HIP_EINVAL, "NULL argument");
    if (primCount == 0 || primCount >= (1u << 30)) return prt_fail(PRT_HIP_EINVAL, "primCount must be 1 .. 2^30 - 1");
    for (size_t k = 0; k < (size_t)primCount * 3; k++)
        if (indices[k] >= vertexCount) return prt_fail(PRT_HIP_EINVAL, "vertex index out of range");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint32_t n = primCount, maxNodes = 2 * n;
    DevMem mem;
    BuildArrays B{};
    B.n = n;
    uint32_t* dIdx;
    float* dPos;
    HIP_TRY(mem.get(&dIdx, (size_t)n * 3));
    HIP_TRY(mem.get(&dPos, (size_t)vertexCount * 3));
    B.idx = dIdx;
    B.pos = dPos;
    HIP_TRY(mem.get(&B.remap, n)); HIP_TRY(mem.get(&B.remapNext, n)); HIP_TRY(mem.get(&B.posNode, n));
    HIP_TRY(mem.get(&B.flag, n)); HIP_TRY(mem.get(&B.rank, n));
    HIP_TRY(mem.get(&B.nStart, maxNodes)); HIP_TRY(mem.get(&B.nEnd, maxNodes)); HIP_TRY(mem.get(&B.nChild, maxNodes));
    HIP_TRY(mem.get(&B.nAxis, maxNodes)); HIP_TRY(mem.get(&B.nBox, (size_t)maxNodes * 6)); HIP_TRY(mem.get(&B.nSlot, maxNodes));
    HIP_TRY(mem.get(&B.nSize, maxNodes)); HIP_TRY(mem.get(&B.nLeaves, maxNodes)); HIP_TRY(mem.get(&B.nIndex, maxNodes));
    HIP_TRY(mem.get(&B.nLeafIndex, maxNodes));
    const uint32_t maxActive = n / (kLeafMax + 1) + 1; // an active node holds at least 9 triangles
    HIP_TRY(mem.get(&B.active, maxActive)); HIP_TRY(mem.get(&B.firstRight, maxActive)); HIP_TRY(mem.get(&B.splitPos, maxActive));
    HIP_TRY(mem.get(&B.bins, (size_t)maxActive * 3 * kBuckets * kBinWords));
    HIP_TRY(mem.get(&B.counters, 16));
    prt_bvh_node* dOut;
    HIP_TRY(mem.get(&dOut, maxNodes));
    size_t scanBytes = 0;
    HIP_TRY(hipcub::DeviceScan::InclusiveSumByKey(nullptr, scanBytes, B.posNode, B.flag, B.rank, (int)n, hipcub::Equality(), s));
    char* scanTemp;
    HIP_TRY(mem.get(&scanTemp, scanBytes));

    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipMemcpyAsync(dIdx, indices, (size_t)n * 12, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dPos, positions, (size_t)vertexCount * 12, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(e0, s));
    const dim3 blk(256), grdN((n + 255) / 256);
    hipLaunchKernelGGL(iota_kernel, grdN, blk, 0, s, n, B.remap, B.posNode);
    // node 0 = the root over [0, n - 1]
    {
        const uint32_t rootInit[4] = {0, n - 1, kNone, 0};
        HIP_TRY(hipMemcpyAsync(B.nStart, &rootInit[0], 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(B.nEnd, &rootInit[1], 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(B.nChild, &rootInit[2], 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(B.nAxis, &rootInit[3], 4, hipMemcpyHostToDevice, s));
        const uint32_t box[6] = {KEY_MAX_FLT, KEY_MAX_FLT, KEY_MAX_FLT, KEY_LOWEST_FLT, KEY_LOWEST_FLT, KEY_LOWEST_FLT};
        HIP_TRY(hipMemcpyAsync(B.nBox, box, sizeof(box), hipMemcpyHostToDevice, s));
        const uint32_t cnt[4] = {1, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(B.counters, cnt, sizeof(cnt), hipMemcpyHostToDevice, s));
    }
    std::vector<uint32_t> levelFirst; // first node id of each level; the last entry is the node count
    uint32_t firstNew = 0, nodeCount = 1;
    for (uint32_t level = 0; level < 4096; level++) {
        levelFirst.push_back(firstNew);
        const uint32_t created = nodeCount - firstNew;
        hipLaunchKernelGGL(bounds_kernel, grdN, blk, 0, s, B, firstNew);
        HIP_TRY(hipMemsetAsync(B.counters + 1, 0, 4, s));
        hipLaunchKernelGGL(classify_kernel, dim3((created + 255) / 256), blk, 0, s, B, firstNew, nodeCount);
        uint32_t activeCount = 0;
        HIP_TRY(hipMemcpyAsync(&activeCount, B.counters + 1, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (activeCount == 0) break;
        if (activeCount > maxActive) return prt_fail(PRT_HIP_ELAUNCH, "BVH build: more active nodes than triangles allow");
        hipLaunchKernelGGL(clear_bins_kernel, dim3((uint32_t)(((size_t)activeCount * 3 * kBuckets + 255) / 256)), blk, 0, s, B, activeCount);
        hipLaunchKernelGGL(bin_kernel, grdN, blk, 0, s, B);
        hipLaunchKernelGGL(split_kernel, dim3(activeCount), dim3(64), 0, s, B, activeCount);
        hipLaunchKernelGGL(flag_kernel, grdN, blk, 0, s, B);
        HIP_TRY(hipcub::DeviceScan::InclusiveSumByKey(scanTemp, scanBytes, B.posNode, B.flag, B.rank, (int)n, hipcub::Equality(), s));
        hipLaunchKernelGGL(scatter_kernel, grdN, blk, 0, s, B);
        hipLaunchKernelGGL(children_kernel, dim3((activeCount + 255) / 256), blk, 0, s, B, activeCount);
        hipLaunchKernelGGL(assign_kernel, grdN, blk, 0, s, B);
        std::swap(B.remap, B.remapNext);
        firstNew = nodeCount;
        nodeCount += 2 * activeCount;
        if (nodeCount > maxNodes) return prt_fail(PRT_HIP_ELAUNCH, "BVH build: node count exceeds 2 * primCount");
    }
    levelFirst.push_back(nodeCount);
    // node ids are allocated level by level, children of a level in any order: sizes bottom-up, indices top-down
    for (size_t l = levelFirst.size() - 1; l-- > 0;) {
        const uint32_t first = levelFirst[l], last = levelFirst[l + 1];
        if (last > first) hipLaunchKernelGGL(size_kernel, dim3((last - first + 255) / 256), blk, 0, s, B, first, last);
    }
    HIP_TRY(hipMemsetAsync(B.nIndex, 0, 4, s));
    HIP_TRY(hipMemsetAsync(B.nLeafIndex, 0, 4, s));
    for (size_t l = 0; l + 1 < levelFirst.size(); l++) {
        const uint32_t first = levelFirst[l], last = levelFirst[l + 1];
        if (last > first) hipLaunchKernelGGL(index_kernel, dim3((last - first + 255) / 256), blk, 0, s, B, first, last);
    }
    hipLaunchKernelGGL(emit_kernel, dim3((nodeCount + 255) / 256), blk, 0, s, B, nodeCount, dOut);
    HIP_TRY(hipEventRecord(e1, s));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return prt_fail(PRT_HIP_ELAUNCH, std::string("BVH build kernels: ") + hipGetErrorString(le));
    HIP_TRY(hipMemcpyAsync(nodes_out, dOut, (size_t)nodeCount * sizeof(prt_bvh_node), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(primRemapping_out, B.remap, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (buildMs) *buildMs = ms;
    *nodeCount_out = nodeCount;
    return PRT_HIP_OK;
}
