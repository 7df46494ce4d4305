// prt_bvh.cpp -- host-side BVH construction producing exactly the tree the reference builds
// (binned SAH over 32 buckets and 3 axes, bvh.cpp:31-171; DFS linearisation, bvh.cpp:230-243), so
// that node order, leaf order and therefore the kernels' tie-breaking agree with the reference.
// The tree is built as a pointer structure (subtrees at depth 3 are built concurrently; the index
// ranges are disjoint, so the topology does not depend on scheduling) and then flattened.
#include <float.h>
#include <string.h>

#include <future>
#include <thread>

#include "prt.h"

namespace prt
{

BBox BBox::init()
{
    BBox b;
    b.lower = std::numeric_limits<float>::max();
    b.upper = std::numeric_limits<float>::lowest();
    return b;
}
void BBox::merge(const BBox& o)
{
    lower = Vector3f(std::fmin(lower.x, o.lower.x), std::fmin(lower.y, o.lower.y), std::fmin(lower.z, o.lower.z));
    upper = Vector3f(std::fmax(upper.x, o.upper.x), std::fmax(upper.y, o.upper.y), std::fmax(upper.z, o.upper.z));
}
void BBox::merge(const Vector3f& p)
{
    lower = Vector3f(std::fmin(lower.x, p.x), std::fmin(lower.y, p.y), std::fmin(lower.z, p.z));
    upper = Vector3f(std::fmax(upper.x, p.x), std::fmax(upper.y, p.y), std::fmax(upper.z, p.z));
}
float BBox::surfaceArea() const
{
    auto e = upper - lower;
    return 2.0f * (e.x * e.y + e.y * e.z + e.z * e.x);
}

namespace
{

struct BuildNode {
    BBox bbox = BBox::init();
    std::unique_ptr<BuildNode> child[2];
    int32_t primIndex = -1; // leaf: first entry of the remap range
    int32_t primCount = 0;
    uint32_t splitAxis = 0;
};

struct Builder {
    const uint32_t* indices;
    const Vector3f* positions;
    uint32_t* remap;
    int forkLevel; // build() forks its two halves while level < forkLevel

    Vector3f centroid(uint32_t prim) const
    {
        Vector3f temp(0.0f);
        for (uint32_t j = 0; j < 3; j++) temp = temp + positions[indices[3 * prim + j]];
        return 1.0f / 3.0f * temp; // bvh.cpp:85,132: 1.0f/kVertexCountPerPrim*temp
    }

    // cvttss2si semantics for the bucket index (bvh.cpp:87): NaN / out of range -> INT_MIN
    static int32_t toInt(float f)
    {
        if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
        return (int32_t)f;
    }

    void build(BuildNode* node, int32_t start, int32_t end, int level) const
    {
        for (int32_t i = start; i <= end; i++)
            for (uint32_t j = 0; j < 3; j++) node->bbox.merge(positions[indices[3 * remap[i] + j]]);
        const int32_t primCount = end - start + 1;
        if (primCount <= 8) { // kMaxPrimCountInNode = kLaneCount (bvh.h:21)
            node->primIndex = start;
            node->primCount = primCount;
            return;
        }
        const BBox& bbox = node->bbox;
        const Vector3f extent = bbox.upper - bbox.lower;
        const uint32_t kBucketCount = 32;
        float lowestCost = std::numeric_limits<float>::max();
        uint32_t lowestDim = 0;
        int32_t lowestCostSplit = -1;
        for (uint32_t dim = 0; dim < 3; dim++) {
            const float splitExtent = extent.v[dim] == 0.0f ? 0.0001f : extent.v[dim];
            const float lowerPos = bbox.lower.v[dim];
            uint32_t count[32];
            BBox bounds[32];
            for (uint32_t k = 0; k < kBucketCount; k++) { count[k] = 0; bounds[k] = BBox::init(); }
            for (int32_t i = start; i <= end; i++) {
                Vector3f temp(0.0f);
                BBox primBounds = BBox::init();
                for (uint32_t j = 0; j < 3; j++) {
                    const Vector3f& v = positions[indices[3 * remap[i] + j]];
                    temp = temp + v;
                    primBounds.merge(v);
                }
                temp = 1.0f / 3.0f * temp;
                int32_t b = toInt((float)kBucketCount * (temp.v[dim] - lowerPos) / splitExtent);
                if ((uint32_t)b >= kBucketCount) b = (int32_t)kBucketCount - 1; // bvh.cpp:88 compares as unsigned
                count[b]++;
                bounds[b].merge(primBounds);
            }
            for (uint32_t i = 0; i < kBucketCount - 1; i++) {
                uint32_t countLeft = 0, countRight = 0;
                BBox left = BBox::init(), right = BBox::init();
                for (uint32_t j = 0; j <= i; j++) { countLeft += count[j]; left.merge(bounds[j]); }
                for (uint32_t j = i + 1; j < kBucketCount; j++) { countRight += count[j]; right.merge(bounds[j]); }
                float cost = 0.125f + ((float)countLeft * left.surfaceArea() + (float)countRight * right.surfaceArea());
                if (lowestCost > cost) {
                    lowestDim = dim;
                    lowestCost = cost;
                    lowestCostSplit = (int32_t)i;
                }
            }
        }
        const uint32_t dim = lowestDim;
        const float splitExtent = extent.v[dim] == 0.0f ? 0.0001f : extent.v[dim];
        const float lowerPos = bbox.lower.v[dim];
        const float splitPos = lowerPos + (float)(lowestCostSplit + 1) * splitExtent / (float)kBucketCount;
        auto splitLeft = [&](int32_t i) { return centroid(remap[i]).v[dim] < splitPos; };
        int32_t cursor;
        for (cursor = start; cursor <= end; cursor++)
            if (!splitLeft(cursor)) break;
        for (int32_t i = cursor + 1; i <= end; i++) {
            if (splitLeft(i)) {
                std::swap(remap[i], remap[cursor]);
                cursor++;
            }
        }
        int32_t mid = cursor - 1;
        if (mid <= start || end <= mid) mid = (start + end) / 2;
        node->splitAxis = dim;
        node->child[0].reset(new BuildNode);
        node->child[1].reset(new BuildNode);
        if (level < forkLevel && primCount > 4096) {
            auto f = std::async(std::launch::async, [&]() { build(node->child[0].get(), start, mid, level + 1); });
            build(node->child[1].get(), mid + 1, end, level + 1);
            f.get();
        } else {
            build(node->child[0].get(), start, mid, level + 1);
            build(node->child[1].get(), mid + 1, end, level + 1);
        }
    }
};

void flatten(const BuildNode* node, std::vector<prt_bvh_node>& out, uint32_t& leafCount)
{
    const size_t self = out.size();
    out.emplace_back();
    {
        prt_bvh_node& l = out[self];
        memcpy(l.lower, &node->bbox.lower, 12);
        memcpy(l.upper, &node->bbox.upper, 12);
        l.splitAxis = node->splitAxis;
        l.triVectorIndex = 0;
    }
    if (node->primIndex < 0) {
        out[self].primCount = 0xf; // LinearBvhNode::kInternalNode
        flatten(node->child[0].get(), out, leafCount);
        out[self].primOrSecondNodeIndex = (uint32_t)out.size();
        flatten(node->child[1].get(), out, leafCount);
    } else {
        out[self].primOrSecondNodeIndex = (uint32_t)node->primIndex;
        out[self].primCount = (uint32_t)node->primCount;
        out[self].triVectorIndex = leafCount++;
    }
}

} // namespace

void buildBvhArrays(uint32_t primCount, const uint32_t* indices, const Vector3f* positions, std::vector<prt_bvh_node>& nodes,
                    std::vector<uint32_t>& remap, int threads)
{
    remap.resize(primCount);
    for (uint32_t i = 0; i < primCount; i++) remap[i] = i;
    nodes.clear();
    if (primCount == 0) return;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    // levels 0..2 fork (up to 8 concurrent subtrees) when more than one thread is available
    Builder b{indices, positions, remap.data(), threads > 1 ? 3 : 0};
    BuildNode root;
    b.build(&root, 0, (int32_t)primCount - 1, 0);
    uint32_t leafCount = 0;
    nodes.reserve(primCount / 2 + 16);
    flatten(&root, nodes, leafCount);
}

void Bvh::build(Mesh&& mesh)
{
    m_mesh = std::move(mesh);
    buildBvhArrays(m_mesh.getPrimCount(), m_mesh.m_indices.data(), m_mesh.m_positions.data(), m_nodes, m_primRemapping);
    logPrintf(LogLevel::kVerbose, "LinearBvhNode (count=%u)\n", (unsigned)m_nodes.size());
}

} // namespace prt
