// tools/ta_lanes.hip -- does a 16-byte-per-lane gather cost the vector-memory path per INSTRUCTION or per ACTIVE LANE?
// (diagnostic, not product).  Dependent chains as in ta_bench.hip; only every `stride`-th lane is active, and an active lane
// loads `pieces` 16-byte pieces of its 64-byte record.  Output: ns per step of the whole chip and lane-loads per ns per CU.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint32_t xs(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int PIECES>
__global__ __launch_bounds__(256) void k(const float4* tab, uint32_t mask, uint32_t iters, uint32_t stride, float* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    if (lane % stride == 0u) {
        for (uint32_t i = 0; i < iters; i++) {
            x = xs(x);
            const float4* p = tab + 4 * (size_t)(x & mask);
            float4 a = p[0];
            acc += a.x;
            if (PIECES > 1) { float4 b = p[1]; acc += b.y; }
            if (PIECES > 2) { float4 c = p[2]; acc += c.z; }
            if (PIECES > 3) { float4 d = p[3]; acc += d.w; }
            x += __float_as_uint(acc) & 1u;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int blocks = 256 * 8;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    for (uint32_t logRecs : {8u, 14u, 18u}) { // 16 KB (L1), 1 MB (L2), 16 MB
        uint32_t recs = 1u << logRecs;
        float4* tab;
        hipMalloc(&tab, (size_t)recs * 64);
        hipMemset(tab, 0, (size_t)recs * 64);
        for (int pieces : {4, 2, 1})
            for (uint32_t stride : {1u, 2u, 4u, 8u}) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                const uint32_t iters = 2000;
                auto launch = [&](uint32_t it) {
                    if (pieces == 4) k<4><<<blocks, 256>>>(tab, recs - 1, it, stride, out);
                    else if (pieces == 2) k<2><<<blocks, 256>>>(tab, recs - 1, it, stride, out);
                    else k<1><<<blocks, 256>>>(tab, recs - 1, it, stride, out);
                };
                launch(50);
                hipEventRecord(e0);
                launch(iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double laneLoads = (double)blocks * 256 / stride * pieces * iters;
                const double waveInstr = (double)blocks * 4 * pieces * iters;
                printf("table %6u KB  %d pieces  %2u of 64 lanes: %7.3f ms  %6.2f lane-loads/ns/CU  %6.3f wave-loads/ns/CU  (%s)\n", recs * 64 / 1024, pieces,
                       64 / stride, ms, laneLoads / (ms * 1e6) / 256, waveInstr / (ms * 1e6) / 256, hipGetErrorString(hipGetLastError()));
            }
        hipFree(tab);
    }
    return 0;
}
