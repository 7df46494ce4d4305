// tools/gather_bench.hip -- how many uncoalesced 16-byte gathers per cycle can a CU sustain?  (diagnostic, not product)
// Each lane reads `per` records of 64 B (4 x dwordx4, like a BVH node) at pseudo-random indices of a table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void gather(const float4* tab, uint32_t mask, uint32_t iters, int loadsPerRec, float* out)
{
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const float4* p = tab + 4 * (size_t)(x & mask);
        float4 a = p[0];
        acc += a.x;
        if (loadsPerRec > 1) { float4 b = p[1]; acc += b.y; }
        if (loadsPerRec > 2) { float4 c = p[2]; acc += c.z; }
        if (loadsPerRec > 3) { float4 d = p[3]; acc += d.w; }
        x += __float_as_uint(acc) & 1u; // dependent chain like a traversal
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    for (uint32_t logRecs : {14u, 16u, 18u, 20u}) { // 1 MB, 4 MB, 16 MB, 64 MB tables
        uint32_t recs = 1u << logRecs;
        float4* tab; float* out;
        hipMalloc(&tab, (size_t)recs * 64); hipMemset(tab, 0, (size_t)recs * 64);
        int blocks = 256 * 7;
        hipMalloc(&out, blocks * 256 * 4);
        for (int lp : {1, 4}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            uint32_t iters = 2000;
            gather<<<blocks, 256>>>(tab, recs - 1, 100, lp, out);
            hipEventRecord(e0);
            gather<<<blocks, 256>>>(tab, recs - 1, iters, lp, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double req = (double)blocks * 256 * iters * lp;
            printf("table %5u KB  loads/rec %d: %.2f G lane-requests/s = %.3f per CU per ns (%.2f ms)\n", recs * 64 / 1024, lp, req / ms / 1e6,
                   req / ms / 1e6 / 256.0, ms);
        }
        hipFree(tab); hipFree(out);
    }
    return 0;
}
