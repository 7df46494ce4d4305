// tools/fetch_calib.hip -- calibrate rocprofv3 FETCH_SIZE for this path's access patterns (diagnostic, not product):
//   gather: every lane reads one 64-byte record (4 x dwordx4) at a pseudo-random index of a 2 GiB table (past L2 and Infinity Cache)
//   stream: every lane reads 16 bytes, coalesced, once over 2 GiB
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void gather64(const float4* tab, uint32_t mask, uint32_t iters, float* out)
{
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const float4* p = tab + 4 * (size_t)(x & mask);
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void stream16(const float4* tab, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) acc += tab[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    const uint32_t recs = 1u << 25; // 2 GiB
    float4* tab; float* out;
    hipMalloc(&tab, (size_t)recs * 64); hipMemset(tab, 0, (size_t)recs * 64);
    const int blocks = 2048; const uint32_t iters = 64;
    hipMalloc(&out, blocks * 256 * 4);
    gather64<<<blocks, 256>>>(tab, recs - 1, iters, out);
    stream16<<<blocks, 256>>>(tab, (size_t)recs * 4, out);
    hipDeviceSynchronize();
    printf("gather64: %llu bytes requested (64 B x %llu records)\n", 64ull * blocks * 256 * iters, (unsigned long long)blocks * 256 * iters);
    printf("stream16: %llu bytes requested\n", (unsigned long long)recs * 64);
    return 0;
}
