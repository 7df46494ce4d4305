// tools/rec_gather.hip -- what does a gather of RECORDS cost beyond the L2 as a function of the record's size?
// (diagnostic, not product).  Dependent chains at the frame kernel's occupancy (8 waves per SIMD): every lane walks a random
// chain through a table of aligned records and reads BYTES of each record as 16-byte pieces.  Tables: inside the Infinity
// Cache (16 MB), around it (256 MB) and far beyond it (2 GB).  Output: records per ns of the whole chip and GB/s of the bytes
// the lanes asked for.  The question behind it: a two-level node record is 128 bytes instead of 64 -- does it gather at the
// rate of a 64-byte record (the rate of L2 MISSES is the roof of the large scenes) or at half of it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint32_t xs(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int BYTES, int USED>
__global__ __launch_bounds__(256) void k(const float4* tab, uint32_t mask, uint32_t iters, float* out)
{
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        x = xs(x);
        const float4* p = tab + (size_t)(BYTES / 16) * (size_t)(x & mask);
#pragma unroll
        for (int j = 0; j < USED / 16; j++) {
            float4 a = p[j];
            acc += a.x + a.w;
        }
        x += __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int BYTES, int USED>
static void run(const float4* tab, size_t tableBytes, float* out)
{
    const int blocks = 256 * 8;
    const uint32_t recs = (uint32_t)(tableBytes / BYTES);
    const uint32_t iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<BYTES, USED><<<blocks, 256>>>(tab, recs - 1, 50, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<BYTES, USED><<<blocks, 256>>>(tab, recs - 1, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("table %5zu MB  record %3d B  read %3d B: %8.3f ms  %6.2f records/ns  %7.1f GB/s asked for  (%s)\n", tableBytes >> 20, BYTES, USED, ms,
           n / (ms * 1e6), n * USED / (ms * 1e6), hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv)
{
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    if (argc > 1) { // rec_gather MB MB ... : 64-byte records only, tables of the given sizes (powers of two): where does the Infinity Cache stop helping?
        for (int a = 1; a < argc; a++) {
            const size_t bytes = (size_t)atoi(argv[a]) << 20;
            float4* tab;
            if (hipMalloc(&tab, bytes) != hipSuccess) { printf("no %s MB\n", argv[a]); continue; }
            hipMemset(tab, 0, bytes);
            run<64, 64>(tab, bytes, out);
            hipFree(tab);
        }
        return 0;
    }
    for (size_t mb : {16u, 256u, 2048u}) {
        const size_t bytes = mb << 20;
        float4* tab;
        if (hipMalloc(&tab, bytes) != hipSuccess) { printf("no %zu MB\n", mb); continue; }
        hipMemset(tab, 0, bytes);
        run<64, 64>(tab, bytes, out);
        run<64, 16>(tab, bytes, out);
        run<128, 128>(tab, bytes, out);
        run<128, 112>(tab, bytes, out);
        run<128, 64>(tab, bytes, out);
        run<256, 256>(tab, bytes, out);
        run<32, 32>(tab, bytes, out);
        hipFree(tab);
    }
    return 0;
}
