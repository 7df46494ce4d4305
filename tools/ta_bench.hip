// tools/ta_bench.hip -- what does the texture-address path (TA/TCP) charge for a 64-byte record gather, by access shape?
// (diagnostic, not product).  Every variant fetches ONE 64-byte record per lane and step at a pseudo-random index of a
// table, in a dependent chain like a BVH walk; what differs is which lane issues which 16-byte piece:
//   A  "own":  lane l loads the 4 pieces of ITS record with 4 dwordx4 loads (each instruction: 64 lanes, 64 different lines)
//   B  "quad": 4 rounds; in round r lanes 4i..4i+3 load the 4 pieces of the record of lane 16r+i (each instruction: 16 lines,
//              every quad of lanes inside one 64-byte segment); the pieces are handed to the owner by ds_bpermute
//   C  "quad-lds": as B, but the loads are LDS-DMA (global_load_lds_dwordx4) and the owner reads its record back from LDS
// Output: records per ns per CU for each shape and table size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xs(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

__global__ __launch_bounds__(256) void own_kernel(const float4* tab, uint32_t mask, uint32_t iters, float* out)
{
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        x = xs(x);
        const float4* p = tab + 4 * (size_t)(x & mask);
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        x += __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// B: the record index of lane j travels to the four loader lanes by ds_bpermute, the loaded pieces travel back the same way
__global__ __launch_bounds__(256) void quad_kernel(const float4* tab, uint32_t mask, uint32_t iters, float* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        x = xs(x);
        const uint32_t rec = x & mask;
        float4 mine[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t owner = 16u * r + (lane >> 2);
            const uint32_t orec = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner * 4u), (int)rec);
            mine[r] = tab[4 * (size_t)orec + (lane & 3u)];
        }
        // owner j = 16 r + i gets piece c from lane 4 i + c of round r: 4 pieces x 4 dwords
        const uint32_t r = lane >> 4, srcBase = 4u * (lane & 15u);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                // every lane takes part in every permute; the owner keeps the round that is its own
                float vx = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)((srcBase + c) * 4u), (int)__float_as_uint(c == 0 ? mine[rr].x : c == 1 ? mine[rr].y : c == 2 ? mine[rr].z : mine[rr].w)));
                if ((uint32_t)rr == r) s += vx;
            }
        }
        acc += s;
        x += __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// C: LDS-DMA.  Round r: lane L = 4 i + c loads piece c of the record of lane 16 r + i; the DMA writes it at
// stage[wave][r][L] (M0 base + 16 * lane), i.e. record-contiguous; the owner reads 4 x b128.
__global__ __launch_bounds__(256) void quadlds_kernel(const float4* tab, uint32_t mask, uint32_t iters, float* out)
{
    __shared__ float4 stage[4][4][64]; // [wave][round][lane] = 16 KB per block
    __shared__ uint32_t recs[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        x = xs(x);
        const uint32_t rec = x & mask;
        recs[wave][lane] = rec;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t orec = recs[wave][16u * r + (lane >> 2)];
            const float4* src = tab + 4 * (size_t)orec + (lane & 3u);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)&stage[wave][r][0], 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0)
        const float4* mine = &stage[wave][lane >> 4][4u * (lane & 15u)];
        float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
        acc += a.x + b.y + c.z + d.w;
        x += __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// D: the quad shape with NO hand-back (pure address-path cost of 16 lines per instruction against 64)
__global__ __launch_bounds__(256) void quadraw_kernel(const float4* tab, uint32_t mask, uint32_t iters, float* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 256 + (threadIdx.x & ~3u)) * 2654435761u + 12345u; // one chain per quad
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            x = xs(x);
            float4 v = tab[4 * (size_t)(x & mask) + (lane & 3u)];
            s += v.x;
        }
        acc += s;
        x += __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int blocks = 256 * 7;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    for (uint32_t logRecs : {8u, 14u, 18u, 20u}) { // 16 KB (L1), 1 MB (L2), 16 MB, 64 MB
        uint32_t recs = 1u << logRecs;
        float4* tab;
        hipMalloc(&tab, (size_t)recs * 64);
        hipMemset(tab, 0, (size_t)recs * 64);
        for (int v = 0; v < 4; v++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const uint32_t iters = 1000;
            auto launch = [&](uint32_t it) {
                if (v == 0) own_kernel<<<blocks, 256>>>(tab, recs - 1, it, out);
                else if (v == 1) quad_kernel<<<blocks, 256>>>(tab, recs - 1, it, out);
                else if (v == 2) quadlds_kernel<<<blocks, 256>>>(tab, recs - 1, it, out);
                else quadraw_kernel<<<blocks, 256>>>(tab, recs - 1, it, out);
            };
            launch(50);
            hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double records = (double)blocks * 256 * iters;
            const char* names[4] = {"own (4 x dwordx4 per lane)", "quad + ds_bpermute hand-back", "quad LDS-DMA + ds_read_b128", "quad, no hand-back"};
            printf("table %6u KB  %-30s %8.3f ms  %.3f records/ns/CU  (%s)\n", recs * 64 / 1024, names[v], ms, records / ms / 1e6 / 256.0,
                   hipGetErrorString(hipGetLastError()));
        }
        hipFree(tab);
    }
    return 0;
}
