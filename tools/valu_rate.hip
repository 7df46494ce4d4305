// tools/valu_rate.hip -- how many cycles does a wave64 vector instruction occupy a SIMD for?  (diagnostic, not product)
// Every wave runs a long chain of independent v_fma_f32 (8 accumulators); 8 waves per SIMD; variants: all 64 lanes, lanes 0..31 only,
// every other lane, 16 lanes; and a chain of v_pk_fma_f32.  Output: wave-instructions per cycle per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    const unsigned lane = threadIdx.x & 63u;
    bool on = true;
    if (MODE == 1) on = lane < 32u;
    if (MODE == 2) on = (lane & 1u) == 0u;
    if (MODE == 3) on = lane < 16u;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 0.5f;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(256) void kpk(float* out, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    const f2 m = {1.0000001f, 1.0000002f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                         : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y;
}

int main()
{
    const int blocks = 256 * 8, iters = 2000;
    float* out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    int clockKHz = 0;
    (void)hipDeviceGetAttribute(&clockKHz, hipDeviceAttributeClockRate, 0);
    const char* names[5] = {"64 lanes", "lanes 0..31", "every other lane", "lanes 0..15", "v_pk_fma_f32, 64 lanes"};
    for (int v = 0; v < 5; v++) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        auto launch = [&](int it) {
            if (v == 0) k<0><<<blocks, 256>>>(out, it);
            else if (v == 1) k<1><<<blocks, 256>>>(out, it);
            else if (v == 2) k<2><<<blocks, 256>>>(out, it);
            else if (v == 3) k<3><<<blocks, 256>>>(out, it);
            else kpk<<<blocks, 256>>>(out, it);
        };
        launch(10);
        (void)hipEventRecord(e0);
        launch(iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double waveInstr = (double)blocks * 4 * iters * 128;                 // per wave: iters x 128 vector instructions
        const double perSimdPerNs = waveInstr / (256.0 * 4) / (ms * 1e6);
        printf("%-26s %8.3f ms   %.3f wave-instructions per ns per SIMD  = one per %.2f cycles at %.2f GHz (%s)\n", names[v], ms, perSimdPerNs,
               (clockKHz * 1e-6) / perSimdPerNs, clockKHz * 1e-6, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
