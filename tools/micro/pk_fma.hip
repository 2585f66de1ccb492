// Micro-benchmark: issue rate of v_fma_f32 against v_pk_fma_f32 on one wavefront per SIMD and on four.
// Build: hipcc -O3 --offload-arch=gfx950 -o pk_fma pk_fma.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, int iters, long long *cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 1.0001f, d = 0.5f;
    const f2 c2 = {c, c}, d2 = {d, d};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d);
                a4 = __builtin_fmaf(a4, c, d); a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                p0 = __builtin_elementwise_fma(p0, c2, d2); p1 = __builtin_elementwise_fma(p1, c2, d2);
                p2 = __builtin_elementwise_fma(p2, c2, d2); p3 = __builtin_elementwise_fma(p3, c2, d2);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float *out; long long *cyc, h;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int mode = 0; mode < 2; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            dim3 grid(256 * 4), block(64 * waves); // 4 workgroups per CU -> `waves` wavefronts per SIMD
            if (mode == 0) k<0><<<grid, block>>>(out, 10, cyc); else k<1><<<grid, block>>>(out, 10, cyc);
            hipEventRecord(e0);
            if (mode == 0) k<0><<<grid, block>>>(out, iters, cyc); else k<1><<<grid, block>>>(out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double fma_per_lane = (double)iters * 64;
            printf("waves/SIMD %d %s: %.3f ms, %.1f TFLOP/s (fp32), clock ticks/iter %.1f\n", waves, mode ? "v_pk_fma_f32" : "v_fma_f32   ", ms,
                   2.0 * fma_per_lane * 64 * waves * 1024 / (ms * 1e-3) / 1e12, (double)h / iters);
        }
    }
    return 0;
}
