// Micro test: does v_cvt_pk_u8_f32 follow the wavefront's f32 rounding mode?  With MODE.FP_ROUND (bits 1:0)
// set to 3 (toward zero) a truncating conversion would need no v_trunc_f32 in front of it.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    float v = i < n ? in[i] : 0.f;
    unsigned r;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %1, 1, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=v"(r) : "v"(v), "v"(0xAABBCCDDu));
    if (i < n) out[i] = r;
}
int main()
{
    float h[] = {0.f, 0.4f, 0.5f, 0.6f, 0.99f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.9f, 255.f, 255.4f, 255.6f, 256.f, 300.f, 1e9f,
                 -0.4f, -0.6f, -1.f, -300.f, NAN, INFINITY, -INFINITY, 127.99f, 128.5f, 129.5f};
    int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o, ho[64];
    hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    int same = 1;
    for (int i = 0; i < n; i++) {
        float t = truncf(h[i]);
        unsigned want = isnan(h[i]) ? 0u : (t < 0.f ? 0u : (t > 255.f ? 255u : (unsigned)t));
        unsigned got = (ho[i] >> 8) & 255;
        printf("%12g -> byte1 %3u want %3u %s\n", h[i], got, want, got == want ? "" : "DIFF");
        same &= got == want;
    }
    printf(same ? "RTZ honoured: truncation without v_trunc\n" : "RTZ not honoured\n");
    return 0;
}
