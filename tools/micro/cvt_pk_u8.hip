// Micro test: semantics of v_cvt_pk_u8_f32 on gfx950 (rounding and saturation), needed to decide whether
// the colour stage may use it for the reference's truncating, saturating `as u8` cast.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xAABBCCDDu);
}
int main()
{
    float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.9f, 255.f, 255.4f, 255.6f, 256.f, 300.f, 1e9f,
                 -0.4f, -0.6f, -1.f, -300.f, NAN, INFINITY, -INFINITY, 127.99f, 128.5f, 129.5f};
    int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o, ho[64];
    hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%12g -> byte1 %3u (word %08x)\n", h[i], (ho[i] >> 8) & 255, ho[i]);
    return 0;
}
