// Micro benchmark: what HBM rate a kernel of the IDCT/colour kernel's access shape can reach on this GPU, without
// any arithmetic: (1) a linear copy, 16 bytes a lane; (2) the same with four pieces in flight per lane; (3) reads as
// the kernel does them (12 KiB of coefficients per workgroup step, contiguous) and writes as it does them (16 pixel
// rows of 768 bytes at a pitch of 5760 bytes); (4) read only; (5) write only.  Sizes: 12.6 GB in, 12.4 GB out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy1(const uint4 *src, uint4 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy4(const uint4 *src, uint4 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x;
    uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
}
__global__ __launch_bounds__(256) void read_only(const uint4 *src, uint32_t *sink, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x;
    uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    const uint32_t s = a.x ^ b.y ^ c.z ^ d.w;
    if (s == 0x12345678u) sink[0] = s;
}
__global__ __launch_bounds__(256) void write_only(uint4 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x;
    const uint4 v = make_uint4((uint32_t)i, 1, 2, 3);
    dst[i] = v; dst[i + 256] = v; dst[i + 512] = v; dst[i + 768] = v;
}
// One 1920x1080 4:2:0 image: 8160 MCUs x 6 blocks x 128 B of coefficients, 1080 rows x 5760 B of pixels.  A workgroup
// takes `ns` strips of 16 MCUs (12 KiB in, 16 rows x 768 B out), one after the other, the next strip's loads issued
// before this strip's stores.
// `wmode`: how the 16 rows x 768 bytes of a strip are spread over the four wavefronts' store instructions:
//   0  as the kernel does: a wavefront owns a 192-byte column of all 16 rows (row fragments of one and a half lines)
//   1  a wavefront owns four whole 768-byte rows (six full 128-byte lines each)
//   2  no pixel rows at all: the output is written as contiguous 12 KiB per strip (what a linear layout would cost)
template <int WMODE>
__global__ __launch_bounds__(256) void k2_shape(const uint4 *coef, uint8_t *rgb, uint32_t ns, uint32_t strided)
{
    const uint32_t img = blockIdx.y, grp = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t strips_x = 8, nstrips = 8 * 68, ngrp = (nstrips + ns - 1) / ns;
    const uint4 *cimg = coef + (size_t)img * (8160u * 6u * 8u);
    uint8_t *pimg = rgb + (size_t)img * 6220800u;
    uint32_t strip = strided ? grp : grp * ns;
    const uint32_t step = strided ? ngrp : 1u, end = strided ? nstrips : min(nstrips, grp * ns + ns);
    auto load = [&](uint32_t s, uint4 (&v)[3]) {
        const uint32_t my = s / strips_x, sx = s % strips_x;
        const uint32_t x0 = sx * 16u + wave * 4u;
        const uint32_t nm = x0 < 120u ? min(4u, 120u - x0) : 0u;
        const uint4 *p = cimg + ((size_t)my * 120u + x0) * 48u;
        for (int i = 0; i < 3; i++) {
            const uint32_t q = lane + i * 64u;
            v[i] = make_uint4(0, 0, 0, 0);
            if ((q >> 3) < nm * 6u) v[i] = p[q];
        }
    };
    uint4 v[3], w[3];
    if (strip < end) load(strip, v);
    for (; strip < end; strip += step) {
        const uint32_t my = strip / strips_x, sx = strip % strips_x;
        const uint32_t x0 = sx * 16u + wave * 4u;
        for (int i = 0; i < 3; i++) w[i] = v[i];
        if (strip + step < end) load(strip + step, v);
        const uint32_t row_valid = x0 < 120u ? min(192u, (1920u - x0 * 16u) * 3u) : 0u;
        const uint32_t rows_valid = min(16u, 1080u - min(1080u, my * 16u));
        for (int i = 0; i < 3; i++) {
            if (WMODE == 0) {
                const uint32_t q = lane + i * 64u, r = q / 12u, o = (q % 12u) * 16u;
                if (r < rows_valid && o < row_valid) *reinterpret_cast<uint4 *>(pimg + (size_t)(my * 16u + r) * 5760u + x0 * 48u + o) = w[i];
            } else if (WMODE == 1) {
                const uint32_t q = lane + i * 64u, r = wave * 4u + q / 48u, o = (q % 48u) * 16u; // 48 pieces per 768-byte row
                const uint32_t xs = sx * 768u;
                if (r < rows_valid && xs + o < 5760u) *reinterpret_cast<uint4 *>(pimg + (size_t)(my * 16u + r) * 5760u + xs + o) = w[i];
            } else {
                const uint32_t q = wave * 192u + lane + i * 64u;
                *reinterpret_cast<uint4 *>(pimg + (size_t)strip * 12288u + q * 16u) = w[i];
            }
        }
    }
}

int main()
{
    const uint32_t nimg = 2000;
    const size_t cbytes = (size_t)nimg * 8160u * 6u * 128u, pbytes = (size_t)nimg * 6220800u;
    void *a, *b;
    CHECK(hipMalloc(&a, cbytes + 65536));
    CHECK(hipMalloc(&b, cbytes + 65536));
    CHECK(hipMemset(a, 1, cbytes));
    CHECK(hipMemset(b, 2, cbytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t n16 = cbytes / 16 / 1024 * 1024;
    auto timeit = [&](const char *name, double bytes, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("%-72s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
        return 0;
    };
    timeit("linear copy, 16 B per lane", 2.0 * n16 * 16, [&] { hipLaunchKernelGGL(copy1, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, n16); });
    timeit("linear copy, 4 x 16 B per lane", 2.0 * n16 * 16, [&] { hipLaunchKernelGGL(copy4, dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, n16); });
    timeit("read only, 4 x 16 B per lane", 1.0 * n16 * 16, [&] { hipLaunchKernelGGL(read_only, dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const uint4 *)a, (uint32_t *)b, n16); });
    timeit("write only, 4 x 16 B per lane", 1.0 * n16 * 16, [&] { hipLaunchKernelGGL(write_only, dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (uint4 *)b, n16); });
    for (uint32_t strided = 0; strided < 2; strided++)
        for (uint32_t ns : {1u, 4u, 16u}) {
            char name[128];
            snprintf(name, sizeof name, "kernel's shape, %2u strips/workgroup%s, 192 B columns", ns, strided ? ", interleaved" : "");
            timeit(name, (double)cbytes + (double)pbytes, [&] { hipLaunchKernelGGL(k2_shape<0>, dim3((8 * 68 + ns - 1) / ns, nimg), dim3(256), 0, 0, (const uint4 *)a, (uint8_t *)b, ns, strided); });
            snprintf(name, sizeof name, "kernel's shape, %2u strips/workgroup%s, whole 768 B rows", ns, strided ? ", interleaved" : "");
            timeit(name, (double)cbytes + (double)pbytes, [&] { hipLaunchKernelGGL(k2_shape<1>, dim3((8 * 68 + ns - 1) / ns, nimg), dim3(256), 0, 0, (const uint4 *)a, (uint8_t *)b, ns, strided); });
            snprintf(name, sizeof name, "kernel's shape, %2u strips/workgroup%s, linear output", ns, strided ? ", interleaved" : "");
            timeit(name, (double)cbytes + (double)pbytes, [&] { hipLaunchKernelGGL(k2_shape<2>, dim3((8 * 68 + ns - 1) / ns, nimg), dim3(256), 0, 0, (const uint4 *)a, (uint8_t *)b, ns, strided); });
        }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return 0;
}
