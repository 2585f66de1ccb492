// Micro benchmark: what rocprofv3's FETCH_SIZE tallies, per byte a kernel really asks memory for, in the read patterns of
// the entropy kernels.  Every kernel below reads a 1 GiB buffer (four times the Infinity Cache) exactly once -- or, where
// it says so, a known part of it -- so the counter's value per launch can be set against a known byte count; the
// program prints each kernel's time, from which the rate at which the bytes really moved can be told as well.
//   wide      16 bytes a lane, neighbouring lanes neighbouring pieces (the pattern the counter is known to halve)
//   lane64    the sync pass: a lane walks its own stretch of 16 KiB in steps of 64 bytes (4 x 16), the lanes of a
//             wavefront 16 KiB apart, sixteen wavefronts per CU
//   half64    the same, but every lane takes only the first 64 bytes of each 128: half the buffer is asked for
//   win64     the position pass: a lane takes 64 bytes (4 x 16), neighbouring lanes neighbouring 64 bytes
//   head56    the write pass: a lane takes 56 bytes (3 x 16 + 8) from a 4-byte aligned address, neighbouring lanes 24
//             bytes apart
//   pos8      the DC pass: a lane takes 8 bytes, neighbouring lanes 24 bytes apart
// Run: ./fetch_calib            (times)
//      rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./fetch_calib   (counter per kernel)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

static constexpr size_t N = 1ull << 30;       // bytes
static constexpr uint32_t STRETCH = 16384;     // bytes a lane walks in lane64 / half64

__global__ __launch_bounds__(256) void calib_wide(const uint4 *src, uint32_t *sink)
{
    const size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x;
    const uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    const uint32_t s = a.x ^ b.y ^ c.z ^ d.w;
    if (s == 0x12345678u) sink[0] = s;
}

template <uint32_t STEP>
__global__ __launch_bounds__(256) void calib_lane64(const uint8_t *src, uint32_t *sink)
{
    __shared__ uint32_t pad[39 * 256]; // 39 KB: four workgroups per CU, as the sync pass has
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint8_t *p = src + ((size_t)(blockIdx.x * 4u + wave) * 64u + lane) * STRETCH;
    uint32_t s = 0;
    for (uint32_t o = 0; o < STRETCH; o += STEP) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p + o);
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        s ^= a.x ^ b.y ^ c.z ^ d.w;
        pad[threadIdx.x] = s; // keeps the steps apart a little and the array alive
    }
    if (s == 0x12345678u) sink[0] = s + pad[(threadIdx.x + 1) & 255];
}

__global__ __launch_bounds__(256) void calib_win64(const uint4 *src, uint32_t *sink)
{
    const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
    const uint4 a = src[i], b = src[i + 1], c = src[i + 2], d = src[i + 3];
    const uint32_t s = a.x ^ b.y ^ c.z ^ d.w;
    if (s == 0x12345678u) sink[0] = s;
}

__global__ __launch_bounds__(256) void calib_head56(const uint32_t *src, uint32_t *sink)
{
    const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 6u; // 24 bytes a lane
    const uint32_t *q = src + i;
    const u32x4_a4 a = *reinterpret_cast<const u32x4_a4 *>(q), b = *reinterpret_cast<const u32x4_a4 *>(q + 4),
                   c = *reinterpret_cast<const u32x4_a4 *>(q + 8);
    const u32x2_a4 d = *reinterpret_cast<const u32x2_a4 *>(q + 12);
    const uint32_t s = a.x ^ b.y ^ c.z ^ d.y;
    if (s == 0x12345678u) sink[0] = s;
}

__global__ __launch_bounds__(256) void calib_pos8(const uint32_t *src, uint32_t *sink)
{
    const size_t i = ((size_t)blockIdx.x * 256u + threadIdx.x) * 6u;
    const u32x2_a4 d = *reinterpret_cast<const u32x2_a4 *>(src + i);
    const uint32_t s = d.x ^ d.y;
    if (s == 0x12345678u) sink[0] = s;
}

int main()
{
    uint8_t *buf;
    uint32_t *sink;
    CHECK(hipMalloc(&buf, N + 4096));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, N + 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t waves = (uint32_t)(N / (64u * STRETCH));
    struct Case { const char *name; double asked; } cases[] = {
        {"wide", (double)N}, {"lane64", (double)N}, {"half64", (double)N / 2}, {"win64", (double)N}, {"head56", (double)N}, {"pos8", (double)N / 3},
    };
    for (int k = 0; k < 6; k++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            switch (k) {
            case 0: hipLaunchKernelGGL(calib_wide, dim3((uint32_t)(N / 16384)), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(buf), sink); break;
            case 1: hipLaunchKernelGGL(calib_lane64<64>, dim3(waves / 4), dim3(256), 0, 0, buf, sink); break;
            case 2: hipLaunchKernelGGL(calib_lane64<128>, dim3(waves / 4), dim3(256), 0, 0, buf, sink); break;
            case 3: hipLaunchKernelGGL(calib_win64, dim3((uint32_t)(N / 16384)), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(buf), sink); break;
            case 4: hipLaunchKernelGGL(calib_head56, dim3((uint32_t)(N / (24 * 256))), dim3(256), 0, 0, reinterpret_cast<const uint32_t *>(buf), sink); break;
            case 5: hipLaunchKernelGGL(calib_pos8, dim3((uint32_t)(N / (24 * 256))), dim3(256), 0, 0, reinterpret_cast<const uint32_t *>(buf), sink); break;
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-7s asked for %.3f GB (lines touched: %.3f GB)  %.3f ms  %.2f TB/s of lines touched\n", cases[k].name, cases[k].asked / 1e9, (double)N / 1e9, best,
               (double)N / best / 1e9);
    }
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
