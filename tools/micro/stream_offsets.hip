// Do five row streams that advance in lockstep (three read, two written, one 480-byte row per half-wave and step, at most three loads in flight per
// wave: the stage kernels' situation, latency-bound) run faster or slower depending on how far apart the five arrays start?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_offsets tools/micro/stream_offsets.hip && /tmp/stream_offsets [GiB of the one allocation = 24]
// All five arrays live in ONE allocation; array k starts at k * (S + D) for S = 1 474 560 000 bytes (a (60, 3 072 000) fp64 field) and a sweep of gaps D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k5(const v4u *__restrict__ a, const v4u *__restrict__ b, const v4u *__restrict__ c, v4u *__restrict__ o1,
                                          v4u *__restrict__ o2, int64_t nRows)
{
    const int l = threadIdx.x & 31;
    const int64_t hw = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * 256) >> 5;
    if (l >= 30) return;
    for (int64_t r = hw; r < nRows; r += nhw) {
        const int64_t off = r * 30 + l;
        const v4u x = a[off], y = b[off], z = c[off];
        o1[off] = x ^ y;
        o2[off] = x ^ z;
    }
}

int main(int argc, char **argv)
{
    const size_t GiB = (size_t)1 << 30, total = (size_t)(argc > 1 ? atoi(argv[1]) : 24) * GiB;
    const size_t S = 1474560000ull;
    unsigned char *base;
    CK(hipMalloc((void **)&base, total));
    CK(hipMemset(base, 0x5A, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int dev; hipDeviceProp_t pr; CK(hipGetDevice(&dev)); CK(hipGetDeviceProperties(&pr, dev));
    const int grid = pr.multiProcessorCount * 4;          // four workgroups per CU: the stage kernels' occupancy
    const int64_t nRows = S / 480;
    const size_t KiB = 1024, MiB = 1024 * KiB;
    std::vector<size_t> gaps = {0, 480, 4 * KiB, 64 * KiB, 160 * KiB, 256 * KiB, 1 * MiB, 2 * MiB - 4096, 2 * MiB, 2 * MiB + 4096, 3 * MiB, 8 * MiB, 17 * MiB, 32 * MiB,
                                33 * MiB, 64 * MiB, 100 * MiB, 128 * MiB, 256 * MiB, 500 * MiB, 512 * MiB, 1024 * MiB, 0, 2 * MiB, 64 * MiB};
    for (int rep = 0; rep < 2; ++rep)
        for (size_t D : gaps) {
            const size_t step = (S + D + 15) & ~(size_t)15;
            if (4 * step + S > total) continue;
            float best = 1e30f;
            for (int i = 0; i < 4; ++i) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k5, dim3(grid), dim3(256), 0, 0, (const v4u *)base, (const v4u *)(base + step), (const v4u *)(base + 2 * step),
                                   (v4u *)(base + 3 * step), (v4u *)(base + 4 * step), nRows);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (i) best = ms < best ? ms : best;
            }
            printf("rep %d gap %12zu B (%8.2f MiB): %7.3f ms  %7.1f GB/s\n", rep, D, D / 1048576.0, best, 5.0 * S / (best * 1e-3) / 1e9);
            fflush(stdout);
        }
    return 0;
}
