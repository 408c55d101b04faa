// Micro-benchmark 2: how does the cost of a float atomic instruction depend on the
// arrangement of its 64 addresses inside cache lines?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k(float *buf, const int *idx, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        __hip_atomic_fetch_add(buf + idx[i], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
    const long n = 1L << 25;
    const long words = 1L << 21;   // 8 MB
    std::vector<int> h(n);
    float *buf; int *idx;
    CK(hipMalloc(&buf, words * 4)); CK(hipMalloc(&idx, n * 4)); CK(hipMemset(buf, 0, words * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"sequential", "permuted within 64 dwords", "32 dwords x2 (pair dup)",
                           "16 dwords x4", "1 dword x64 (same addr)", "runs of 32, scattered",
                           "runs of 16, scattered", "runs of 8, scattered", "runs of 4, scattered",
                           "runs of 2, scattered", "random", "permuted within 32 dwords (1 line), lines scattered per half",
                           "sequential but only 32 lanes active addresses distinct lines (stride 32)"};
    for (int pat = 0; pat < 13; pat++) {
        unsigned s = 777;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        for (long w = 0; w < n / 64; w++) {
            long base = (rnd() % (words / 64)) * 64;
            int perm[64];
            for (int l = 0; l < 64; l++) perm[l] = l;
            for (int l = 63; l > 0; l--) std::swap(perm[l], perm[rnd() % (l + 1)]);
            for (int l = 0; l < 64; l++) {
                long a;
                switch (pat) {
                    case 0: a = base + l; break;
                    case 1: a = base + perm[l]; break;
                    case 2: a = base + l / 2; break;
                    case 3: a = base + l / 4; break;
                    case 4: a = base; break;
                    case 5: case 6: case 7: case 8: case 9: {
                        int run = 32 >> (pat - 5);
                        static long rb; if (l % run == 0) rb = (rnd() % (words / 64)) * 64;
                        a = rb + l % run; break; }
                    case 10: a = rnd() % words; break;
                    case 11: { static long hb; if (l % 32 == 0) hb = (rnd() % (words / 32)) * 32; a = hb + perm[l] % 32; break; }
                    default: a = (base + (long)l * 32) % words; break;
                }
                h[w * 64 + l] = (int)a;
            }
        }
        CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, 8192, 256, 0, 0, buf, idx, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("%-72s %7.3f ms %8.2f Gop/s %7.2f Ginstr/s\n", names[pat], best, n / best * 1e-6, n / 64 / best * 1e-6);
    }
    return 0;
}
