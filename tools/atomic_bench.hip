// Micro-benchmark: what does a scattered float atomicAdd cost on MI355X, compared with
// plain scattered stores / loads of the same pattern?  (BP sweep design input.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void k(float *buf, const int *idx, const float *val, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int a = idx[i];
        float v = val[i];
        if (MODE == 0) __hip_atomic_fetch_add(buf + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) buf[a] = v;
        if (MODE == 2) { float x = buf[a]; if (x == 12345.f) buf[a] = v; }
        if (MODE == 3) atomicAdd((int *)buf + a, 1);
        if (MODE == 4) __hip_atomic_fetch_add(buf + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 5) { float old = __hip_atomic_fetch_add(buf + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (old == 12345.f) buf[a] = 0; }
    }
}

int main() {
    const long n = 1L << 26;   // 67M ops
    std::vector<int> h(n);
    float *buf, *val; int *idx;
    CK(hipMalloc(&val, n * 4)); CK(hipMalloc(&idx, n * 4));
    CK(hipMemset(val, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *modes[] = {"atomic_f32_agent", "store", "load", "atomic_i32", "atomic_f32_wg", "atomic_f32_ret"};
    for (long words : {1L << 18, 1L << 21, 1L << 24}) {       // 1 MB, 8 MB, 64 MB targets
        CK(hipMalloc(&buf, words * 4)); CK(hipMemset(buf, 0, words * 4));
        for (int pat = 0; pat < 3; pat++) {
            // 0: random, 1: sequential (coalesced), 2: ray-like: each wave walks 64 consecutive
            //    z-columns of a 128^3 grid (stride 128 words between lanes)
            unsigned s = 12345;
            for (long i = 0; i < n; i++) {
                if (pat == 0) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 4) % words); }
                else if (pat == 1) h[i] = (int)(i % words);
                else h[i] = (int)(((i / 64) * 7919 + (i % 64) * 128) % words);
            }
            CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
            for (int m = 0; m < 6; m++) {
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    switch (m) {
                        case 0: hipLaunchKernelGGL(k<0>, 8192, 256, 0, 0, buf, idx, val, n); break;
                        case 1: hipLaunchKernelGGL(k<1>, 8192, 256, 0, 0, buf, idx, val, n); break;
                        case 2: hipLaunchKernelGGL(k<2>, 8192, 256, 0, 0, buf, idx, val, n); break;
                        case 3: hipLaunchKernelGGL(k<3>, 8192, 256, 0, 0, buf, idx, val, n); break;
                        case 4: hipLaunchKernelGGL(k<4>, 8192, 256, 0, 0, buf, idx, val, n); break;
                        case 5: hipLaunchKernelGGL(k<5>, 8192, 256, 0, 0, buf, idx, val, n); break;
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("target %3ld MB pattern %d %-18s %7.3f ms  %7.2f Gop/s\n", words * 4 >> 20, pat, modes[m], best, n / best * 1e-6);
            }
        }
        CK(hipFree(buf));
    }
    return 0;
}
