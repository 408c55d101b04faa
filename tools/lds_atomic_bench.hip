// Micro-benchmark 3: throughput of LDS float atomics (ds_add_f32) by address pattern,
// compared with plain LDS read-modify-write of the same addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITER = 4096;
__device__ __forceinline__ int addr_of(int pat, int tid, int it) {
    const int lane = tid & 63, w = tid >> 6;
    unsigned h = (unsigned)(tid * 2654435761u) ^ (unsigned)(it * 40503u);
    switch (pat) {
        case 0: return (tid + it * 256) & 4095;                    // consecutive, conflict-free
        case 1: return ((lane * 32) + w + it) & 4095;              // same bank, distinct addresses
        case 2: return ((lane >> 1) + w * 64 + it * 256) & 4095;   // pairs share an address
        case 3: return ((lane >> 3) + w * 64 + it * 256) & 4095;   // 8 lanes share an address
        case 4: return (w + it) & 4095;                            // 64 lanes share an address
        case 5: h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return h & 4095;   // random
        case 6: return ((lane * 17) + w * 1100 + it * 7) & 4095;   // odd stride
        default: return ((lane * 169) + w * 1100 + it * 7) & 4095; // 13x13 stride (box rows)
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int pat) {
    __shared__ float box[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) box[i] = 0.0f;
    __syncthreads();
    float r = 0.0f;
    for (int it = 0; it < ITER; it++) {
        const int a = addr_of(pat, threadIdx.x, it);
        if (MODE == 0) __hip_atomic_fetch_add(box + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 1) r += __hip_atomic_fetch_add(box + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) atomicAdd(reinterpret_cast<int *>(box) + a, 1);
        else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long *>(box) + (a >> 1), 1ull);
        else if (MODE == 5) __hip_atomic_fetch_add(reinterpret_cast<double *>(box) + (a >> 1), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else { box[a] += 1.0f; }          // non-atomic read-modify-write (wrong sums, timing only)
    }
    __syncthreads();
    float s = r;
    for (int i = threadIdx.x; i < 4096; i += 256) s += box[i];
    if (s == -1.0f) out[0] = s;
}
int main() {
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *pn[] = {"consecutive", "same bank", "pairs dup", "8-fold dup", "64-fold dup", "random", "stride 17", "stride 169"};
    const char *mn[] = {"ds_add_f32", "ds_add_rtn_f32", "ds_add_u32", "plain rmw", "ds_add_u64", "ds_add_f64"};
    const int blocks = 256 * 8;
    for (int mode = 0; mode < 6; mode++)
        for (int pat = 0; pat < 8; pat++) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, blocks, 256, 0, 0, out, pat);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, blocks, 256, 0, 0, out, pat);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, blocks, 256, 0, 0, out, pat);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, blocks, 256, 0, 0, out, pat);
                else if (mode == 4) hipLaunchKernelGGL(k<4>, blocks, 256, 0, 0, out, pat);
                else hipLaunchKernelGGL(k<5>, blocks, 256, 0, 0, out, pat);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double ops = (double)blocks * 256 * ITER;
            printf("%-15s %-12s %8.3f ms  %8.1f G lane-ops/s  (%.2f per clk per CU @2.4GHz)\n", mn[mode], pn[pat], best,
                   ops / best / 1e6, ops / best / 1e6 / 256 / 2.4);
        }
    return 0;
}
