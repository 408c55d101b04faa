// How fast do the BP kernels' row streams go, and does the row LAYOUT matter?  One wavefront
// per ray reads `count` entries of three rows and writes one (k_bp's memory side without the
// gather), rows either M = 384 slots apart (the resident layout: a mean ray uses 137 of them)
// or packed back to back (CSR).   hipcc --offload-arch=gfx950 -O3 tools/row_stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k(int n, const int *__restrict__ cnt, const long long *__restrict__ off,
                                         const float *__restrict__ a, const int *__restrict__ b,
                                         const float *__restrict__ c, float *d, int nt) {
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n) return;
    const int lane = threadIdx.x & 63, count = cnt[r];
    const long long o = off[r];
    for (int i = lane; i < count; i += 64) {
        float x = a[o + i] + c[o + i] + (float)b[o + i];
        if (nt) __builtin_nontemporal_store(x, d + o + i); else d[o + i] = x;
    }
}
// the same, and before it ends every wavefront requests the rows of ray r + K (results
// unused): a later wavefront then finds its rows in the L2 -- a prefetch ACROSS wavefronts that
// costs no registers in the wavefront that profits
__global__ __launch_bounds__(256) void kp(int n, const int *__restrict__ cnt, const long long *__restrict__ off,
                                          const float *__restrict__ a, const int *__restrict__ b,
                                          const float *__restrict__ c, float *d, int nt, int K) {
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n) return;
    const int lane = threadIdx.x & 63, count = cnt[r];
    const long long o = off[r];
    const int rp = r + K;
    int cp = 0; long long op = 0;
    if (rp < n) { cp = cnt[rp]; op = off[rp]; }
    for (int i = lane; i < count; i += 64) {
        float x = a[o + i] + c[o + i] + (float)b[o + i];
        if (nt) __builtin_nontemporal_store(x, d + o + i); else d[o + i] = x;
    }
    // 16 bytes per lane: lanes 0 .. ceil(cp / 4) - 1 cover the future ray's rows
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 pa, pb, pc;
    bool did = false;
    if (4 * lane < cp) {
        const v4 *qa = reinterpret_cast<const v4 *>(a + op) + lane;
        const v4 *qb = reinterpret_cast<const v4 *>(b + op) + lane;
        const v4 *qc = reinterpret_cast<const v4 *>(c + op) + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pa) : "v"(qa));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pb) : "v"(qb));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pc) : "v"(qc));
        did = true;
    }
    if (did) asm volatile("" ::"v"(pa), "v"(pb), "v"(pc));   // keeps the registers reserved to the end
}
int main() {
    const int n = 1536000, M = 384;
    std::vector<int> cnt(n); std::vector<long long> offs(n), offc(n);
    srand(1); long long tot = 0;
    for (int i = 0; i < n; i++) { int c = 2 + rand() % 271; cnt[i] = c; offs[i] = (long long)i * M; offc[i] = tot; tot += (c + 15) / 16 * 16; }
    printf("rays %d, mean count %.1f, strided %.2f GB per array, packed %.2f GB\n", n, (double)tot / n, n * (double)M * 4 / 1e9, tot * 4.0 / 1e9);
    int *dcnt; long long *doff; float *a, *c, *d; int *b;
    hipMalloc(&dcnt, n * 4); hipMalloc(&doff, n * 8);
    size_t bytes = (size_t)n * M * 4;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&d, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    hipMemcpy(dcnt, cnt.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double used = 0; for (int i = 0; i < n; i++) used += cnt[i];
    for (int layout = 0; layout < 2; layout++) for (int nt = 0; nt < 2; nt++) {
        hipMemcpy(doff, layout ? offc.data() : offs.data(), n * 8, hipMemcpyHostToDevice);
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3((n + 3) / 4), dim3(256), 0, 0, n, dcnt, doff, a, b, c, d, nt);
        hipEventRecord(e0);
        for (int w = 0; w < 10; w++) hipLaunchKernelGGL(k, dim3((n + 3) / 4), dim3(256), 0, 0, n, dcnt, doff, a, b, c, d, nt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%s rows, %s stores: %.3f ms, %.0f GB/s (12 B read + 4 B written per entry)\n", layout ? "packed " : "strided", nt ? "non-temporal" : "plain       ", ms, used * 16 / ms / 1e6);
    }
    hipMemcpy(doff, offs.data(), n * 8, hipMemcpyHostToDevice);
    for (int K : {0, 256, 1024, 4096, 16384}) {
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kp, dim3((n + 3) / 4), dim3(256), 0, 0, n, dcnt, doff, a, b, c, d, 1, K ? K : n);
        hipEventRecord(e0);
        for (int w = 0; w < 10; w++) hipLaunchKernelGGL(kp, dim3((n + 3) / 4), dim3(256), 0, 0, n, dcnt, doff, a, b, c, d, 1, K ? K : n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("strided rows, non-temporal stores, rows of ray r + %d requested ahead: %.3f ms, %.0f GB/s\n", K, ms, used * 16 / ms / 1e6);
    }
    return 0;
}
