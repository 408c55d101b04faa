// Round 6 microbenchmark: do k_bp-like row streams run faster on COMPACT rows?
// One wavefront per ray reads the first c_r entries of three [n][*] float arrays and writes one
// (k_bp's 16 HBM bytes per voxel visit, no arithmetic, no accumulator gather), c_r from a
// distribution like config 2's (mean ~137 of M = 384), in two layouts:
//   padded : row r at r * M            (the product's: 1536-byte rows, ~550 bytes of each used)
//   compact: row r at off[r]           (rows back to back, 16-byte aligned)
// hipcc --offload-arch=gfx950 -O3 tools/row_stream_bench.hip -o /tmp/row_stream_bench && /tmp/row_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool COMPACT, bool NT>
__global__ __launch_bounds__(256) void k_rows(int n, int M, const int *__restrict__ cnt,
                                              const long long *__restrict__ off,
                                              const float *__restrict__ a, const float *__restrict__ b,
                                              const float *__restrict__ c, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n) return;
    const int count = __builtin_amdgcn_readfirstlane(cnt[r]);
    const long long base = COMPACT ? off[r] : (long long)r * M;
    float va[6], vb[6], vc[6];
#pragma unroll
    for (int ch = 0; ch < 6; ch++) {
        const int i = ch * 64 + lane;
        va[ch] = vb[ch] = vc[ch] = 0.f;
        if (ch * 64 < count && i < count) {
            if (NT) {
                va[ch] = __builtin_nontemporal_load(a + base + i);
                vb[ch] = __builtin_nontemporal_load(b + base + i);
                vc[ch] = __builtin_nontemporal_load(c + base + i);
            } else {
                va[ch] = a[base + i]; vb[ch] = b[base + i]; vc[ch] = c[base + i];
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 6; ch++) {
        const int i = ch * 64 + lane;
        if (ch * 64 < count && i < count) {
            const float v = va[ch] + vb[ch] + vc[ch];
            if (NT) __builtin_nontemporal_store(v, out + base + i); else out[base + i] = v;
        }
    }
}

template <bool NT_PAD, bool NT_CMP>
__global__ __launch_bounds__(256) void k_rows_mixed(int n, int M, const int *__restrict__ cnt,
                                                    const long long *__restrict__ off,
                                                    const float *__restrict__ a, const float *__restrict__ b,
                                                    const float *__restrict__ c, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n) return;
    const int count = __builtin_amdgcn_readfirstlane(cnt[r]);
    const long long bp = (long long)r * M, bc = off[r];
    float va[6], vb[6], vc[6];
#pragma unroll
    for (int ch = 0; ch < 6; ch++) {
        const int i = ch * 64 + lane;
        va[ch] = vb[ch] = vc[ch] = 0.f;
        if (ch * 64 < count && i < count) {
            va[ch] = NT_PAD ? __builtin_nontemporal_load(a + bp + i) : a[bp + i];
            vb[ch] = NT_CMP ? __builtin_nontemporal_load(b + bc + i) : b[bc + i];
            vc[ch] = NT_CMP ? __builtin_nontemporal_load(c + bc + i) : c[bc + i];
        }
    }
#pragma unroll
    for (int ch = 0; ch < 6; ch++) {
        const int i = ch * 64 + lane;
        if (ch * 64 < count && i < count) {
            const float v = va[ch] + vb[ch] + vc[ch];
            if (NT_CMP) __builtin_nontemporal_store(v, out + bc + i); else out[bc + i] = v;
        }
    }
}

int main() {
    const int n = 1536000, M = 384;
    std::vector<int> cnt(n);
    std::vector<long long> off(n);
    srand(7);
    long long visits = 0, o = 0;
    for (int r = 0; r < n; r++) {
        // 5 % empty rays, the rest 40 .. 250 (mean ~145), neighbouring rows similar (patch order)
        int c = (rand() % 100 < 5) ? 0 : 40 + ((r / 256) * 37 % 170) + rand() % 40;
        cnt[r] = c; off[r] = o; o += (c + 31) & ~31; visits += c;
    }
    printf("rays %d, mean count %.1f, padded %.2f GB per array, compact %.2f GB\n", n, (double)visits / n,
           (double)n * M * 4 / 1e9, (double)o * 4 / 1e9);
    int *d_cnt; long long *d_off; float *A, *B, *C, *O;
    CK(hipMalloc(&d_cnt, n * 4)); CK(hipMalloc(&d_off, n * 8));
    const size_t bytes = (size_t)n * M * 4;
    CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMalloc(&C, bytes)); CK(hipMalloc(&O, bytes));
    CK(hipMemset(A, 0, bytes)); CK(hipMemset(B, 0, bytes)); CK(hipMemset(C, 0, bytes));
    CK(hipMemcpy(d_cnt, cnt.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, off.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kernel) {
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kernel, dim3((n + 3) / 4), dim3(256), 0, 0, n, M, d_cnt, d_off, A, B, C, O);
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kernel, dim3((n + 3) / 4), dim3(256), 0, 0, n, M, d_cnt, d_off, A, B, C, O);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-28s %.3f ms  %.0f GB/s (16 bytes per visit)\n", name, ms, 16.0 * visits / ms / 1e6);
    };
    run("padded rows", k_rows<false, false>);
    run("compact rows", k_rows<true, false>);
    run("padded rows, non-temporal", k_rows<false, true>);
    run("compact rows, non-temporal", k_rows<true, true>);
    run("1 padded (nt) + 3 compact", k_rows_mixed<true, false>);
    run("1 padded + 3 compact", k_rows_mixed<false, false>);
    run("1 padded (nt) + 3 compact (nt)", k_rows_mixed<true, true>);
    return 0;
}
