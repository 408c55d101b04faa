// Micro-benchmark 4: what a CU's vector L1 delivers for the plane sweep's access pattern --
// a wavefront instruction of 64 x 16 B in which 8 lanes share one 128-B line and the 8 lines
// are scattered over a footprint -- as a function of where the footprint lives:
//   <= 16 KB per CU        L1 hits (data return path only)
//   a few MB               L1 misses served by the XCD's L2 (return + fill)
//   hundreds of MB         Infinity Cache / HBM
// Reported: bytes per clock and CU at the measured kernel time and 2.4 GHz, and GB/s chip-wide.
//   hipcc --offload-arch=gfx950 -O3 tools/l1_gather_bench.hip -o tools/l1_gather_bench && tools/l1_gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;
constexpr int UNROLL = 8;          // loads in flight per wave, like two load rounds of the sweep

// private = 1: every workgroup gathers from its own window of `lines` lines (L1-resident when
// small); private = 0: all workgroups gather from the same footprint of `lines` lines
__global__ __launch_bounds__(256) void k_gather(const float4v *__restrict__ src, long lines, int priv,
                                                float *out) {
    const int lane = threadIdx.x & 63, grp = lane >> 3, part = lane & 7;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long window = priv ? (long)blockIdx.x * lines : 0;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    unsigned h = wave * 2654435761u + grp * 40503u;
    for (int it = 0; it < ITER; it += UNROLL) {
        float4v v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            h = h * 1664525u + 1013904223u;
            const long line = window + (long)((h >> 8) % (unsigned long)lines);
            v[u] = src[line * 8 + part];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.0f) out[0] = acc.x;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t max_bytes = 1ull << 30;
    float4v *src;
    float *out;
    CK(hipMalloc(&src, max_bytes));
    CK(hipMemset(src, 0, max_bytes));
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = cus * 6;         // 24 waves per CU, the sweep's occupancy
    struct Case { const char *name; long lines; int priv; };
    const Case cases[] = {
        {"8 KB per workgroup (L1 hits)", 64, 1},
        {"32 KB per workgroup", 256, 1},
        {"128 KB per workgroup (L2)", 1024, 1},
        {"2 MB shared (L2 of every XCD)", 16384, 0},
        {"16 MB shared (L2, 4 MB per XCD)", 131072, 0},
        {"205 MB shared (Infinity Cache)", 1601562, 0},
        {"1 GB shared (HBM)", 8388608, 0},
    };
    printf("%d CUs, %d workgroups x 4 waves, %d x 1 KB loads per wave\n", cus, blocks, ITER);
    for (const Case &c : cases) {
        if ((c.priv ? (size_t)c.lines * blocks : (size_t)c.lines) * 128 > max_bytes) continue;
        hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, src, c.lines, c.priv, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, src, c.lines, c.priv, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)blocks * 4 * ITER * 1024.0;
        printf("%-36s %8.3f ms  %7.1f GB/s  %5.1f B/clk/CU at 2.4 GHz\n", c.name, ms,
               bytes / ms / 1e6, bytes / (ms * 1e-3) / cus / 2.4e9);
    }
    return 0;
}
