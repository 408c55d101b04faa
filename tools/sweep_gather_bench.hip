// sweep_gather_bench.hip -- the plane sweep's GATHERS alone, in different orders (round 5).
//
// Question: what would another decomposition of k_sweep_map buy on the
// memory side, before building it?  Every kernel here performs exactly the feature-vector loads
// of the plane sweep -- for every live ray, every neighbour view and every depth plane the 128-byte
// vector the real sweep gathers (the offsets come from the library's own index arithmetic,
// rn_selftest_feature_offsets) -- and nothing else: no projection, no pair products, no mapping.
// What differs is WHO loads WHAT WHEN:
//   k_ray_coop      the product's order: one wavefront per ray, 64 planes per chunk, 8 lanes fetch
//                   one vector (16 B each), 8 planes per load round
//   k_ray_lane      one wavefront per ray, lane = plane: a lane fetches its plane's whole vector
//                   (8 x 16 B), no cooperation, no exchange of offsets
//   k_tile_blocked  a workgroup owns a 16 x 16-pixel tile of rays (256 consecutive patch-ordered
//                   rows) and walks the D planes in bands of PB planes: every wavefront takes
//                   64 / PB rays x PB planes at a time, the tile finishes a band before any of its
//                   wavefronts starts the next (the partial columns of the real kernel would wait
//                   in LDS: D x 256 x 4 bytes, which is what limits it to one workgroup per CU --
//                   emulated with `lds_bytes` of dynamic LDS)
//   k_tile_lds      (round 6, VERDICT r5 item 1: the FOOTPRINT-STAGED tile) the same tile and bands,
//                   but per (neighbour view, band) the workgroup first finds the band's footprint in
//                   that view's map -- per map row the [xmin, xmax] its 256 x PB samples touch, an
//                   oblique parallelogram -- loads those row segments ONCE, densely (LDS-DMA,
//                   global_load_lds_dwordx4: 8 consecutive pixels = 1 KB per wavefront instruction),
//                   and serves the 8-lane vector reads from LDS (ds_read_b128).  A band whose
//                   footprint does not fit `lds_bytes` (or spans more than 64 map rows) gathers from
//                   global memory as k_tile_blocked does; `stats` counts bands / fallbacks / staged
//                   pixels.  The partial columns a real kernel would also hold in LDS (256 x D x 4
//                   bytes) are NOT allocated here: the variant measures the memory side at its best.
// Timed with events and counted with rocprofv3 --pmc by tools/sweep_gather_bench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int WAVE = 64, NXCD = 8;
typedef float float4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) char *gptr;
typedef const __attribute__((address_space(1))) float4v *gptr4;

struct Views {
    const float *v[16];
};

__device__ __forceinline__ int xcd_block_rt(int b, int nblocks, int chunk) {
    const int full = nblocks / (NXCD * chunk) * (NXCD * chunk);
    if (b >= full) return b;
    const int xcd = b % NXCD, pos = b / NXCD;
    return ((pos / chunk) * NXCD + xcd) * chunk + pos % chunk;
}

__device__ __forceinline__ float sum4(float4v f) { return (f.x + f.y) + (f.z + f.w); }

// one 16-byte load with a cache policy: 0 default, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt
template <int POL>
__device__ __forceinline__ float4v load16(const float *base, unsigned off) {
    float4v f;
    if (POL == 0) return *(gptr4)((gptr)base + off);
    if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(f) : "v"(off), "s"(base) : "memory");
    if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc0" : "=v"(f) : "v"(off), "s"(base) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(f) : "v"(off), "s"(base) : "memory");
    if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(f) : "v"(off), "s"(base) : "memory");
    if (POL == 5) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1 nt" : "=v"(f) : "v"(off), "s"(base) : "memory");
    return f;
}

// the product's order with a cache policy on the gathers (asm loads: waited for once per chunk)
template <int NV, int POL>
__global__ __launch_bounds__(256) void k_ray_coop_pol(int n, int D, const int32_t *__restrict__ offs,
                                                      const int32_t *__restrict__ live, Views fv,
                                                      float *out, int chunk) {
    const int lane = threadIdx.x & 63;
    const int b = xcd_block_rt(blockIdx.x, (n + 3) / 4, chunk);
    const int r = __builtin_amdgcn_readfirstlane(b * 4 + (int)(threadIdx.x >> 6));
    if (r >= n || live[r] <= 1) return;
    const int sub = lane >> 3, part = lane & 7;
    float acc = 0.f;
    for (int base = 0; base < D; base += WAVE) {
        int ob[NV];
#pragma unroll
        for (int v = 1; v < NV; v++)
            ob[v] = offs[((size_t)r * NV + v) * D + min(base + lane, D - 1)] << 7;
        float4v f[8][NV];
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int v = 1; v < NV; v++)
                f[t][v] = load16<POL>(fv.v[v], (unsigned)__shfl(ob[v], t * 8 + sub) + 16u * part);
        if (POL != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int v = 1; v < NV; v++) acc += sum4(f[t][v]);
    }
    out[(size_t)r * WAVE + lane] = acc;
}

// one wavefront per ray, the product's cooperative rounds
template <int NV>
__global__ __launch_bounds__(256) void k_ray_coop(int n, int D, const int32_t *__restrict__ offs,
                                                  const int32_t *__restrict__ live, Views fv,
                                                  float *out, int chunk) {
    extern __shared__ float lds_cap[];                   // (only there to cap occupancy: --lds)
    const int lane = threadIdx.x & 63;
    const int b = xcd_block_rt(blockIdx.x, (n + 3) / 4, chunk);
    const int r = __builtin_amdgcn_readfirstlane(b * 4 + (int)(threadIdx.x >> 6));
    if (r >= n || live[r] <= 1) return;
    if (n < 0) lds_cap[lane] = 0.f;
    const int sub = lane >> 3, part = lane & 7;
    float acc = 0.f;
    for (int base = 0; base < D; base += WAVE) {
        int ob[NV];
#pragma unroll
        for (int v = 1; v < NV; v++)
            ob[v] = offs[((size_t)r * NV + v) * D + min(base + lane, D - 1)] << 7;
#pragma unroll
        for (int t = 0; t < 8; t++) {
#pragma unroll
            for (int v = 1; v < NV; v++) {
                const unsigned o = (unsigned)__shfl(ob[v], t * 8 + sub) + 16u * part;
                acc += sum4(*(gptr4)((gptr)fv.v[v] + o));
            }
        }
    }
    out[(size_t)r * WAVE + lane] = acc;
}

// one wavefront per ray, lane = plane
template <int NV>
__global__ __launch_bounds__(256) void k_ray_lane(int n, int D, const int32_t *__restrict__ offs,
                                                  const int32_t *__restrict__ live, Views fv,
                                                  float *out, int chunk) {
    const int lane = threadIdx.x & 63;
    const int b = xcd_block_rt(blockIdx.x, (n + 3) / 4, chunk);
    const int r = __builtin_amdgcn_readfirstlane(b * 4 + (int)(threadIdx.x >> 6));
    if (r >= n || live[r] <= 1) return;
    float acc = 0.f;
    for (int base = 0; base < D; base += WAVE) {
        if (base + lane < D) {
#pragma unroll
            for (int v = 1; v < NV; v++) {
                const unsigned o = (unsigned)offs[((size_t)r * NV + v) * D + base + lane] << 7;
#pragma unroll
                for (int q = 0; q < 8; q++) acc += sum4(*(gptr4)((gptr)fv.v[v] + o + 16u * q));
            }
        }
    }
    out[(size_t)r * WAVE + lane] = acc;
}

// a workgroup per 256-row tile, planes in bands of PB; WAVES wavefronts per workgroup
template <int NV, int PB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_tile_blocked(int n, int D,
                                                             const int32_t *__restrict__ offs,
                                                             const int32_t *__restrict__ live,
                                                             Views fv, float *out, int chunk) {
    extern __shared__ float lds[];
    constexpr int RPW = WAVE / PB;                       // rays per wavefront step
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ntiles = (n + 255) / 256;
    const int tile = xcd_block_rt(blockIdx.x, ntiles, chunk);
    const int row0 = tile * 256;
    const int rsub = lane / PB, plane = lane % PB;
    const int sub = lane >> 3, part = lane & 7;
    float acc = 0.f;
    if (threadIdx.x == 0) lds[0] = 0.f;                  // (the LDS is only there to cap occupancy)
    for (int band = 0; band < D; band += PB) {
        for (int it = w; it < 256 / RPW; it += WAVES) {
            const int r = row0 + it * RPW + rsub;        // this lane's (ray, plane) sample
            const bool ok = r < n && live[min(r, n - 1)] > 1;
            int ob[NV];
#pragma unroll
            for (int v = 1; v < NV; v++)
                ob[v] = ok ? (offs[((size_t)r * NV + v) * D + band + plane] << 7) : -1;
#pragma unroll
            for (int t = 0; t < 8; t++) {
#pragma unroll
                for (int v = 1; v < NV; v++) {
                    // (a dead ray's lanes fetch vector 0: no branch around a load)
                    const int o = max(__shfl(ob[v], t * 8 + sub), 0);
                    acc += sum4(*(gptr4)((gptr)fv.v[v] + (unsigned)o + 16u * part));
                }
            }
        }
        __syncthreads();                                 // the tile finishes a band together
    }
    out[(size_t)blockIdx.x * (WAVES * 64) + threadIdx.x] = acc;
}

// footprint-staged tile: see the header.  LDS: [64 rows: xmin, xmax, base][4 scalars][stage bytes]
template <int NV, int PB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_tile_lds(int n, int D, int Wf,
                                                         const int32_t *__restrict__ offs,
                                                         const int32_t *__restrict__ live,
                                                         Views fv, float *out, int chunk,
                                                         int stage_bytes, unsigned *stats) {
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    constexpr int ROWS = 64;
    constexpr int RPW = WAVE / PB;                       // rays per wavefront step
    constexpr int ITS = 256 / RPW / WAVES;               // samples a lane holds per (view, band)
    static_assert(ITS >= 1, "PB too small for this many waves");
    int *rmin = lds_i, *rmax = lds_i + ROWS, *rbase = lds_i + 2 * ROWS, *scal = lds_i + 3 * ROWS;
    char *stage = reinterpret_cast<char *>(lds_i + 3 * ROWS + 8);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ntiles = (n + 255) / 256;
    const int tile = xcd_block_rt(blockIdx.x, ntiles, chunk);
    const int row0 = tile * 256;
    const int rsub = lane / PB, plane = lane % PB;
    const int sub = lane >> 3, part = lane & 7;
    float acc = 0.f;
    unsigned n_bands = 0, n_fallback = 0, n_pixels = 0;
    for (int band = 0; band < D; band += PB) {
        for (int v = 1; v < NV; v++) {
            // ---- the lane's samples of this (view, band)
            int px[ITS];                                  // vector index in the view's map, -1: dead
#pragma unroll
            for (int k = 0; k < ITS; k++) {
                const int it = w + k * WAVES;
                const int r = row0 + it * RPW + rsub;
                const bool ok = r < n && live[min(r, n - 1)] > 1;
                px[k] = ok ? offs[((size_t)r * NV + v) * D + band + plane] : -1;
            }
            // ---- footprint: per map row (direct-mapped on fy & 63) the touched [xmin, xmax]
            for (int i = threadIdx.x; i < ROWS; i += WAVES * 64) {
                rmin[i] = 0x7fffffff;
                rmax[i] = -1;
            }
            if (threadIdx.x == 0) {
                scal[0] = 0x7fffffff;                     // ymin
                scal[1] = -1;                             // ymax
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < ITS; k++) {
                if (px[k] >= 0) {
                    const int fy = px[k] / Wf, fx = px[k] - fy * Wf;
                    atomicMin(&rmin[fy & (ROWS - 1)], fx);
                    atomicMax(&rmax[fy & (ROWS - 1)], fx);
                    atomicMin(&scal[0], fy);
                    atomicMax(&scal[1], fy);
                }
            }
            __syncthreads();
            // ---- row bases (one wavefront: 64 rows), segments padded to whole 8-pixel loads
            if (w == 0) {
                const int wd = rmax[lane] >= 0 ? ((rmax[lane] - rmin[lane] + 8) & ~7) : 0;
                int incl = wd;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                rbase[lane] = incl - wd;
                if (lane == 63) scal[2] = incl;           // pixels staged
            }
            __syncthreads();
            const int ymin = scal[0], ymax = scal[1], total = scal[2];
            const bool staged = ymax >= 0 && ymax - ymin < ROWS && total * 128 <= stage_bytes;
            n_bands++;
            if (ymax < 0) continue;                       // a dead tile (uniform)
            if (staged) {
                n_pixels += total;
                // ---- dense fill: a wavefront instruction = 8 consecutive pixels of one map row
                for (int y = ymin + w; y <= ymax; y += WAVES) {
                    const int yr = y & (ROWS - 1);
                    const int x0 = rmin[yr], x1 = rmax[yr];
                    if (x1 < 0) continue;
                    const char *src = reinterpret_cast<const char *>(fv.v[v]) + ((size_t)y * Wf + x0) * 128;
                    char *dst = stage + (size_t)rbase[yr] * 128;
                    const int room = (Wf - x0) * 128;     // bytes left in this map row (no over-read past it)
                    for (int x = 0; x <= x1 - x0; x += 8) {
                        typedef const __attribute__((address_space(1))) void *gp;
                        typedef __attribute__((address_space(3))) void *lp;
                        if (x * 128 + lane * 16 < room)
                            __builtin_amdgcn_global_load_lds((gp)(src + x * 128 + lane * 16),
                                                             (lp)(dst + x * 128), 16, 0, 0);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // ---- the 8-lane vector reads, from LDS
#pragma unroll
                for (int k = 0; k < ITS; k++) {
                    int a = -1;
                    if (px[k] >= 0) {
                        const int fy = px[k] / Wf, fx = px[k] - fy * Wf, yr = fy & (ROWS - 1);
                        a = (rbase[yr] + fx - rmin[yr]) * 128;
                    }
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int o = __shfl(a, t * 8 + sub);
                        if (o >= 0) acc += sum4(*reinterpret_cast<const float4v *>(stage + o + 16 * part));
                    }
                }
            } else {
                n_fallback++;
#pragma unroll
                for (int k = 0; k < ITS; k++) {
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int o = max(__shfl(px[k], t * 8 + sub), 0);
                        acc += sum4(*(gptr4)((gptr)fv.v[v] + ((unsigned)o << 7) + 16u * part));
                    }
                }
            }
            __syncthreads();                              // the stage is free again
        }
    }
    out[(size_t)blockIdx.x * (WAVES * 64) + threadIdx.x] = acc;
    if (threadIdx.x == 0 && stats) {
        atomicAdd(&stats[0], n_bands);
        atomicAdd(&stats[1], n_fallback);
        atomicAdd(&stats[2], n_pixels >> 4);              // (in units of 16 pixels)
    }
}
}  // namespace

static int g_wf = 0;
static unsigned *g_stats = nullptr;
extern "C" void sgb_set_lds_variant(int Wf, unsigned *stats) {
    g_wf = Wf;
    g_stats = stats;
}

template <int NV>
static int launch(int variant, int n, int D, const int32_t *offs, const int32_t *live, Views fv,
                  float *out, int chunk, int lds_bytes, hipStream_t st) {
    const int ntiles = (n + 255) / 256;
#define TILE(PB, WV)                                                                              \
    {                                                                                             \
        auto k = k_tile_blocked<NV, PB, WV>;                                                      \
        if (lds_bytes > 65536)                                                                    \
            (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      lds_bytes);                                                 \
        hipLaunchKernelGGL(k, dim3(ntiles), dim3(WV * 64), lds_bytes, st, n, D, offs, live, fv,   \
                           out, chunk);                                                           \
    }
    switch (variant) {
        case 0: {
            auto k = k_ray_coop<NV>;
            if (lds_bytes > 65536)
                (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            hipLaunchKernelGGL(k, dim3((n + 3) / 4), dim3(256), lds_bytes, st, n, D, offs, live, fv, out, chunk);
            break;
        }
        case 1: hipLaunchKernelGGL(k_ray_lane<NV>, dim3((n + 3) / 4), dim3(256), 0, st, n, D, offs, live, fv, out, chunk); break;
        case 10: hipLaunchKernelGGL((k_ray_coop_pol<NV, 0>), dim3((n + 3) / 4), dim3(256), 0, st, n, D, offs, live, fv, out, chunk); break;
        case 11: hipLaunchKernelGGL((k_ray_coop_pol<NV, 1>), dim3((n + 3) / 4), dim3(256), 0, st, n, D, offs, live, fv, out, chunk); break;
        case 12: hipLaunchKernelGGL((k_ray_coop_pol<NV, 2>), dim3((n + 3) / 4), dim3(256), 0, st, n, D, offs, live, fv, out, chunk); break;
        case 13: hipLaunchKernelGGL((k_ray_coop_pol<NV, 3>), dim3((n + 3) / 4), dim3(256), 0, st, n, D, offs, live, fv, out, chunk); break;
        case 14: hipLaunchKernelGGL((k_ray_coop_pol<NV, 4>), dim3((n + 3) / 4), dim3(256), 0, st, n, D, offs, live, fv, out, chunk); break;
        case 15: hipLaunchKernelGGL((k_ray_coop_pol<NV, 5>), dim3((n + 3) / 4), dim3(256), 0, st, n, D, offs, live, fv, out, chunk); break;
        case 2: TILE(16, 16) break;
        case 3: TILE(8, 16) break;
        case 4: TILE(32, 16) break;
        case 5: TILE(16, 8) break;
        case 6: TILE(64, 16) break;
#define TILE_LDS(PB, WV)                                                                          \
    {                                                                                             \
        auto k = k_tile_lds<NV, PB, WV>;                                                          \
        const int total = lds_bytes + (3 * 64 + 8) * 4;                                           \
        (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, total); \
        hipLaunchKernelGGL(k, dim3(ntiles), dim3(WV * 64), total, st, n, D, g_wf, offs, live, fv,  \
                           out, chunk, lds_bytes, g_stats);                                       \
    }
        case 20: TILE_LDS(8, 16) break;
        case 21: TILE_LDS(4, 16) break;
        case 22: TILE_LDS(16, 16) break;
        case 23: TILE_LDS(8, 8) break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

extern "C" int sgb_run(int variant, int n, int NV, int D, const int32_t *offs, const int32_t *live,
                       const float *const *views_host, float *out, int chunk, int lds_bytes,
                       void *stream) {
    Views fv;
    for (int v = 0; v < NV; v++) fv.v[v] = views_host[v];
    hipStream_t st = (hipStream_t)stream;
    if (NV == 5) return launch<5>(variant, n, D, offs, live, fv, out, chunk, lds_bytes, st);
    if (NV == 9) return launch<9>(variant, n, D, offs, live, fv, out, chunk, lds_bytes, st);
    return -1;
}
