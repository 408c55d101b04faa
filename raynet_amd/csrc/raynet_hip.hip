// raynet_hip.hip -- __global__ kernels and the C ABI (include/raynet_hip.h) of the
// RayNet forward_pass hot path for MI355X / gfx950.  No CPU fallback lives here.
#include "raynet_kernels.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/raynet_hip.h"

using namespace rn;

namespace {

constexpr int BLOCK = 256;               // 4 wavefronts = 4 rays per workgroup
constexpr int BOX_PROBE_LAUNCHES = 12;   // scatter launches after a reset whose overflow counters are read back
constexpr int RAY_BLOCK = 256;           // threads per workgroup of the wave-per-ray MRF kernels (k_bp, k_depth)
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
// The plane sweep's own workgroup size.  The hardware deals consecutive
// workgroups to different CUs, so only a workgroup's own rays share a CU's L1, and the kernel runs
// at the rate its L1s get lines from the L2 -- yet more rays per workgroup do not pay: 384 / 512 /
// 768 / 1024 threads are 22 / 7 / 13 / 36 % slower than 256 at config 2 and 14 / 2 / 14 / 10 % at
// config 4 (profiles/r05_exp_sweep_block.txt; a first run that seemed to gain 11 % had silently
// dropped the folded first BP iteration, whose LDS rows no longer fitted).
constexpr int SWEEP_BLOCK = 256;
constexpr int SWEEP_WAVES = SWEEP_BLOCK / WAVE;
constexpr int NXCD = 8;

// XCD-aware remap: the dispatcher is observed to place block b on XCD b % 8; give
// every XCD one contiguous slice of the ray list so its private L2 sees
// neighbouring rays (speed only -- any placement is correct).
// CHUNK = 0: every XCD gets one contiguous eighth of the list.  CHUNK > 0: chunks of CHUNK
// consecutive work items stay on one XCD and the chunks go round the XCDs -- the XCDs then walk
// the list side by side.  The work per item varies along the list (rays through the middle of
// the box are long, border tiles are empty), so contiguous eighths leave some XCDs idle at the
// end of a launch; what a kernel gains from locality decides its chunk
// (profiles/r03_exp_xcd_chunk.txt): the plane sweep lives off neighbouring rays sharing feature
// rows in L2, the scatter wants the balance.  (The sweep had its eighth until its list loads
// became non-temporal; re-measured after that, chunks of 256 - 1024 rays -- one to four 16 x 16
// pixel tiles -- are 1.5 % faster at config 2 and 6.5 % at config 4, 64 rays are slower, 16384
// much slower.)
#define RN_XCD_CHUNK_SWEEP (2048 * 64 / 256)      /* workgroups: 2048 rays */
#define RN_XCD_CHUNK_BP 256
#define RN_XCD_CHUNK_DEPTH 1024    /* (contiguous eighths until round 6: config 4's k_depth 2.50 -> 2.32 ms per step with chunks of 1024 workgroups, config 2 within noise -- profiles/r06_l_variants_probe.txt) */
#define RN_XCD_CHUNK_SCATTER 8
template <int CHUNK = 0>
__device__ __forceinline__ int xcd_block(int b, int nblocks) {
    if (CHUNK > 0) {
        constexpr int C = CHUNK > 0 ? CHUNK : 1;
        const int full = nblocks / (NXCD * C) * (NXCD * C);
        if (b >= full) return b;
        const int xcd = b % NXCD, pos = b / NXCD;
        return ((pos / C) * NXCD + xcd) * C + pos % C;
    }
    const int q = nblocks / NXCD, r = nblocks % NXCD;
    const int xcd = b % NXCD, pos = b / NXCD;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + pos;
}

// the same with a RUN-TIME chunk (work items per XCD group), for launches whose caller sets it
__device__ __forceinline__ int xcd_block_rt(int b, int nblocks, int chunk) {
    const int full = nblocks / (NXCD * chunk) * (NXCD * chunk);
    if (b >= full) return b;
    const int xcd = b % NXCD, pos = b / NXCD;
    return ((pos / chunk) * NXCD + xcd) * chunk + pos % chunk;
}

// Every wave-per-ray kernel is launched as BS threads x ceil(n / (BS / 64)) workgroups, so the
// grid and workgroup sizes follow from the kernel's own argument n: reading gridDim / blockDim
// instead costs a wavefront that lives for ONE ray two more dependent fetches (the hidden
// kernel arguments; blockDim even through the vector memory path) before its first row load:
// k_depth -4 %, k_bp and k_sweep_map -1 % (profiles/r02_exp_wave_startup.txt).
template <int BS = BLOCK, int CHUNK = 0>
__device__ __forceinline__ int ray_of_wave(int n, int &lane, int chunk_rt = 0) {
    constexpr int WPB = BS / WAVE;
    lane = threadIdx.x & (WAVE - 1);
    const int b = chunk_rt > 0 ? xcd_block_rt(blockIdx.x, (n + WPB - 1) / WPB, chunk_rt)
                               : xcd_block<CHUNK>(blockIdx.x, (n + WPB - 1) / WPB);
    // the wave's ray index lives in an SGPR (the compiler cannot see that threadIdx.x >> 6 is
    // wave-uniform): row addresses become scalar base + per-lane 32-bit offset
    const int r = uniform(b * WPB + (int)(threadIdx.x >> 6));
    return r < n ? r : -1;
}

// ------------------------------------------------------------- small kernels
__global__ void k_fill_f32(float *dst, int64_t n, float v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = v;
}
__global__ void k_fill_i32(int32_t *dst, int64_t n, int32_t v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = v;
}
// per-axis centre tables out of the [gx][gy][gz][3] array
__global__ void k_extract_axes(const float *grid, int gx, int gy, int gz, float *axes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < gx)
        axes[i] = grid[((size_t)i * gy * gz) * 3 + 0];
    else if (i < gx + gy)
        axes[i] = grid[((size_t)(i - gx) * gz) * 3 + 1];
    else if (i < gx + gy + gz)
        axes[i] = grid[((size_t)(i - gx - gy)) * 3 + 2];
}
__global__ void k_acc_combine(float *part, int copies, int64_t G, float prior, float *out,
                              int add_prior) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < G;
         i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int c = 0; c < copies; c++) {
            s += part[c * G + i];
            part[c * G + i] = 0.0f;
        }
        out[i] = add_prior ? prior + s : s;
    }
}
// the exact arithmetic shortcut next to the expression it replaces (rn_selftest_arith)
__global__ __launch_bounds__(BLOCK) void k_selftest_arith(int n, const float *__restrict__ a,
                                                          float *out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    out[i] = roundf(a[i]);
    out[(size_t)n + i] = round_half_away(a[i]);
}
// round_quotient_fast next to the division it replaces (rn_selftest_quotient); per LANE here,
// where the sweep falls back per wavefront
__global__ __launch_bounds__(BLOCK) void k_selftest_quotient(int n, const float *__restrict__ x,
                                                             const float *__restrict__ d,
                                                             float *out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    bool sure;
    const float rd = __builtin_amdgcn_rcpf(d[i]);
    const float fast = round_quotient_fast(x[i], rd, sure);
    sure = sure && reciprocal_is_normal(rd);
    out[i] = round_half_away(x[i] / d[i]);
    out[(size_t)n + i] = fast;
    out[2 * (size_t)n + i] = sure ? 1.0f : 0.0f;
}
// the plane sweep's index arithmetic on its own (rn_selftest_feature_offsets): lane = plane, as
// in sweep_coop / sweep_generic
__global__ __launch_bounds__(BLOCK) void k_selftest_offsets(Params p, int n,
                                                            const float *__restrict__ P,
                                                            const float *__restrict__ starts,
                                                            const float *__restrict__ ends,
                                                            int32_t *out) {
    int lane;
    const int r = ray_of_wave(n, lane);
    if (r < 0) return;
    float s[3], e[3];
    for (int i = 0; i < 3; i++) {
        s[i] = starts[3 * r + i];
        e[i] = ends[3 * r + i];
    }
    const float pad_shift = (float)(p.padding - (p.padding - 1) / 2);
    for (int base = 0; base < p.D; base += WAVE) {
        const int k = min(base + lane, p.D - 1);        // (all lanes take part in the ballots)
        float point[3];
        plane_point(s, e, k, p.D, point);
        for (int v = 0; v < p.N; v++) {
            const int generic = feature_offset(p, P + 12 * v, point) / p.F;
            // the cooperative sweep's form (128-byte vectors): byte offset -> vector index
            const int coop = feature_offset_bytes<7>(p, P + 12 * v, point, pad_shift) >> 7;
            if (base + lane < p.D) {
                int32_t *o = out + (((size_t)r * p.N + v) * p.D + k) * 2;
                o[0] = generic;
                o[1] = coop;
            }
        }
    }
}
// the mapping's two exact shortcuts next to the expressions they replace (rn_selftest_mapping)
__global__ __launch_bounds__(BLOCK) void k_selftest_mapping(Params p, int n,
                                                            const float *__restrict__ a,
                                                            const float *__restrict__ b,
                                                            const float *__restrict__ t,
                                                            float *out) {
    extern __shared__ float pos[];
    for (int i = threadIdx.x; i <= p.D; i += BLOCK) pos[i] = 0.0f + i * p.plane_step;
    __syncthreads();
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float y = 1.0f / b[i];
    out[i] = a[i] / b[i];
    out[(size_t)n + i] = markstein_div(a[i], b[i], y);
    out[2 * (size_t)n + i] = markstein_ok(b[i]) && markstein_ok_dividend(a[i]) ? 1.0f : 0.0f;
    const float tc = clampf(t[i], 1e-4f, 1 - 1e-4f);
    out[3 * (size_t)n + i] = (float)plane_index_walk(tc, p.D, p.plane_step);
    out[4 * (size_t)n + i] = (float)plane_index_from_table(pos, tc, p.D);
}
// resident (bricked) accumulator <-> the reference's [gx][gy][gz] array
template <bool TO_GRID>
__global__ void k_acc_regrid(Params p, const float *__restrict__ src, float *dst) {
    const int64_t G = (int64_t)p.gx * p.gy * p.gz;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < G;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % p.gz), y = (int)((i / p.gz) % p.gy), x = (int)(i / ((int64_t)p.gz * p.gy));
        const int64_t b = ((((int64_t)(x >> 2) * p.nby + (y >> 2)) * p.nbz + (z >> 2)) << 6) |
                          ((x & 3) << 4) | ((y & 3) << 2) | (z & 3);
        if (TO_GRID) dst[i] = src[b];
        else dst[b] = src[i];
    }
}
// out[i] = rows[index[i]]: four entries per thread, one 16-byte store -- `out` may be page-locked
// host memory, and whole 256-byte segments per wavefront are what a PCIe write wants
__global__ void k_stitch_rows(int64_t n, const float *__restrict__ rows,
                              const int32_t *__restrict__ index, float *out) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int4 ix = reinterpret_cast<const int4 *>(index)[i];
        reinterpret_cast<float4 *>(out)[i] = make_float4(rows[ix.x], rows[ix.y], rows[ix.z], rows[ix.w]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        out[i] = rows[index[i]];
    }
}
__global__ void k_add_scalar(float *a, int64_t n, float v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        a[i] = v + a[i];
}

#include "raynet_prepare.inl"
#include "raynet_mrf.inl"

}  // namespace

// =============================================================== host side
struct rn_ctx {
    rn_config cfg;
    Params p;
    float *axes;          // device, gx+gy+gz
    bool have_axes;
    int scatter_mode;     // rn_options: -1 by row layout (default), 0 slab, 2 LDS box
    int generic_sweep;    // rn_options: reference-order plane sweep even for F = 32
    int sweep_rpw;        // rn_options.sweep_rays_per_wave: 0 by D, 1 one ray per wavefront
    // LDS-box scatter: level in use (launch_bp), {chunks, overflowed chunks} of the previous
    // launches on the device and its pinned host mirror
    int box_level, box_level0;
    bool box_pin;         // RAYNET_HIP_BOX_PIN: stay at the starting level (A/B runs)
    // device counters {chunks, overflowed chunks}, cumulative over launches; their pinned host
    // mirror (an asynchronous copy, it may lag a launch); what the launcher had seen of them at
    // its previous look; the difference = the launches in between (rn_scatter_state), and how
    // many more launches copy the counters out: the tile shape settles within the first
    // launches after a reset, and two 8-byte operations behind every scatter are two more
    // dependent items on a stream whose kernels take 70 us each on an eight-rank shard
    unsigned *box_stats, *box_stats_host;
    unsigned box_seen[2], box_obs[2];
    int box_probe;
    bool box_probe_used;  // some scatter has run since the context was created
    bool box_rebase;      // after a reset: the next counters that arrive are a baseline, not an observation
    // occupancy_to_ray(prior, 0) as the device evaluates it, for the prior it was last asked for
    float first_prior, first_occ;
    bool have_first_occ;
    float *scalar_dev;        // 4 bytes of device scratch owned by the context
    hipEvent_t ev0, ev1;
    // second stream of the resident-scene launchers (RAYNET_HIP_OVERLAP=0 / 1, default: by
    // the scatter's tile level): the accumulator scatter of one half of a launch's rows runs
    // next to the BP sweep of the other half, the traversal of half of the images next to the
    // plane sweep of the rest.  Measured (profiles/r02_exp_overlap.txt): config 2 8.72 ->
    // 8.84 ms/step (either kernel alone already keeps the VALUs of every CU busy), config 4
    // 44.3 -> 42.5 (its scatter waits on L2 atomics at 3.9 hits per voxel) -- and config 4 is
    // where the adaptive scatter has stepped to its second tile shape, so that is the switch
    // slab boxes (rn_scene_bind_slab_boxes): table, the list buffer it describes, and the row
    // range rn_scene_prepare_all last filled
    const int32_t *sb_vox;
    int64_t sb_rows, sb_valid_lo, sb_valid_hi;
    int2 *sb_boxes;
    // work list of the box scatter (rn_scene_bind_scatter_items): the row range and tile level it
    // was built for
    const int32_t *sc_vox, *sc_items;
    int64_t sc_rows;
    int sc_level, sc_count;
    int overlap;          // 0 off, 1 on, 2 (default) when the scatter runs at tile level >= 1
    hipStream_t aux;
    hipEvent_t ev_fork, ev_join;
    // per-launch profiling (rn_prof_begin / rn_prof_end)
    bool prof_on;
    uint32_t prof_mask;       // which rn_kernel_id families are bracketed (rn_prof_select)
    int prof_cap, prof_n;
    hipEvent_t *prof_ev;      // 2 * prof_cap
    int32_t *prof_id, *prof_rays;
    // a failed opt-in to more than 64 KB of dynamic LDS (launch_sweep_t), reported by the launch check
    hipError_t lds_optin_error;
    size_t lds_optin_bytes;
    char err[512];
};

// Brackets one kernel launch with two events on its stream when profiling is on.
struct ProfScope {
    rn_ctx *c;
    hipStream_t st;
    int slot;
    ProfScope(rn_ctx *ctx, int id, int n_rays, hipStream_t s) : c(ctx), st(s), slot(-1) {
        if (c->prof_on && ((c->prof_mask >> id) & 1u) && c->prof_n < c->prof_cap) {
            slot = c->prof_n++;
            c->prof_id[slot] = id;
            c->prof_rays[slot] = n_rays;
            (void)hipEventRecord(c->prof_ev[2 * slot], st);
        }
    }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(c->prof_ev[2 * slot + 1], st);
    }
};

namespace {

int fail(rn_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define RN_HIP(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(ctx, RN_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define RN_LAUNCH_CHECK(ctx)                                                           \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if ((ctx)->lds_optin_error != hipSuccess) {                                    \
            const hipError_t o_ = (ctx)->lds_optin_error;                              \
            (ctx)->lds_optin_error = hipSuccess;                                       \
            return fail(ctx, RN_ERR_HIP, "the plane sweep's opt-in to %zu bytes of LDS "  \
                        "(hipFuncSetAttribute) failed: %s; its launch: %s",            \
                        (ctx)->lds_optin_bytes, hipGetErrorString(o_), hipGetErrorString(e_)); \
        }                                                                              \
        if (e_ != hipSuccess)                                                          \
            return fail(ctx, RN_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int ray_blocks(int n) { return (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK; }
inline int ray_blocks_mrf(int n) { return (n + RAY_BLOCK / WAVE - 1) / (RAY_BLOCK / WAVE); }
inline int thread_blocks(int n) { return (n + BLOCK - 1) / BLOCK; }
inline int fill_blocks(int64_t n) {
    int64_t b = (n + BLOCK - 1) / BLOCK;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
inline size_t sweep_lds(const Params &p, int rows = 1) {
    return sizeof(float) * ((size_t)((p.gx + p.gy + p.gz + 3) & ~3) + (size_t)((p.D + 4) & ~3) +
                            (size_t)SWEEP_WAVES * (p.D + (size_t)rows * p.M));
}
inline int sweep_blocks(int n) { return (n + SWEEP_WAVES - 1) / SWEEP_WAVES; }
// rays a wavefront of the cooperative sweep takes (k_sweep_map_packed for 2 / 4)
inline int sweep_rays_per_wave(const rn_ctx *ctx) {
    if (ctx->sweep_rpw == 1) return 1;
    return ctx->p.D <= 16 ? 4 : ctx->p.D <= 32 ? 2 : 1;
}

// floats of one resident (bricked) accumulator: every axis padded to a multiple of 4
inline int64_t acc_floats(const rn_ctx *ctx) {
    return (int64_t)((ctx->p.gx + 3) / 4) * ctx->p.nby * ctx->p.nbz * 64;
}

FeatureViews stacked_views(const Params &p, const float *features) {
    FeatureViews fv;
    const size_t dim = (size_t)p.Hf * p.Wf * p.F;
    for (int v = 0; v < MAX_VIEWS; v++) fv.v[v] = v < p.N ? features + dim * v : nullptr;
    return fv;
}

inline void lds_opt_in(rn_ctx *ctx, const void *kernel, size_t lds, size_t &granted) {
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        granted = lds;
    } else {
        (void)hipGetLastError();
        ctx->lds_optin_error = e;
        ctx->lds_optin_bytes = lds;
    }
}

struct SweepArgs {
    int n;
    const int32_t *ray_idxs;
    FeatureViews fv;
    const float *P, *P_inv, *cc, *starts, *ends, *S_in;
    const int32_t *vox, *rvc;
    float *S_planes, *S_voxel, *depth_from_planes, *points;
    const int32_t *order = nullptr;
    // scene-wide launch (rn_scene_prepare_all): one grid row per reference image
    const float *const *fv_table = nullptr;
    int cam_stride = 0;
    int64_t rows_per_image = 0;
    int n_images = 1;
    const float *seg = nullptr;     // [rows][8]: ray segments written by k_traverse
    float *msgs_out = nullptr;      // MAPMODE 3: BP iteration 0's messages
    float prior = 0.0f;             // MAPMODE 3: occupancy_to_ray(prior, 0), see first_occupancy()
    float *zero = nullptr;          // MAPMODE 3: cleared on the side (rn_acc_size floats)
    int xcd_chunk = 0;              // workgroups per XCD group (0: RN_XCD_CHUNK_SWEEP)
};

template <int SIM, int NV, int LPS, int MAPMODE, bool PACKED>
void launch_sweep_t(rn_ctx *ctx, const SweepArgs &a, hipStream_t st) {
    ProfScope prof(ctx, RN_K_SWEEP_MAP, a.n * a.n_images, st);
    const size_t lds = sweep_lds(ctx->p, MAPMODE == 3 ? 3 : 1);
    // gfx950 has 160 KB per CU; beyond 64 KB a kernel has to say so -- once per instantiation and
    // size (one process drives one GPU), and a refusal is kept for the launch check to report
    static size_t granted = 64 * 1024;
    if (lds > granted)
        lds_opt_in(ctx, (const void *)k_sweep_map<SIM, NV, LPS, MAPMODE, PACKED>, lds, granted);
    hipLaunchKernelGGL((k_sweep_map<SIM, NV, LPS, MAPMODE, PACKED>),
                       dim3(sweep_blocks(a.n), a.n_images), dim3(SWEEP_BLOCK), lds, st,
                       ctx->p, a.n, a.ray_idxs, a.fv, a.P, a.P_inv, a.cc, a.starts, a.ends, a.S_in,
                       ctx->axes, a.vox, a.rvc, a.S_planes, a.S_voxel, a.depth_from_planes,
                       a.points, a.order, a.fv_table, a.cam_stride, a.rows_per_image, a.seg,
                       a.msgs_out, a.prior, reinterpret_cast<float4 *>(a.zero),
                       a.zero ? (int)(acc_floats(ctx) / 4) : 0, a.xcd_chunk);
}

// k_sweep_map_packed: RPW rays per wavefront (D <= 64 / RPW)
inline size_t sweep_lds_packed(const Params &p, int rows, int rpw) {
    return sizeof(float) * ((size_t)((p.gx + p.gy + p.gz + 3) & ~3) + (size_t)((p.D + 4) & ~3) +
                            (size_t)SWEEP_WAVES * ((size_t)rpw * p.D + (size_t)rows * p.M));
}
template <int NV, int LPS, int MAPMODE, bool PACKED, int RPW>
void launch_sweep_packed_t(rn_ctx *ctx, const SweepArgs &a, hipStream_t st) {
    ProfScope prof(ctx, RN_K_SWEEP_MAP, a.n * a.n_images, st);
    const size_t lds = sweep_lds_packed(ctx->p, MAPMODE == 3 ? 3 : 1, RPW);
    static size_t granted = 64 * 1024;
    if (lds > granted)
        lds_opt_in(ctx, (const void *)k_sweep_map_packed<NV, LPS, MAPMODE, PACKED, RPW>, lds, granted);
    const int nwaves = (a.n + RPW - 1) / RPW;
    hipLaunchKernelGGL((k_sweep_map_packed<NV, LPS, MAPMODE, PACKED, RPW>),
                       dim3(sweep_blocks(nwaves), a.n_images), dim3(SWEEP_BLOCK), lds, st,
                       ctx->p, a.n, a.ray_idxs, a.fv, a.P, a.P_inv, a.cc, a.starts, a.ends,
                       ctx->axes, a.vox, a.rvc, a.S_planes, a.S_voxel, a.depth_from_planes,
                       a.points, a.order, a.fv_table, a.cam_stride, a.rows_per_image, a.seg,
                       a.msgs_out, a.prior, reinterpret_cast<float4 *>(a.zero),
                       a.zero ? (int)(acc_floats(ctx) / 4) : 0,
                       a.xcd_chunk > 0 ? (a.xcd_chunk + RPW - 1) / RPW : 0);
}

// pick the plane-sweep flavour: cooperative for F=32 and 2..9 views (two / four rays per
// wavefront for D <= 32 / 16 unless rn_options.sweep_rays_per_wave says 1), generic otherwise
template <int MAPMODE, bool PACKED>
void launch_sweep(rn_ctx *ctx, const SweepArgs &a, bool have_features, hipStream_t st) {
    const Params &p = ctx->p;
    if (!have_features) {
        launch_sweep_t<0, 1, 8, MAPMODE, PACKED>(ctx, a, st);
        return;
    }
    if (p.F == 32 && !ctx->generic_sweep) {
        const int rpw = sweep_rays_per_wave(ctx);
        switch (p.N) {
#define RN_CASE(NV_)                                                  \
    case NV_:                                                         \
        if (rpw == 4)                                                 \
            launch_sweep_packed_t<NV_, 8 / SWEEP_V4, MAPMODE, PACKED, 4>(ctx, a, st);   \
        else if (rpw == 2)                                            \
            launch_sweep_packed_t<NV_, 8 / SWEEP_V4, MAPMODE, PACKED, 2>(ctx, a, st);   \
        else                                                          \
            launch_sweep_t<2, NV_, 8 / SWEEP_V4, MAPMODE, PACKED>(ctx, a, st);       \
        return;
            RN_CASE(2) RN_CASE(3) RN_CASE(4) RN_CASE(5) RN_CASE(6) RN_CASE(7) RN_CASE(8) RN_CASE(9)
#undef RN_CASE
            default: break;
        }
    }
    launch_sweep_t<1, 1, 8, MAPMODE, PACKED>(ctx, a, st);
}

// the slab-box rows that describe `vox` (a pointer into the bound list buffer), or null
inline int2 *slab_boxes_for(const rn_ctx *ctx, const int32_t *vox, int64_t n, bool need_valid) {
    if (!ctx->sb_boxes || !vox || vox < ctx->sb_vox) return nullptr;
    const int64_t off = vox - ctx->sb_vox;
    if (off % ctx->p.M) return nullptr;
    const int64_t row0 = off / ctx->p.M;
    if (row0 % WAVE || row0 + n > ctx->sb_rows) return nullptr;
    if (need_valid && (row0 < ctx->sb_valid_lo || row0 + n > ctx->sb_valid_hi)) return nullptr;
    return ctx->sb_boxes + (row0 / WAVE) * slab_box_count(ctx->p.M);
}

// workgroups per box-scatter tile (grid.y): enough of them for ~16 per CU
inline int box_split(int n, int tile_rays) {
    const int tiles = (n + tile_rays - 1) / tile_rays;
#define RN_BOX_SPLIT_TARGET 4096
#define RN_BOX_SPLIT_MAX 4
    // (RAYNET_HIP_BOX_SPLIT="target,max": A/B override, read once)
    static int target = 0, most = 0;
    if (!target) {
        target = RN_BOX_SPLIT_TARGET;
        most = RN_BOX_SPLIT_MAX;
        if (const char *e = getenv("RAYNET_HIP_BOX_SPLIT")) (void)sscanf(e, "%d,%d", &target, &most);
        if (target < 1) target = 1;
        if (most < 1) most = 1;
    }
    return max(1, min(most, target / max(tiles, 1)));
}

// One BP sweep: k_bp (messages) + the accumulator scatter that fits the row layout.
// how a sweep reads its accumulator and what it clears on the side (the plan path, rn_scene_run)
struct AccMode {
    bool uniform = false;      // every voxel holds acc_in[0]
    bool biased = false;       // acc_in holds sums only: the prior `bias` is added at the gather
    float bias = 0.0f;
    float *zero = nullptr;     // cleared by the FIRST k_bp launch of this call (rn_acc_size floats)
};

template <bool PACKED, bool CLIP_IN>
void launch_bp_kernel(rn_ctx *ctx, int n, const float *Sv, const int32_t *vox, const int32_t *rvc,
                      const float *acc_in, const float *msgs_in, float *msgs_out, hipStream_t st,
                      const AccMode &am, bool clear) {
    const int nch = (ctx->p.M + WAVE - 1) / WAVE;
    ProfScope prof(ctx, RN_K_BP, n, st);
    float4 *zero = clear ? reinterpret_cast<float4 *>(am.zero) : nullptr;
    const int zero4 = zero ? (int)(acc_floats(ctx) / 4) : 0;
#define RN_BP_(NCH_, STEADY_)                                                                   \
    hipLaunchKernelGGL((k_bp<NCH_, PACKED, CLIP_IN, STEADY_>), dim3(ray_blocks_mrf(n)),             \
                       dim3(RAY_BLOCK), 0, st, ctx->p, n, Sv, vox, rvc, acc_in, msgs_in,         \
                       msgs_out, am.uniform ? 1 : 0, am.bias, am.biased ? 1 : 0, zero, zero4)
    // the plan path's iterations after the first: everything the kernel would test per chunk
    // is known here (k_bp's STEADY)
    const bool steady = PACKED && !CLIP_IN && msgs_in && !am.uniform && am.biased;
#define RN_BP(NCH_) do { if (steady) RN_BP_(NCH_, true); else RN_BP_(NCH_, false); } while (0)
    if (nch <= 2) RN_BP(2);
    else if (nch <= 4) RN_BP(4);
    else if (nch <= 6) RN_BP(6);
    else if (nch <= 8) RN_BP(8);
    else if (nch <= 12) RN_BP(12);
    else RN_BP(16);
#undef RN_BP
#undef RN_BP_
}

// the scatter kernel for `level` (see launch_bp) over rows [0, n)
template <bool PACKED>
void launch_scatter_kernel(rn_ctx *ctx, int n, const float *msgs, const int32_t *vox,
                           const int32_t *rvc, void *acc_out, hipStream_t st, int level,
                           bool fixed) {
    ProfScope prof(ctx, RN_K_SCATTER, n, st);
    // a work list bound for exactly these rows and this tile shape (else: tiles x box_split)
    const int32_t *items = PACKED && ctx->sc_items && vox == ctx->sc_vox && n == ctx->sc_rows &&
                           level == ctx->sc_level ? ctx->sc_items : nullptr;
#define RN_BOX(RAYS, STEPS, FIXED_, CAP)                                                          \
    hipLaunchKernelGGL((k_scatter_box<PACKED, RAYS, STEPS, FIXED_>),                              \
                       items ? dim3(ctx->sc_count, 1) : dim3((n + RAYS - 1) / RAYS, box_split(n, RAYS)), \
                       dim3(BLOCK), (CAP) * sizeof(double), st, ctx->p, n, msgs, vox, rvc, acc_out, \
                       ctx->box_stats, CAP,                                                       \
                       (const int2 *)(PACKED ? slab_boxes_for(ctx, vox, n, true) : nullptr), items)
#define RN_BOX0_CAP 4096
    if (level == 0) {
        if (fixed) RN_BOX(128, 32, true, RN_BOX0_CAP); else RN_BOX(128, 32, false, RN_BOX0_CAP);
    } else if (level == 1) {
        if (fixed) RN_BOX(256, 16, true, 6144); else RN_BOX(256, 16, false, 6144);
    } else if (fixed) {
        hipLaunchKernelGGL((k_scatter_direct_fixed<PACKED>), dim3(ray_blocks(n)), dim3(BLOCK), 0, st,
                           ctx->p, n, msgs, vox, rvc, static_cast<unsigned long long *>(acc_out));
    } else {
        hipLaunchKernelGGL((k_scatter_slab<PACKED>),
                           dim3(((n + WAVE - 1) / WAVE) *
                                ((ctx->p.M + SLAB_STEPS - 1) / SLAB_STEPS)),
                           dim3(WAVE), 0, st, ctx->p, n, msgs, vox, rvc,
                           static_cast<float *>(acc_out));
    }
#undef RN_BOX
}

template <bool PACKED, bool CLIP_IN>
int launch_bp(rn_ctx *ctx, int n, const float *Sv, const int32_t *vox, const int32_t *rvc,
              const float *acc_in, const float *msgs_in, void *acc_out, float *msgs_out,
              hipStream_t st, bool patch_rows = false, bool fixed = false,
              const AccMode &am = AccMode(), bool skip_bp = false) {
    // Patch-ordered rows start with the LDS-box scatter on 128-ray x 32-step tiles.  The
    // kernel counts the chunks whose bounding box did not fit its LDS budget; the count of
    // the previous launches is copied out asynchronously (it may lag a launch) and when too
    // many overflowed the launcher steps down: more LDS, then narrower chunks, then -- pixel
    // spacing above the voxel size, nothing to sum per voxel anyway -- the slab scatter.
    // levels: 0 = 128 x 32 tiles with a 4096-voxel box (32 KB), 1 = 256 x 16 tiles with 6144
    // voxels (48 KB; measured best of 4096..8192 on the 256^3 grid of config 4), 2 = slab
    // scatter.  0 -> 1 above 2 % overflowed chunks, 1 -> 2 only above 25 % (the box kernel's
    // quarter-chunk fallback still beats the slab scatter below that).
    constexpr int LAST = 2;
    int level = ctx->scatter_mode == 0 ? LAST : (ctx->scatter_mode == 2 || patch_rows) ? 0 : LAST;
    if (level == 0) {
        const unsigned c0 = ctx->box_stats_host[0], c1 = ctx->box_stats_host[1];
        if (c0 != ctx->box_seen[0] && ctx->box_rebase) {
            // the counters are cumulative and were last looked at before the reset: what has
            // arrived mixes launches of the previous scene / tile shape in -- a baseline only
            ctx->box_seen[0] = c0;
            ctx->box_seen[1] = c1;
            ctx->box_rebase = false;
        } else if (c0 != ctx->box_seen[0]) {   // counters of more launches have arrived
            ctx->box_obs[0] = c0 - ctx->box_seen[0];
            ctx->box_obs[1] = c1 - ctx->box_seen[1];
            ctx->box_seen[0] = c0;
            ctx->box_seen[1] = c1;
            const unsigned per = ctx->box_level < LAST - 1 ? 50u : 4u;
            if (!ctx->box_pin && ctx->box_level < LAST && ctx->box_obs[1] * per > ctx->box_obs[0]) {
                ctx->box_level++;
                ctx->box_probe = BOX_PROBE_LAUNCHES;      // look at the new shape as well
            }
        }
        level = ctx->box_level;
    }
    // k_bp is bound by VALU issue and its dependent row / gather round trips, the box scatter
    // by LDS atomics and barriers: with the rows in two halves the scatter of the first half
    // runs (on the context's second stream) while the second half's messages are computed.
    const size_t M = (size_t)ctx->p.M, VW = PACKED ? 1 : 3;
    const bool split = ctx->overlap == 1 || (ctx->overlap == 2 && level >= 1 && level < LAST);
    const int nA = split && PACKED && n >= 65536 && !skip_bp ? (n / 2 + 255) / 256 * 256 : n;
    if (!skip_bp) {     // (else: the plane sweep wrote the messages and cleared am.zero)
        launch_bp_kernel<PACKED, CLIP_IN>(ctx, nA, Sv, vox, rvc, acc_in, msgs_in, msgs_out, st, am, true);
        RN_LAUNCH_CHECK(ctx);
    }
    if (nA < n) {
        RN_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
        launch_bp_kernel<PACKED, CLIP_IN>(ctx, n - nA, Sv + nA * M, vox + nA * M * VW, rvc + nA, acc_in,
                                          msgs_in ? msgs_in + nA * M : nullptr, msgs_out + nA * M, st,
                                          am, false);
        RN_LAUNCH_CHECK(ctx);
        RN_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        launch_scatter_kernel<PACKED>(ctx, nA, msgs_out, vox, rvc, acc_out, ctx->aux, level, fixed);
        RN_LAUNCH_CHECK(ctx);
        RN_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
        launch_scatter_kernel<PACKED>(ctx, n - nA, msgs_out + nA * M, vox + nA * M * VW, rvc + nA,
                                      acc_out, st, level, fixed);
        RN_LAUNCH_CHECK(ctx);
        RN_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    } else {
        launch_scatter_kernel<PACKED>(ctx, n, msgs_out, vox, rvc, acc_out, st, level, fixed);
        RN_LAUNCH_CHECK(ctx);
    }
    ctx->box_probe_used = true;
    if (ctx->box_probe > 0) {
        ctx->box_probe--;
        if (level < LAST)
            (void)hipMemcpyAsync(ctx->box_stats_host, ctx->box_stats, 2 * sizeof(unsigned),
                                 hipMemcpyDeviceToHost, st);
    }
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

template <bool PACKED, bool CLIP_IN>
int launch_depth(rn_ctx *ctx, int n, const float *Sv, const int32_t *vox, const int32_t *rvc,
                 const float *acc, const float *msgs, const float *cc, float *S_new,
                 float *depth_map, hipStream_t st, int rays_per_center = 0,
                 const AccMode &am = AccMode(), int cc_stride = 4,
                 const DepthDest &dest = DepthDest()) {
    const int nch = (ctx->p.M + WAVE - 1) / WAVE;
    ProfScope prof(ctx, RN_K_DEPTH, n, st);
#define RN_DE_(NCH_, STEADY_)                                                                   \
    hipLaunchKernelGGL((k_depth<NCH_, PACKED, CLIP_IN, STEADY_>), dim3(ray_blocks_mrf(n)),          \
                       dim3(RAY_BLOCK), 0, st, ctx->p, n, Sv, vox, rvc, acc, msgs, ctx->axes, cc, \
                       S_new, depth_map, rays_per_center, am.bias, am.biased ? 1 : 0, cc_stride, dest)
    // (k_depth's STEADY form -- its flags known at compile time, as k_bp's: slower with plain
    // row loads, 0.746 -> 0.772 ms per step, faster with the non-temporal ones, 0.717 -> 0.699;
    // -DRN_DEPTH_NO_STEADY: the generic kernel)
    const bool steady = PACKED && !CLIP_IN && msgs && !S_new && depth_map && am.biased;
#define RN_DE(NCH_) do { if (steady) RN_DE_(NCH_, true); else RN_DE_(NCH_, false); } while (0)
    if (nch <= 2) RN_DE(2);
    else if (nch <= 4) RN_DE(4);
    else if (nch <= 6) RN_DE(6);
    else if (nch <= 8) RN_DE(8);
    else if (nch <= 12) RN_DE(12);
    else RN_DE(16);
#undef RN_DE
#undef RN_DE_
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

inline AccMode uniform_mode(bool uniform) {
    AccMode am;
    am.uniform = uniform;
    return am;
}

int need_axes(rn_ctx *ctx) {
    if (!ctx->have_axes)
        return fail(ctx, RN_ERR_STATE, "rn_set_voxel_grid must be called before this entry point");
    return RN_OK;
}

}  // namespace

extern "C" {

// The version string names every compile-time knob the library was built with (all of them
// result-neutral A/B switches or the exact-arithmetic variants of DESIGN.md section 10a; the
// timing-only ablations of earlier rounds live in tools/experiments/ as a patch, not here) and
// whatever extra flags the build script passed (RN_BUILD_EXTRA): tests/test_abi.py holds the
// shipped library to "knobs: none".
#ifndef RN_BUILD_EXTRA
#define RN_BUILD_EXTRA ""
#endif
const char *rn_version(void) {
    static const char v[] = "raynet_hip 0.4 (gfx950) knobs:"
#ifdef RN_EXACT_OCC_EXP
        " RN_EXACT_OCC_EXP"
#endif
#ifdef RN_EXACT_SOFTMAX_EXP
        " RN_EXACT_SOFTMAX_EXP"
#endif
#ifdef RN_EXACT_BP_MATH
        " RN_EXACT_BP_MATH"
#endif
        " | extra:" RN_BUILD_EXTRA;
    return v;
}

const char *rn_last_error(const rn_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int rn_create(const rn_config *cfg, rn_ctx **out) {
    if (!cfg || !out) return RN_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RN_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return RN_ERR_NO_DEVICE;
    if (cfg->M < 1 || cfg->M > 1024 || cfg->D < 2 || cfg->D > 4096 || cfg->N < 2 ||
        cfg->N > MAX_VIEWS || cfg->F < 1 || cfg->H < 1 || cfg->W < 1 || cfg->padding < 0)
        return RN_ERR_INVALID;
    for (int i = 0; i < 3; i++)
        if (cfg->grid[i] < 1 || cfg->grid[i] > 1024 || !(cfg->bbox[3 + i] > cfg->bbox[i]))
            return RN_ERR_INVALID;
    if ((int64_t)((cfg->grid[0] + 3) / 4) * ((cfg->grid[1] + 3) / 4) * ((cfg->grid[2] + 3) / 4) >=
        ((int64_t)1 << 24))
        return RN_ERR_INVALID;        // accumulator indices are 32-bit
    if ((int64_t)(cfg->H + cfg->padding + 1) * (cfg->W + cfg->padding + 1) * cfg->F >=
        ((int64_t)1 << 29))
        return RN_ERR_INVALID;        // feature vectors are addressed with 32-bit byte offsets
    if (hipSetDevice(cfg->device) != hipSuccess) return RN_ERR_HIP;
    rn_ctx *ctx = new rn_ctx();
    memset(ctx, 0, sizeof(*ctx));
    ctx->cfg = *cfg;
    Params &p = ctx->p;
    p.M = cfg->M; p.D = cfg->D; p.N = cfg->N; p.F = cfg->F;
    p.H = cfg->H; p.W = cfg->W; p.padding = cfg->padding;
    p.gx = cfg->grid[0]; p.gy = cfg->grid[1]; p.gz = cfg->grid[2];
    p.nby = (p.gy + 3) / 4; p.nbz = (p.gz + 3) / 4;
    p.plane_step = (1.0f - 0.0f) / (p.D - 1);
    p.Hf = cfg->H + cfg->padding + 1;
    p.Wf = cfg->W + cfg->padding + 1;
    for (int i = 0; i < 6; i++) p.bbox[i] = cfg->bbox[i];
    const char *sm = getenv("RAYNET_HIP_SCATTER_MODE");
    ctx->scatter_mode = sm ? atoi(sm) : -1;
    const char *bs = getenv("RAYNET_HIP_BOX_LEVEL");      // A/B knob: start at this tile shape
    ctx->box_level = ctx->box_level0 = bs ? max(0, min(2, atoi(bs))) : 0;
    ctx->box_pin = getenv("RAYNET_HIP_BOX_PIN") != nullptr;
    const char *ov = getenv("RAYNET_HIP_OVERLAP");
    ctx->overlap = ov ? (atoi(ov) != 0 ? 1 : 0) : 2;
    ctx->generic_sweep = getenv("RAYNET_HIP_GENERIC_SWEEP") != nullptr;
    const char *rw = getenv("RAYNET_HIP_SWEEP_RAYS_PER_WAVE");
    ctx->sweep_rpw = rw && atoi(rw) == 1 ? 1 : 0;
    ctx->prof_mask = ~0u;
    if (sweep_lds(p) > 160 * 1024) {      // before anything is allocated
        delete ctx;
        return RN_ERR_INVALID;
    }
    if (hipMalloc(&ctx->axes, sizeof(float) * (p.gx + p.gy + p.gz)) != hipSuccess ||
        hipMalloc(&ctx->box_stats, 2 * sizeof(unsigned)) != hipSuccess ||
        hipMalloc(&ctx->scalar_dev, sizeof(float)) != hipSuccess ||
        hipHostMalloc(&ctx->box_stats_host, 2 * sizeof(unsigned)) != hipSuccess ||
        hipMemset(ctx->box_stats, 0, 2 * sizeof(unsigned)) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        rn_destroy(ctx);                  // frees whatever was created (all null-checked)
        return RN_ERR_HIP;
    }
    ctx->box_stats_host[0] = ctx->box_stats_host[1] = 0;
    ctx->box_probe = BOX_PROBE_LAUNCHES;
    *out = ctx;
    return RN_OK;
}

void rn_destroy(rn_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->cfg.device);
    if (ctx->axes) hipFree(ctx->axes);
    if (ctx->box_stats) hipFree(ctx->box_stats);
    if (ctx->scalar_dev) hipFree(ctx->scalar_dev);
    if (ctx->box_stats_host) hipHostFree(ctx->box_stats_host);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->aux) hipStreamDestroy(ctx->aux);
    for (int i = 0; i < 2 * ctx->prof_cap; i++) hipEventDestroy(ctx->prof_ev[i]);
    delete[] ctx->prof_ev;
    delete[] ctx->prof_id;
    delete[] ctx->prof_rays;
    delete ctx;
}

int rn_get_options(const rn_ctx *ctx, rn_options *out) {
    if (!ctx || !out) return RN_ERR_INVALID;
    out->scatter_mode = ctx->scatter_mode;
    out->box_level = ctx->box_level0;
    out->box_pin = ctx->box_pin ? 1 : 0;
    out->overlap = ctx->overlap;
    out->generic_sweep = ctx->generic_sweep;
    out->sweep_rays_per_wave = ctx->sweep_rpw;
    return RN_OK;
}

int rn_set_options(rn_ctx *ctx, const rn_options *opt) {
    if (!ctx || !opt) return RN_ERR_INVALID;
    if ((opt->scatter_mode != -1 && opt->scatter_mode != 0 && opt->scatter_mode != 2) ||
        opt->box_level < 0 || opt->box_level > 2 || opt->overlap < 0 || opt->overlap > 2 ||
        (opt->sweep_rays_per_wave != 0 && opt->sweep_rays_per_wave != 1))
        return fail(ctx, RN_ERR_INVALID, "rn_set_options: value out of range");
    ctx->scatter_mode = opt->scatter_mode;
    ctx->box_level = ctx->box_level0 = opt->box_level;
    ctx->box_pin = opt->box_pin != 0;
    ctx->overlap = opt->overlap;
    ctx->generic_sweep = opt->generic_sweep != 0;
    ctx->sweep_rpw = opt->sweep_rays_per_wave;
    ctx->box_obs[0] = ctx->box_obs[1] = 0;
    ctx->box_probe = BOX_PROBE_LAUNCHES;
    ctx->box_rebase = ctx->box_probe_used;
    return RN_OK;
}

int rn_set_voxel_grid(rn_ctx *ctx, const float *voxel_grid, void *stream) {
    if (!ctx || !voxel_grid) return fail(ctx, RN_ERR_INVALID, "null argument");
    const Params &p = ctx->p;
    const int tot = p.gx + p.gy + p.gz;
    hipLaunchKernelGGL(k_extract_axes, dim3(thread_blocks(tot)), dim3(BLOCK), 0, S(stream),
                       voxel_grid, p.gx, p.gy, p.gz, ctx->axes);
    RN_LAUNCH_CHECK(ctx);
    ctx->have_axes = true;
    return RN_OK;
}

int rn_fill_f32(rn_ctx *ctx, float *dst, int64_t count, float value, void *stream) {
    if (!ctx || (!dst && count)) return fail(ctx, RN_ERR_INVALID, "null argument");
    if (count <= 0) return RN_OK;
    hipLaunchKernelGGL(k_fill_f32, dim3(fill_blocks(count)), dim3(BLOCK), 0, S(stream), dst, count,
                       value);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_fill_i32(rn_ctx *ctx, int32_t *dst, int64_t count, int32_t value, void *stream) {
    if (!ctx || (!dst && count)) return fail(ctx, RN_ERR_INVALID, "null argument");
    if (count <= 0) return RN_OK;
    hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks(count)), dim3(BLOCK), 0, S(stream), dst, count,
                       value);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_sample_rays(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *P_inv,
                   const float *camera_center, float *ray_start, float *ray_end, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !P_inv || !camera_center || !ray_start || !ray_end)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    hipLaunchKernelGGL(k_sample_rays, dim3(thread_blocks(n)), dim3(BLOCK), 0, S(stream), ctx->p, n,
                       ray_idxs, P_inv, camera_center, ray_start, ray_end);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_sample_points(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *P_inv,
                     const float *camera_center, float *points, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !P_inv || !camera_center || !points)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    hipLaunchKernelGGL(k_sample_points, dim3(ray_blocks(n)), dim3(BLOCK), 0, S(stream), ctx->p, n,
                       ray_idxs, P_inv, camera_center, points);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_compute_similarities(rn_ctx *ctx, int32_t n, const float *features, const float *P,
                            const float *ray_start, const float *ray_end, float *Sp,
                            void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !features || !P || !ray_start || !ray_end || !Sp)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    SweepArgs a{n, nullptr, stacked_views(ctx->p, features), P, nullptr, nullptr, ray_start,
                ray_end, nullptr, nullptr, nullptr, Sp, nullptr, nullptr, nullptr};
    launch_sweep<0, false>(ctx, a, true, S(stream));
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_voxel_traversal(rn_ctx *ctx, int32_t n, const float *ray_start, const float *ray_end,
                       int32_t *rvi, int32_t *rvc, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_start || !ray_end || !rvi || !rvc)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    {
        ProfScope prof(ctx, RN_K_TRAVERSE, n, S(stream));
        hipLaunchKernelGGL((k_traverse<false>), dim3((n + WAVE - 1) / WAVE), dim3(WAVE), 0, S(stream),
                           ctx->p, n, (const int32_t *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, ray_start, ray_end, rvi, rvc, 0, (int64_t)0,
                           (float *)nullptr);
    }
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_planes_to_voxels(rn_ctx *ctx, int32_t n, const int32_t *rvi, const int32_t *rvc,
                        const float *ray_start, const float *ray_end, const float *Sp,
                        float *S_new, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !rvi || !rvc || !ray_start || !ray_end || !Sp || !S_new)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    SweepArgs a{n, nullptr, FeatureViews{}, nullptr, nullptr, nullptr, ray_start, ray_end, Sp,
                rvi, rvc, nullptr, S_new, nullptr, nullptr};
    launch_sweep<1, false>(ctx, a, false, S(stream));
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_bp_sweep(rn_ctx *ctx, int32_t n, const float *Sv, const int32_t *rvi, const int32_t *rvc,
                const float *acc_in, const float *msgs_in, float *acc_out, float *msgs_out,
                void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !Sv || !rvi || !rvc || !acc_in || !acc_out || !msgs_out)
        return fail(ctx, RN_ERR_INVALID, "bad argument");   /* msgs_in == NULL: all-zero messages */
    if (n == 0) return RN_OK;
    return launch_bp<false, true>(ctx, n, Sv, rvi, rvc, acc_in, msgs_in, acc_out, msgs_out,
                                  S(stream));
}

int rn_depth_estimation(rn_ctx *ctx, int32_t n, const float *Sv, const int32_t *rvi,
                        const int32_t *rvc, const float *acc, const float *msgs, float *S_new,
                        void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !Sv || !rvi || !rvc || !acc || !msgs || !S_new)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    return launch_depth<false, true>(ctx, n, Sv, rvi, rvc, acc, msgs, nullptr, S_new, nullptr,
                                     S(stream));
}

int rn_mvcnn_similarities(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                          const float *P, const float *P_inv, const float *camera_center,
                          float *Sp, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !features || !P || !P_inv || !camera_center || !Sp)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    SweepArgs a{n, ray_idxs, stacked_views(ctx->p, features), P, P_inv, camera_center, nullptr,
                nullptr, nullptr, nullptr, nullptr, Sp, nullptr, nullptr, nullptr};
    launch_sweep<0, false>(ctx, a, true, S(stream));
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_mvcnn_depth(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                   const float *P, const float *P_inv, const float *camera_center, float *Sp,
                   float *points, float *depth_map, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !features || !P || !P_inv || !camera_center || !Sp ||
        !points || !depth_map)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    SweepArgs a{n, ray_idxs, stacked_views(ctx->p, features), P, P_inv, camera_center, nullptr,
                nullptr, nullptr, nullptr, nullptr, Sp, nullptr, depth_map, points};
    launch_sweep<0, false>(ctx, a, true, S(stream));
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

static int prefix_api(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                      const float *P, const float *P_inv, const float *cc, int32_t *rvi,
                      int32_t *rvc, float *S_voxel, hipStream_t st) {
    // raynet_fp.py:55-104: traversal (thread per ray), then sweep + mapping (wave per ray)
    {
        ProfScope prof(ctx, RN_K_TRAVERSE, n, st);
        hipLaunchKernelGGL((k_traverse<false>), dim3((n + WAVE - 1) / WAVE), dim3(WAVE), 0, st, ctx->p,
                           n, ray_idxs, P_inv, cc, (const float *)nullptr, (const float *)nullptr,
                           rvi, rvc, 0, (int64_t)0, (float *)nullptr);
    }
    RN_LAUNCH_CHECK(ctx);
    SweepArgs a{n, ray_idxs, stacked_views(ctx->p, features), P, P_inv, cc, nullptr, nullptr,
                nullptr, rvi, rvc, nullptr, S_voxel, nullptr, nullptr};
    launch_sweep<1, false>(ctx, a, true, st);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_mvcnn_voxel_space(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                         const float *P, const float *P_inv, const float *camera_center,
                         int32_t *rvi, int32_t *rvc, float *S_voxel, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !features || !P || !P_inv || !camera_center || !rvi ||
        !rvc || !S_voxel)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    return prefix_api(ctx, n, ray_idxs, features, P, P_inv, camera_center, rvi, rvc, S_voxel,
                      S(stream));
}

int rn_mvcnn_voxel_space_depth(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs,
                               const float *features, const float *P, const float *P_inv,
                               const float *camera_center, int32_t *rvi, int32_t *rvc,
                               float *S_voxel, float *depth_map, void *stream) {
    if (!depth_map) return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = rn_mvcnn_voxel_space(ctx, n, ray_idxs, features, P, P_inv, camera_center, rvi, rvc,
                                  S_voxel, stream);
    if (rc || n == 0) return rc;
    ProfScope prof(ctx, RN_K_DEPTH, n, S(stream));
    hipLaunchKernelGGL((k_argmax_depth<false>), dim3(ray_blocks(n)), dim3(BLOCK), 0, S(stream),
                       ctx->p, n, S_voxel, rvi, rvc, ctx->axes, camera_center, depth_map);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_fused_bp_sweep(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                      const float *P, const float *P_inv, const float *camera_center,
                      int32_t *rvi, int32_t *rvc, float *S_voxel, const float *acc_in,
                      const float *msgs_in, float *acc_out, float *msgs_out, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !features || !P || !P_inv || !camera_center || !rvi ||
        !rvc || !S_voxel || !acc_in || !msgs_in || !acc_out || !msgs_out)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    rc = prefix_api(ctx, n, ray_idxs, features, P, P_inv, camera_center, rvi, rvc, S_voxel,
                    S(stream));
    if (rc) return rc;
    return launch_bp<false, true>(ctx, n, S_voxel, rvi, rvc, acc_in, msgs_in, acc_out, msgs_out,
                                  S(stream));
}

int rn_fused_depth(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                   const float *P, const float *P_inv, const float *camera_center, int32_t *rvi,
                   int32_t *rvc, float *S_voxel, const float *acc, const float *msgs,
                   float *depth_map, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !features || !P || !P_inv || !camera_center || !rvi ||
        !rvc || !S_voxel || !acc || !msgs || !depth_map)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    rc = prefix_api(ctx, n, ray_idxs, features, P, P_inv, camera_center, rvi, rvc, S_voxel,
                    S(stream));
    if (rc) return rc;
    // the reference overwrites S_voxel_space with the final distribution (raynet_fp.py:183-190)
    return launch_depth<false, true>(ctx, n, S_voxel, rvi, rvc, acc, msgs, camera_center, S_voxel,
                                     depth_map, S(stream));
}

// ------------------------------------------------------ resident-scene path
int rn_acc_copies(const rn_ctx *ctx) { return ctx ? 1 : 0; }

int rn_scatter_reset(rn_ctx *ctx) {
    if (!ctx) return RN_ERR_INVALID;
    ctx->box_level = ctx->box_level0;
    ctx->box_obs[0] = ctx->box_obs[1] = 0;
    ctx->box_probe = BOX_PROBE_LAUNCHES;
    // launches since the last look at the (cumulative) counters belong to the old scene
    ctx->box_rebase = ctx->box_probe_used;
    return RN_OK;
}

int rn_scatter_settled(const rn_ctx *ctx) {
    // no scatter launch copies its overflow counters out any more: the tile shape stays as it is
    // until the next rn_scatter_reset / rn_set_options (what a captured step relies on)
    return ctx && ctx->box_probe == 0 ? 1 : 0;
}

int64_t rn_slab_boxes_size(const rn_ctx *ctx, int64_t rows) {
    return ctx && rows >= 0 ? ((rows + WAVE - 1) / WAVE) * slab_box_count(ctx->p.M) * 2 : 0;
}

int rn_scene_bind_slab_boxes(rn_ctx *ctx, const int32_t *vox, int64_t rows, int32_t *boxes) {
    if (!ctx || rows < 0 || (boxes && !vox)) return fail(ctx, RN_ERR_INVALID, "bad argument");
    ctx->sb_vox = boxes ? vox : nullptr;
    ctx->sb_rows = boxes ? rows : 0;
    ctx->sb_boxes = reinterpret_cast<int2 *>(boxes);
    ctx->sb_valid_lo = ctx->sb_valid_hi = 0;
    return RN_OK;
}

int rn_scene_bind_scatter_items(rn_ctx *ctx, const int32_t *vox, int64_t rows, int32_t level,
                                const int32_t *items, int32_t count) {
    if (!ctx || rows < 0 || count < 0 || level < 0 || level > 1 || (items && (!vox || count < 1)))
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    // an item is `tile << 12 | first << 6 | count` in 32 bits: 19 bits of tile index
    if (items && (rows + (level == 0 ? 127 : 255)) / (level == 0 ? 128 : 256) > (int64_t(1) << 19))
        return fail(ctx, RN_ERR_INVALID, "rn_scene_bind_scatter_items: %lld rows are more than "
                    "2^19 tiles of level %d; use the tiles x split launch (no work list)",
                    (long long)rows, level);
    ctx->sc_vox = items ? vox : nullptr;
    ctx->sc_items = items;
    ctx->sc_rows = items ? rows : 0;
    ctx->sc_level = level;
    ctx->sc_count = items ? count : 0;
    return RN_OK;
}

int rn_scatter_state(const rn_ctx *ctx, int32_t *level, uint32_t *chunks, uint32_t *overflowed) {
    if (!ctx || !level || !chunks || !overflowed) return RN_ERR_INVALID;
    *level = ctx->box_level;
    *chunks = ctx->box_obs[0];
    *overflowed = ctx->box_obs[1];
    return RN_OK;
}

int64_t rn_acc_size(const rn_ctx *ctx) { return ctx ? acc_floats(ctx) : 0; }

int rn_acc_to_grid(rn_ctx *ctx, const float *acc, float *grid_out, void *stream) {
    if (!ctx || !acc || !grid_out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int64_t G = (int64_t)ctx->p.gx * ctx->p.gy * ctx->p.gz;
    hipLaunchKernelGGL((k_acc_regrid<true>), dim3(fill_blocks(G)), dim3(BLOCK), 0, S(stream),
                       ctx->p, acc, grid_out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_stitch_rows(rn_ctx *ctx, int64_t n, const float *rows, const int32_t *index, float *out,
                   void *stream) {
    if (!ctx || n < 0 || (n && (!rows || !index || !out)) || ((uintptr_t)out & 15) ||
        ((uintptr_t)index & 15))
        return fail(ctx, RN_ERR_INVALID, "rn_stitch_rows: bad argument");
    if (n == 0) return RN_OK;
    // page-locked host memory: the kernel writes through its device-side address
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeHost) {
        if (!at.devicePointer)
            return fail(ctx, RN_ERR_INVALID, "rn_stitch_rows: host memory that is not mapped");
        out = static_cast<float *>(at.devicePointer);
    } else {
        (void)hipGetLastError();       // (an unregistered pointer: reported by the launch)
    }
    // 64 workgroups keep a PCIe link busy (16: 3 % slower per step of an eight-rank shard); one
    // per CU only takes issue slots from the depth sweep this launch runs under (300: 1.5 %)
    int nb = fill_blocks((n + 3) / 4);
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(k_stitch_rows, dim3(nb), dim3(BLOCK), 0, S(stream), n,
                       rows, index, out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_acc_from_grid(rn_ctx *ctx, const float *grid, float *acc_out, void *stream) {
    if (!ctx || !grid || !acc_out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int64_t G = (int64_t)ctx->p.gx * ctx->p.gy * ctx->p.gz;
    hipLaunchKernelGGL((k_acc_regrid<false>), dim3(fill_blocks(G)), dim3(BLOCK), 0, S(stream),
                       ctx->p, grid, acc_out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_scene_prepare(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs,
                     const float *const *features_views_host, const float *P, const float *P_inv,
                     const float *camera_center, const int32_t *order, int32_t *vox, int32_t *rvc,
                     float *Sr, float *ray_segments, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !ray_idxs || !features_views_host || !P || !P_inv || !camera_center ||
        !vox || !rvc || !Sr)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    if (ctx->sb_boxes && vox >= ctx->sb_vox && vox < ctx->sb_vox + ctx->sb_rows * (int64_t)ctx->p.M)
        ctx->sb_valid_lo = ctx->sb_valid_hi = 0;      // this entry does not maintain slab boxes
    FeatureViews fv;
    for (int v = 0; v < MAX_VIEWS; v++) fv.v[v] = v < ctx->p.N ? features_views_host[v] : nullptr;
    for (int v = 0; v < ctx->p.N; v++)
        if (!fv.v[v]) return fail(ctx, RN_ERR_INVALID, "null feature map for view %d", v);
    {
        ProfScope prof(ctx, RN_K_TRAVERSE, n, S(stream));
        hipLaunchKernelGGL((k_traverse<true>), dim3((n + WAVE - 1) / WAVE), dim3(WAVE), 0, S(stream),
                           ctx->p, n, ray_idxs, P_inv, camera_center, (const float *)nullptr,
                           (const float *)nullptr, vox, rvc, 0, (int64_t)0, ray_segments);
    }
    RN_LAUNCH_CHECK(ctx);
    SweepArgs a{n, ray_idxs, fv, P, P_inv, camera_center, nullptr, nullptr, nullptr, vox, rvc,
                nullptr, Sr, nullptr, nullptr};
    a.order = order;
    a.seg = ray_segments;
    launch_sweep<2, true>(ctx, a, true, S(stream));
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

}  // extern "C"

namespace {
// LDS a workgroup of the plane sweep may take and still leave room for RN_SWEEP_MIN_WAVES
// wavefronts per SIMD (4 per workgroup, 160 KB per CU)
inline bool fold_fits(const Params &p) {
    return sweep_lds(p, 3) <= (size_t)160 * 1024 / ((RN_SWEEP_MIN_WAVES * 4 + SWEEP_WAVES - 1) / SWEEP_WAVES);
}
}  // namespace

// rn_scene_prepare_all; with msgs_fold the plane sweep also writes BP iteration 0's messages
// (k_sweep_map MAPMODE 3, first_sweep_messages)
// The one occupancy of BP iteration 0 (the prior in every voxel, no messages), evaluated by the
// device's own arithmetic once per prior value and kept with the context.
static int first_occupancy(rn_ctx *ctx, float prior, hipStream_t st, float *out) {
    if (!ctx->have_first_occ || ctx->first_prior != prior) {
        hipLaunchKernelGGL(k_first_occupancy, dim3(1), dim3(1), 0, st, prior, ctx->scalar_dev);
        hipError_t e = hipMemcpyAsync(&ctx->first_occ, ctx->scalar_dev, sizeof(float),
                                      hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);      // (once per prior value)
        if (e != hipSuccess) return fail(ctx, RN_ERR_HIP, "first_occupancy: %s", hipGetErrorString(e));
        ctx->first_prior = prior;
        ctx->have_first_occ = true;
    }
    *out = ctx->first_occ;
    return RN_OK;
}

static int scene_prepare_all_impl(rn_ctx *ctx, int32_t n_images, int32_t n, int64_t rows_per_image,
                                  const int32_t *ray_idxs, const float *const *features_views,
                                  const float *cameras, const int32_t *order, int32_t *vox,
                                  int32_t *rvc, float *Sr, float *ray_segments, void *stream,
                                  float *msgs_fold, float prior, float *zero_fold = nullptr,
                                  int sweep_xcd_chunk = 0) {
    if (ctx && n == 0 && n_images >= 1) return RN_OK;   /* a rank without rays: pointers may be null */
    if (!ctx || n_images < 1 || n < 0 || rows_per_image < n || !ray_idxs || !features_views ||
        !cameras || !vox || !rvc || !Sr)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    const int N = ctx->p.N;
    const int cam_stride = 12 * N + 12 + 4;
    const size_t M = (size_t)ctx->p.M;
    float o_first = 0.0f;
    if (msgs_fold) {
        rc = first_occupancy(ctx, prior, S(stream), &o_first);
        if (rc) return rc;
    }
    // slab boxes for the scatter, when a table is bound to this list buffer (rows_per_image is
    // then a multiple of 64 by the binding's contract: checked)
    int2 *boxes = rows_per_image % WAVE == 0
                      ? slab_boxes_for(ctx, vox, (int64_t)n_images * rows_per_image, false) : nullptr;
    if (ctx->sb_boxes && vox >= ctx->sb_vox && vox < ctx->sb_vox + ctx->sb_rows * (int64_t)M) {
        if (boxes) {
            ctx->sb_valid_lo = (vox - ctx->sb_vox) / (int64_t)M;
            ctx->sb_valid_hi = ctx->sb_valid_lo + (int64_t)n_images * rows_per_image;
        } else {
            ctx->sb_valid_lo = ctx->sb_valid_hi = 0;
        }
    }
    // traverse + sweep of images [g0, g0 + ng); the traversal on `trav_st`, the sweep on `st`
    auto traverse = [&](int g0, int ng, hipStream_t trav_st) {
        const float *cam = cameras + (size_t)g0 * cam_stride;
        const size_t row0 = (size_t)g0 * rows_per_image;
        ProfScope prof(ctx, RN_K_TRAVERSE, n * ng, trav_st);
        hipLaunchKernelGGL((k_traverse<true>), dim3((n + WAVE - 1) / WAVE, ng), dim3(WAVE), 0,
                           trav_st, ctx->p, n, ray_idxs, cam + 12 * N, cam + 12 * N + 12,
                           (const float *)nullptr, (const float *)nullptr, vox + row0 * M,
                           rvc + row0, cam_stride, rows_per_image,
                           ray_segments ? ray_segments + row0 * 8 : nullptr,
                           boxes ? boxes + (row0 / WAVE) * slab_box_count(ctx->p.M) : nullptr);
    };
    auto sweep = [&](int g0, int ng, hipStream_t st) {
        const float *cam = cameras + (size_t)g0 * cam_stride;
        const size_t row0 = (size_t)g0 * rows_per_image;
        SweepArgs a{n, ray_idxs, FeatureViews{}, cam, cam + 12 * N, cam + 12 * N + 12, nullptr,
                    nullptr, nullptr, vox + row0 * M, rvc + row0, nullptr, Sr + row0 * M, nullptr,
                    nullptr};
        a.order = order;
        a.fv_table = features_views + (size_t)g0 * N;
        a.cam_stride = cam_stride;
        a.rows_per_image = rows_per_image;
        a.n_images = ng;
        a.seg = ray_segments ? ray_segments + row0 * 8 : nullptr;
        a.xcd_chunk = sweep_xcd_chunk / SWEEP_WAVES;
        if (msgs_fold) {
            a.msgs_out = msgs_fold + row0 * M;
            a.prior = o_first;
            a.zero = g0 == 0 ? zero_fold : nullptr;      // (once per pass)
            launch_sweep<3, true>(ctx, a, true, st);
        } else {
            launch_sweep<2, true>(ctx, a, true, st);
        }
    };
    const bool split = ctx->overlap == 1 || (ctx->overlap == 2 && ctx->box_level == 1);
    const int gA = split && n_images >= 2 && (int64_t)n * n_images >= 65536
                       ? (n_images + 1) / 2 : n_images;
    if (gA < n_images) {
        // the thread-per-ray traversal (a chain of dependent fp32 additions per ray, few waves
        // per CU) of the second half of the images hides under the plane sweep of the first
        RN_HIP(ctx, hipEventRecord(ctx->ev_fork, S(stream)));
        RN_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        traverse(0, gA, S(stream));
        RN_LAUNCH_CHECK(ctx);
        traverse(gA, n_images - gA, ctx->aux);
        RN_LAUNCH_CHECK(ctx);
        RN_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
        sweep(0, gA, S(stream));
        RN_LAUNCH_CHECK(ctx);
        RN_HIP(ctx, hipStreamWaitEvent(S(stream), ctx->ev_join, 0));
        sweep(gA, n_images - gA, S(stream));
        RN_LAUNCH_CHECK(ctx);
    } else {
        traverse(0, n_images, S(stream));
        RN_LAUNCH_CHECK(ctx);
        sweep(0, n_images, S(stream));
        RN_LAUNCH_CHECK(ctx);
    }
    return RN_OK;
}

extern "C" {

int rn_scene_prepare_all(rn_ctx *ctx, int32_t n_images, int32_t n, int64_t rows_per_image,
                         const int32_t *ray_idxs, const float *const *features_views,
                         const float *cameras, const int32_t *order, int32_t *vox, int32_t *rvc,
                         float *Sr, float *ray_segments, void *stream) {
    return scene_prepare_all_impl(ctx, n_images, n, rows_per_image, ray_idxs, features_views, cameras,
                                  order, vox, rvc, Sr, ray_segments, stream, nullptr, 0.0f);
}

int rn_scene_count_voxels(rn_ctx *ctx, int32_t n_images, int32_t n, const int32_t *ray_idxs,
                          const float *cameras, int32_t *rvc, void *stream) {
    if (ctx && n == 0 && n_images >= 1) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n_images < 1 || n < 0 || !ray_idxs || !cameras || !rvc)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int N = ctx->p.N;
    const int cam_stride = 12 * N + 12 + 4;
    ProfScope prof(ctx, RN_K_TRAVERSE, n * n_images, S(stream));
    hipLaunchKernelGGL((k_traverse<true>), dim3((n + WAVE - 1) / WAVE, n_images), dim3(WAVE), 0,
                       S(stream), ctx->p, n, ray_idxs, cameras + 12 * N, cameras + 12 * N + 12,
                       (const float *)nullptr, (const float *)nullptr, (int32_t *)nullptr, rvc,
                       cam_stride, (int64_t)n, (float *)nullptr);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_scene_bp_sweep(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *vox,
                      const int32_t *rvc, const float *acc_in, float *msgs, float *acc_part,
                      int32_t first_sweep, int32_t row_layout, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !Sr || !vox || !rvc || !acc_in || !msgs || !acc_part)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (n == 0) return RN_OK;
    return launch_bp<true, false>(ctx, n, Sr, vox, rvc, acc_in,
                                  (first_sweep & RN_SWEEP_ZERO_MSGS) ? nullptr : msgs, acc_part,
                                  msgs, S(stream), row_layout == RN_ROWS_PATCHES, false,
                                  uniform_mode((first_sweep & RN_SWEEP_UNIFORM_ACC) != 0));
}

int rn_scene_bp_sweep_fixed(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *vox,
                            const int32_t *rvc, const float *acc_in, float *msgs,
                            int64_t *acc_part_fixed, int32_t first_sweep, int32_t row_layout,
                            void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !Sr || !vox || !rvc || !acc_in || !msgs || !acc_part_fixed)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    return launch_bp<true, false>(ctx, n, Sr, vox, rvc, acc_in,
                                  (first_sweep & RN_SWEEP_ZERO_MSGS) ? nullptr : msgs,
                                  acc_part_fixed, msgs, S(stream), row_layout == RN_ROWS_PATCHES,
                                  true, uniform_mode((first_sweep & RN_SWEEP_UNIFORM_ACC) != 0));
}

int rn_acc_combine_fixed(rn_ctx *ctx, int64_t *acc_part_fixed, float prior, float *acc_out,
                         void *stream) {
    if (!ctx || !acc_part_fixed || !acc_out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int64_t G = acc_floats(ctx);
    ProfScope prof(ctx, RN_K_ACC, 0, S(stream));
    hipLaunchKernelGGL(k_acc_combine_fixed, dim3(fill_blocks(G)), dim3(BLOCK), 0, S(stream),
                       reinterpret_cast<unsigned long long *>(acc_part_fixed), G, prior, acc_out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_acc_combine_fixed_range(rn_ctx *ctx, int64_t *acc_part_fixed, int64_t count, float prior,
                               float *acc_out, void *stream) {
    if (!ctx || !acc_part_fixed || !acc_out || count < 0)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (count == 0) return RN_OK;
    ProfScope prof(ctx, RN_K_ACC, 0, S(stream));
    hipLaunchKernelGGL(k_acc_combine_fixed, dim3(fill_blocks(count)), dim3(BLOCK), 0, S(stream),
                       reinterpret_cast<unsigned long long *>(acc_part_fixed), count, prior, acc_out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_acc_combine(rn_ctx *ctx, float *acc_part, float prior, float *acc_out, void *stream) {
    if (!ctx || !acc_part || !acc_out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int64_t G = acc_floats(ctx);
    ProfScope prof(ctx, RN_K_ACC, 0, S(stream));
    hipLaunchKernelGGL(k_acc_combine, dim3(fill_blocks(G)), dim3(BLOCK), 0, S(stream), acc_part,
                       1, G, prior, acc_out, 1);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_acc_reduce_local(rn_ctx *ctx, float *acc_part, float *acc_out, void *stream) {
    if (!ctx || !acc_part || !acc_out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int64_t G = acc_floats(ctx);
    ProfScope prof(ctx, RN_K_ACC, 0, S(stream));
    hipLaunchKernelGGL(k_acc_combine, dim3(fill_blocks(G)), dim3(BLOCK), 0, S(stream), acc_part,
                       1, G, 0.0f, acc_out, 0);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_acc_add_prior(rn_ctx *ctx, float *acc, float prior, void *stream) {
    if (!ctx || !acc) return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int64_t G = acc_floats(ctx);
    hipLaunchKernelGGL(k_add_scalar, dim3(fill_blocks(G)), dim3(BLOCK), 0, S(stream), acc, G,
                       prior);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_scene_depth(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *vox,
                   const int32_t *rvc, const float *acc, const float *msgs,
                   const float *camera_center, int32_t rays_per_center, float *S_new,
                   float *depth_map, void *stream) {
    if (ctx && n == 0) return RN_OK;   /* empty launch: pointers may be null */
    if (!ctx || n < 0 || !Sr || !vox || !rvc || !acc || !msgs || (!S_new && !depth_map) ||
        (depth_map && !camera_center))
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    if (rays_per_center < 0) return fail(ctx, RN_ERR_INVALID, "bad argument");
    return launch_depth<true, false>(ctx, n, Sr, vox, rvc, acc, msgs, camera_center, S_new,
                                     depth_map, S(stream), rays_per_center);
}

int rn_scene_run(rn_ctx *ctx, const rn_scene_plan *pl, int32_t phases, int32_t iteration,
                 int32_t image, void *stream) {
    if (!ctx || !pl || pl->n_images < 1 || pl->n < 0 || pl->rows_per_image < pl->n ||
        pl->rows_per_image % 256 || iteration < 0 ||
        (!(phases & RN_RUN_DEPTH_RANGE) && image >= pl->n_images) ||
        (!pl->ray_idxs && pl->n) || !pl->features_views || !pl->cameras || !pl->vox || !pl->rvc || !pl->Sr || !pl->msgs ||
        !pl->acc[0] || !pl->acc[1] || !pl->depth ||
        (pl->depth_image && pl->depth_image_stride < 1) || pl->sweep_xcd_chunk < 0 ||
        pl->sweep_xcd_chunk % 4 ||
        (phases & ~(RN_RUN_PREPARE | RN_RUN_SWEEP | RN_RUN_COMBINE | RN_RUN_DEPTH | RN_RUN_DEPTH_RANGE)) ||
        ((phases & RN_RUN_DEPTH_RANGE) &&
         (image < 0 || (image >> 16) < 1 || (image & 0xffff) + (image >> 16) > pl->n_images ||
          (phases & RN_RUN_DEPTH))))
        return fail(ctx, RN_ERR_INVALID, "rn_scene_run: bad plan or phase");
    int rc = need_axes(ctx);
    if (rc) return rc;
    const bool fixed = pl->acc_fixed != nullptr;
    const int64_t rows = (int64_t)pl->n_images * pl->rows_per_image;
    if (rows > 0x7fffffff) return fail(ctx, RN_ERR_INVALID, "rn_scene_run: too many rows");
    hipStream_t st = S(stream);
    const int64_t G = acc_floats(ctx);
    bool folded = false;
    if (phases & RN_RUN_PREPARE) {
        if (fixed) RN_HIP(ctx, hipMemsetAsync(pl->acc_fixed, 0, sizeof(int64_t) * G, st));
        // K1 prefix and BP iteration 0 requested together: the plane sweep writes the first
        // messages itself (one occupancy for every voxel, nothing to gather) while the column
        // is still in LDS; SWEEP(0) below is then the scatter alone
        folded = (phases & RN_RUN_SWEEP) && iteration == 0 && fold_fits(ctx->p);
        if (pl->n > 0)
            rc = scene_prepare_all_impl(ctx, pl->n_images, pl->n, pl->rows_per_image, pl->ray_idxs,
                                        pl->features_views, pl->cameras, pl->order, pl->vox,
                                        pl->rvc, pl->Sr, pl->ray_segments, stream,
                                        folded ? pl->msgs : nullptr, pl->prior,
                                        folded && !fixed ? pl->acc[0] : nullptr,
                                        pl->sweep_xcd_chunk);
        if (rc) return rc;
    }
    if (pl->n == 0) {
        // a rank without rays (tiny images, many ranks) still takes part in the exchange: its
        // partial sums are all zero
        if ((phases & RN_RUN_SWEEP) && !fixed)
            RN_HIP(ctx, hipMemsetAsync(pl->acc[iteration & 1], 0, sizeof(float) * G, st));
        if ((phases & RN_RUN_COMBINE) && fixed)
            return rn_acc_combine_fixed(ctx, pl->acc_fixed, pl->prior, pl->acc[iteration & 1], stream);
        return RN_OK;
    }
    if (phases & RN_RUN_SWEEP) {
        AccMode am;
        am.uniform = iteration == 0;        // the prior everywhere, no messages yet
        am.biased = !fixed || iteration == 0;
        am.bias = pl->prior;
        am.zero = fixed ? nullptr : pl->acc[iteration & 1];
        rc = launch_bp<true, false>(ctx, (int)rows, pl->Sr, pl->vox, pl->rvc,
                                    pl->acc[(iteration + 1) & 1], iteration == 0 ? nullptr : pl->msgs,
                                    fixed ? (void *)pl->acc_fixed : (void *)pl->acc[iteration & 1],
                                    pl->msgs, st, pl->row_layout == RN_ROWS_PATCHES, fixed, am,
                                    folded);
        if (rc) return rc;
    }
    if ((phases & RN_RUN_COMBINE) && fixed) {
        rc = rn_acc_combine_fixed(ctx, pl->acc_fixed, pl->prior, pl->acc[iteration & 1], stream);
        if (rc) return rc;
    }
    if (phases & (RN_RUN_DEPTH | RN_RUN_DEPTH_RANGE)) {
        AccMode am;
        am.biased = !fixed;
        am.bias = pl->prior;
        const float *acc = pl->acc[(iteration + 1) & 1];
        const int cam_stride = 12 * ctx->p.N + 12 + 4;
        const float *cc = pl->cameras + 12 * ctx->p.N + 12;
        const size_t M = (size_t)ctx->p.M;
        // depth_image: the maps leave in pixel order (no reordering pass behind the sweep)
        DepthDest dest;
        if (pl->depth_image) {
            dest.pixel_of_row = pl->ray_idxs;
            dest.rows = pl->n;
            dest.image_stride = pl->depth_image_stride;
        }
        auto out = [&](int first) {
            return pl->depth_image ? pl->depth_image + (size_t)first * pl->depth_image_stride
                                   : pl->depth + (size_t)first * pl->rows_per_image;
        };
        if (image < 0)
            return launch_depth<true, false>(ctx, (int)rows, pl->Sr, pl->vox, pl->rvc, acc, pl->msgs, cc,
                                             nullptr, out(0), st, (int)pl->rows_per_image, am,
                                             cam_stride, dest);
        if (phases & RN_RUN_DEPTH_RANGE) {
            const int first = image & 0xffff, count = image >> 16;
            const size_t r0 = (size_t)first * pl->rows_per_image;
            return launch_depth<true, false>(ctx, (int)(count * pl->rows_per_image), pl->Sr + r0 * M,
                                             pl->vox + r0 * M, pl->rvc + r0, acc, pl->msgs + r0 * M,
                                             cc + (size_t)first * cam_stride, nullptr, out(first),
                                             st, (int)pl->rows_per_image, am, cam_stride, dest);
        }
        const size_t row0 = (size_t)image * pl->rows_per_image;
        // (one image: a single group of rows_per_image >= n rows)
        return launch_depth<true, false>(ctx, pl->n, pl->Sr + row0 * M, pl->vox + row0 * M,
                                         pl->rvc + row0, acc, pl->msgs + row0 * M,
                                         cc + (size_t)image * cam_stride, nullptr, out(image),
                                         st, 0, am, 4, dest);
    }
    return RN_OK;
}

int rn_prof_begin(rn_ctx *ctx, int32_t capacity) {
    if (!ctx || capacity < 1) return fail(ctx, RN_ERR_INVALID, "bad argument");
    if (capacity > ctx->prof_cap) {
        hipEvent_t *ev = new hipEvent_t[2 * capacity];
        for (int i = 0; i < 2 * ctx->prof_cap; i++) ev[i] = ctx->prof_ev[i];
        for (int i = 2 * ctx->prof_cap; i < 2 * capacity; i++) RN_HIP(ctx, hipEventCreate(&ev[i]));
        delete[] ctx->prof_ev;
        delete[] ctx->prof_id;
        delete[] ctx->prof_rays;
        ctx->prof_ev = ev;
        ctx->prof_id = new int32_t[capacity];
        ctx->prof_rays = new int32_t[capacity];
        ctx->prof_cap = capacity;
    }
    ctx->prof_n = 0;
    ctx->prof_on = true;
    return RN_OK;
}

int rn_prof_select(rn_ctx *ctx, uint32_t kernel_mask) {
    if (!ctx) return RN_ERR_INVALID;
    ctx->prof_mask = kernel_mask;
    return RN_OK;
}

int rn_prof_end(rn_ctx *ctx, int32_t *count, int32_t *kernel_ids_host, int32_t *n_rays_host,
                float *ms_host) {
    if (!ctx || !count) return fail(ctx, RN_ERR_INVALID, "bad argument");
    ctx->prof_on = false;
    const int n = ctx->prof_n;
    for (int i = 0; i < n; i++) {
        RN_HIP(ctx, hipEventSynchronize(ctx->prof_ev[2 * i + 1]));
        float ms = 0.0f;
        RN_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        if (ms_host) ms_host[i] = ms;
        if (kernel_ids_host) kernel_ids_host[i] = ctx->prof_id[i];
        if (n_rays_host) n_rays_host[i] = ctx->prof_rays[i];
    }
    *count = n;
    return RN_OK;
}

int rn_selftest_arith(rn_ctx *ctx, int32_t n, const float *a, float *out, void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !a || !out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_selftest_arith, dim3(thread_blocks(n)), dim3(BLOCK), 0, S(stream), n, a, out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_selftest_quotient(rn_ctx *ctx, int32_t n, const float *x, const float *d, float *out,
                         void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !x || !d || !out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_selftest_quotient, dim3(thread_blocks(n)), dim3(BLOCK), 0, S(stream), n, x,
                       d, out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_selftest_feature_offsets(rn_ctx *ctx, int32_t n, const float *P, const float *ray_start,
                                const float *ray_end, int32_t *out, void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !P || !ray_start || !ray_end || !out)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_selftest_offsets, dim3(ray_blocks(n)), dim3(BLOCK), 0, S(stream), ctx->p, n,
                       P, ray_start, ray_end, out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_selftest_mapping(rn_ctx *ctx, int32_t n, const float *a, const float *b, const float *t,
                        float *out, void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !a || !b || !t || !out) return fail(ctx, RN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_selftest_mapping, dim3(thread_blocks(n)), dim3(BLOCK),
                       sizeof(float) * (ctx->p.D + 1), S(stream), ctx->p, n, a, b, t, out);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_prof_offsets(rn_ctx *ctx, float *start_ms_host) {
    if (!ctx || !start_ms_host) return fail(ctx, RN_ERR_INVALID, "bad argument");
    for (int i = 0; i < ctx->prof_n; i++)
        RN_HIP(ctx, hipEventElapsedTime(start_ms_host + i, ctx->prof_ev[0], ctx->prof_ev[2 * i]));
    return RN_OK;
}


int rn_timer_start(rn_ctx *ctx, void *stream) {
    if (!ctx) return RN_ERR_INVALID;
    RN_HIP(ctx, hipEventRecord(ctx->ev0, S(stream)));
    return RN_OK;
}

int rn_timer_stop(rn_ctx *ctx, void *stream, float *ms_out) {
    if (!ctx || !ms_out) return RN_ERR_INVALID;
    RN_HIP(ctx, hipEventRecord(ctx->ev1, S(stream)));
    RN_HIP(ctx, hipEventSynchronize(ctx->ev1));
    RN_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
    return RN_OK;
}

}  // extern "C"

// training (differentiable) entry points, SURVEY.md 8f row 2
#include "raynet_train.inl"

// depth maps -> point cloud -> accuracy / completeness, SURVEY.md 8f row 3
#include "raynet_eval.inl"
