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
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int NXCD = 8;

// XCD-aware remap: the dispatcher is observed to place block b on XCD b % 8; give
// every XCD one contiguous slice of the ray list so its private L2 sees
// neighbouring rays (speed only -- any placement is correct).
__device__ __forceinline__ int xcd_block(int b, int nblocks) {
    const int q = nblocks / NXCD, r = nblocks % NXCD;
    const int xcd = b % NXCD, pos = b / NXCD;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + pos;
}

__device__ __forceinline__ int ray_of_wave(int n, int &lane) {
    lane = threadIdx.x & (WAVE - 1);
    const int b = xcd_block(blockIdx.x, gridDim.x);
    const int r = b * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    return r < n ? uniform(r) : -1;
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
__global__ void k_add_scalar(float *a, int64_t n, float v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        a[i] = v + a[i];
}

// ------------------------------------------------------------ K8 / sampling
__global__ void k_sample_rays(Params p, int n, const int32_t *__restrict__ ray_idxs,
                              const float *__restrict__ P_inv, const float *__restrict__ cc,
                              float *starts, float *ends) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float s[3], e[3];
    sample_in_bbox(p, ray_idxs[r], P_inv, cc, s, e);
    for (int i = 0; i < 3; i++) {
        starts[3 * r + i] = s[i];
        ends[3 * r + i] = e[i];
    }
}
// sampling_schemes.cu:92-122: one wave per ray, lanes over planes
__global__ void k_sample_points(Params p, int n, const int32_t *__restrict__ ray_idxs,
                                const float *__restrict__ P_inv, const float *__restrict__ cc,
                                float *points) {
    int lane;
    const int r = ray_of_wave(n, lane);
    if (r < 0) return;
    float s[3], e[3];
    sample_in_bbox(p, ray_idxs[r], P_inv, cc, s, e);
    float4 *row = reinterpret_cast<float4 *>(points) + (size_t)r * p.D;
    for (int k = lane; k < p.D; k += WAVE) {
        float pt[3];
        plane_point(s, e, k, p.D, pt);
        row[k] = make_float4(pt[0], pt[1], pt[2], 1.0f);
    }
}

// ------------------------------------------------------------ K5 traversal
// One thread per ray (the DDA is a chain of sequential fp32 additions, bit-exactness
// forbids re-associating it).  Source of the segment: explicit starts/ends, or the
// camera (sample_in_bbox), as in the fused kernels.
// A thread writing its own row step by step produces one partial-line write per voxel
// (4x write amplification measured).  Each 64-thread block therefore collects
// [64 rays][TRAV_TILE steps] in LDS and writes finished tiles as coalesced row segments.
constexpr int TRAV_TILE = 32;
template <bool PACKED>
__global__ __launch_bounds__(WAVE) void k_traverse(Params p, int n,
                                                   const int32_t *__restrict__ ray_idxs,
                                                   const float *__restrict__ P_inv,
                                                   const float *__restrict__ cc,
                                                   const float *__restrict__ starts,
                                                   const float *__restrict__ ends, int32_t *vox,
                                                   int32_t *rvc, int cam_stride,
                                                   int64_t rows_per_image, float *seg_out) {
    __shared__ int32_t tile[WAVE * (TRAV_TILE + 1)];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * WAVE;
    if (rows_per_image > 0) {       // blockIdx.y = reference image of a scene-wide launch
        const int g = blockIdx.y;
        P_inv += (size_t)g * cam_stride;
        cc += (size_t)g * cam_stride;
        vox += (size_t)g * rows_per_image * p.M * (PACKED ? 1 : 3);
        rvc += (size_t)g * rows_per_image;
        if (seg_out) seg_out += (size_t)g * rows_per_image * 8;
    }
    const int r = r0 + lane;
    const bool live = r < n;
    float s[3] = {0.f, 0.f, 0.f}, e[3] = {0.f, 0.f, 0.f};
    if (live) {
        if (ray_idxs) {
            sample_in_bbox(p, ray_idxs[r], P_inv, cc, s, e);
        } else {
            for (int i = 0; i < 3; i++) {
                s[i] = starts[3 * r + i];
                e[i] = ends[3 * r + i];
            }
        }
        // the plane sweep (one wavefront per ray) reads the segment back instead of repeating
        // the double-precision back-projection 64 lanes wide
        if (seg_out) {
            reinterpret_cast<float4 *>(seg_out)[2 * (size_t)r] = make_float4(s[0], s[1], s[2], 0.f);
            reinterpret_cast<float4 *>(seg_out)[2 * (size_t)r + 1] = make_float4(e[0], e[1], e[2], 0.f);
        }
    }
    // ---- DDA set-up (ray_tracing.pyx:99-161), identical arithmetic to rn::dda
    const float EPS = 1e-2f;
    const int g[3] = {p.gx, p.gy, p.gz};
    float ss[3], ee[3], bin[3], ray[3], tm[3], td[3];
    int step[3], cur[3], last[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ss[i] = s[i] - p.bbox[i];
        ee[i] = e[i] - p.bbox[i];
        bin[i] = (p.bbox[3 + i] - p.bbox[i]) / g[i];
        ray[i] = ee[i] - ss[i];
        step[i] = ray[i] >= 0 ? 1 : -1;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ss[i] += step[i] * bin[i] * EPS;
        ee[i] -= step[i] * bin[i] * EPS;
        cur[i] = (int)floorf(ss[i] / bin[i]);
        last[i] = (int)floorf(ee[i] / bin[i]);
    }
    bool active = live && !(cur[0] < 0 || cur[0] >= g[0] || cur[1] < 0 || cur[1] >= g[1] ||
                            cur[2] < 0 || cur[2] >= g[2]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        tm[i] = FLT_MAX;
        if (ray[i] != 0) {
            const float c = cur[i] * bin[i];
            float b;
            if (step[i] < 0 && c < ss[i])
                b = c;
            else
                b = c + step[i] * bin[i];
            tm[i] = (b - ss[i]) / ray[i];
        }
        td[i] = ray[i] != 0 ? step[i] * bin[i] / ray[i] : FLT_MAX;
    }
    int cx = cur[0], cy = cur[1], cz = cur[2];
    float tx = tm[0], ty = tm[1], tz = tm[2];
    int count = 0;           // voxels emitted so far by this ray
    // `active` = this ray still has a voxel (cx,cy,cz) to emit at index `count`
    for (int base = 0; base < p.M; base += TRAV_TILE) {
        if (__ballot(active) == 0) break;
        for (int k = 0; k < TRAV_TILE && base + k < p.M; k++) {
            if (active) {
                tile[lane * (TRAV_TILE + 1) + k] = pack_voxel(cx, cy, cz);
                count++;
                // advance (ray_tracing.pyx:166-197)
                if ((cx == last[0] && cy == last[1] && cz == last[2]) || count >= p.M) {
                    active = false;
                } else if (tx < ty) {
                    if (tx < tz) {
                        cx += step[0];
                        if (cx < 0 || cx >= g[0]) active = false;
                        tx += td[0];
                    } else {
                        cz += step[2];
                        if (cz < 0 || cz >= g[2]) active = false;
                        tz += td[2];
                    }
                } else {
                    if (ty < tz) {
                        cy += step[1];
                        if (cy < 0 || cy >= g[1]) active = false;
                        ty += td[1];
                    } else {
                        cz += step[2];
                        if (cz < 0 || cz >= g[2]) active = false;
                        tz += td[2];
                    }
                }
            }
        }
        wave_sync();
        // flush: TRAV_TILE consecutive steps of one ray are one contiguous segment
        constexpr int RPI = WAVE / TRAV_TILE;      // rows per instruction
#pragma unroll 4
        for (int j = 0; j < WAVE; j += RPI) {
            const int row = j + lane / TRAV_TILE;
            const int col = lane % TRAV_TILE;
            const int c = __shfl(count, row);
            if (r0 + row < n && base + col < c) {
                const int v = tile[row * (TRAV_TILE + 1) + col];
                const size_t off = (size_t)(r0 + row) * p.M + base + col;
                if (PACKED) {
                    vox[off] = v;
                } else {
                    vox[3 * off] = v >> 20;
                    vox[3 * off + 1] = (v >> 10) & 1023;
                    vox[3 * off + 2] = v & 1023;
                }
            }
        }
        wave_sync();
    }
    if (live) rvc[r] = count;   // written even when 0
}

// ---------------------------------------- plane sweep (+ mapping) per wavefront
// SIM      0: read the plane column from S_in [n][D] (K6)
//          1: generic sweep (any N, F), 2: cooperative sweep (F = 4*LPS, N = NV)
// MAPMODE  0: write the plane column to S_planes [n][D]            (K7 / K9 / K10)
//          1: map to voxels, write S_voxel = vals / sum             (K6 / K11 / K1 / K2 prefix)
//          2: as 1, then clip_and_renorm (mrf_bp.cu:103-111) -> Sr  (resident-scene path)
// Dynamic LDS: [axes gx+gy+gz][per wave: D plane column][per wave: M values]
template <int SIM, int NV, int LPS, int MAPMODE, bool PACKED>
__global__ __launch_bounds__(BLOCK) void k_sweep_map(
    Params p, int n, const int32_t *__restrict__ ray_idxs, FeatureViews fv,
    const float *__restrict__ P, const float *__restrict__ P_inv, const float *__restrict__ cc,
    const float *__restrict__ starts, const float *__restrict__ ends,
    const float *__restrict__ S_in, const float *__restrict__ axes_g,
    const int32_t *__restrict__ vox, const int32_t *__restrict__ rvc, float *S_planes,
    float *S_voxel, float *depth_from_planes, float *points,
    const int32_t *__restrict__ order, const float *const *__restrict__ fv_table, int cam_stride,
    int64_t rows_per_image, const float *__restrict__ seg) {
    if (rows_per_image > 0) {       // blockIdx.y = reference image of a scene-wide launch
        const int g = blockIdx.y;
        P += (size_t)g * cam_stride;
        P_inv += (size_t)g * cam_stride;
        cc += (size_t)g * cam_stride;
        fv_table += (size_t)g * p.N;
        vox += (size_t)g * rows_per_image * p.M * (PACKED ? 1 : 3);
        rvc += (size_t)g * rows_per_image;
        S_voxel += (size_t)g * rows_per_image * p.M;
        if (seg) seg += (size_t)g * rows_per_image * 8;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int naxes = p.gx + p.gy + p.gz;
    float *axes = smem;
    const int wid = threadIdx.x >> 6;
    float *Sl = smem + ((naxes + 3) & ~3) + wid * (p.D + p.M);
    float *vals = Sl + p.D;
    if (MAPMODE != 0) {
        for (int i = threadIdx.x; i < naxes; i += BLOCK) axes[i] = axes_g[i];
        __syncthreads();
    }
    int lane;
    int r = ray_of_wave(n, lane);
    if (r < 0) return;
    if (order) r = uniform(order[r]);      // schedule only: which ray this wavefront takes

    float s[3], e[3];
    if (seg) {                      // k_traverse's endpoints of this very row (same arithmetic)
        const float4 a = reinterpret_cast<const float4 *>(seg)[2 * (size_t)r];
        const float4 b = reinterpret_cast<const float4 *>(seg)[2 * (size_t)r + 1];
        s[0] = a.x; s[1] = a.y; s[2] = a.z;
        e[0] = b.x; e[1] = b.y; e[2] = b.z;
    } else if (ray_idxs) {
        sample_in_bbox(p, ray_idxs[r], P_inv, cc, s, e);
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            s[i] = starts[3 * r + i];
            e[i] = ends[3 * r + i];
        }
    }

    if (SIM == 0) {
        for (int k = lane; k < p.D; k += WAVE) Sl[k] = S_in[(size_t)r * p.D + k];
    } else {
        if (SIM == 1)
            sweep_generic(p, fv, fv_table, P, s, e, lane, Sl);
        else
            sweep_coop<NV, LPS>(p, fv, fv_table, P, s, e, lane, Sl);
        wave_sync();
        softmax_column<MAPMODE == 2>(p.D, lane, Sl);
    }
    wave_sync();

    if (MAPMODE == 0) {
        for (int k = lane; k < p.D; k += WAVE) S_planes[(size_t)r * p.D + k] = Sl[k];
        if (depth_from_planes) {
            // similarities.py:199-227: points, first arg-max plane, distance to the camera
            float best = -INFINITY;
            int best_k = 0;
            for (int k = lane; k < p.D; k += WAVE) {
                float pt[3];
                plane_point(s, e, k, p.D, pt);
                reinterpret_cast<float4 *>(points)[(size_t)r * p.D + k] =
                    make_float4(pt[0], pt[1], pt[2], 1.0f);
                if (Sl[k] > best) {
                    best = Sl[k];
                    best_k = k;
                }
            }
            // first maximum: larger value wins, then smaller index
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o);
                const int ok = __shfl_xor(best_k, o);
                if (ob > best || (ob == best && ok < best_k)) {
                    best = ob;
                    best_k = ok;
                }
            }
            if (lane == 0) {
                float pt[3];
                plane_point(s, e, best_k, p.D, pt);
                float sum = 0.0f;
                for (int i = 0; i < 3; i++) {
                    const float d = pt[i] - cc[i];
                    sum += d * d;
                }
                depth_from_planes[r] = sqrtf(sum);
            }
        }
        return;
    }

    const int count = min(uniform(rvc[r]), p.M);
    const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
    float *out = S_voxel + (size_t)r * p.M;
    // MAPMODE 2 = the resident path: value-only divisions through the hardware reciprocal
    // (the plane index walk inside stays IEEE); MAPMODE 1 = K6 / K11, reference arithmetic
    const float srsum =
        map_planes_to_voxels<PACKED, MAPMODE == 2>(p, axes, vrow, count, s, e, Sl, vals, lane);
    if (MAPMODE == 1) {
        for (int i = lane; i < count; i += WAVE) out[i] = vals[i] / srsum;
    } else {
        float sum = 0.0f;
        const float rs = __builtin_amdgcn_rcpf(srsum);
        for (int i = lane; i < count; i += WAVE) {
            const float v = clampf(vals[i] * rs, (float)1e-5, (float)(1 - 1e-5));
            vals[i] = v;
            sum += v;
        }
        sum = __builtin_amdgcn_rcpf(wave_sum(sum));
        // streamed out, read again only by later kernels: keep it out of the L2 the feature
        // gathers live in
        for (int i = lane; i < count; i += WAVE) __builtin_nontemporal_store(vals[i] * sum, out + i);
    }
}

// ------------------------------------------------------------- K3: BP sweep
// One wavefront per ray, chunks of 64 voxels held in registers.
//   CLIP_IN: S is the raw voxel-space column (API mode) and is clipped + renormalised here;
//            otherwise it is the resident Sr.
//   msgs_in == nullptr means "all messages are zero" (first sweep): nothing is read.
// The messages go to msgs_out; adding them to the accumulator is the scatter kernels' job
// (from inside this kernel the lanes are consecutive voxels of ONE ray: 64 different cache
// lines per atomic instruction, 5x slower in total).
template <bool PACKED>
__device__ __forceinline__ int load_packed(const int32_t *__restrict__ row, int i) {
    if (PACKED) return row[i];
    return pack_voxel(row[3 * i], row[3 * i + 1], row[3 * i + 2]);
}
// Accumulator index of a voxel.  The reference's accumulators are [gx][gy][gz] arrays
// (mrf_bp.cu:3-10) and the K1-K4 entry points keep that.  The resident-scene path (BRICK)
// stores them as 4x4x4 bricks, [gx/4][gy/4][gz/4][4][4][4]: a ray steps through ~4 voxels of
// a brick in a row, so the 64 gathers of a wavefront instruction fall into ~16 cache lines
// instead of 64 when the ray does not travel along z.  Measured: the gather is the largest
// single cost of k_bp (no gather: -37 %); bricks take half of it back.
template <bool BRICK>
__device__ __forceinline__ int lin_xyz(const Params &p, int x, int y, int z) {
    if (BRICK)
        return ((((x >> 2) * p.nby + (y >> 2)) * p.nbz + (z >> 2)) << 6) | ((x & 3) << 4) |
               ((y & 3) << 2) | (z & 3);
    return (x * p.gy + y) * p.gz + z;
}
template <bool BRICK>
__device__ __forceinline__ int lin_of(const Params &p, int v) {
    return lin_xyz<BRICK>(p, v >> 20, (v >> 10) & 1023, v & 1023);
}
__device__ __forceinline__ float gather_acc(const float *__restrict__ acc, int lin) {
    return acc[lin];
}

// One wavefront per ray.  The kernels are instantiated for the launch's M (NCH = chunks of 64
// voxels) but a ray runs the body specialised for ITS chunk count (a uniform branch): the
// mean ray of config 2 has 137 voxels of M = 384, and every chunk a body is compiled for
// costs instructions whether or not the ray reaches it -- measured, k_bp sits at its
// VALU-issue floor (SQ_INSTS_VALU x 4 cycles), so instructions are what there is to save.
// (Also measured and dropped: walking several rays per wavefront with the next ray's rows in
// flight -- no gain, the limit was never the dependent round trips.)
template <int NCH>
struct RayRows {
    float sv[NCH], mv[NCH];
    int pk[NCH];
};
// the first `count` entries of the ray's column, voxel list and (optionally) messages
template <int NCH, bool PACKED>
__device__ __forceinline__ void load_rows(const Params &p, RayRows<NCH> &R,
                                          const float *__restrict__ S,
                                          const int32_t *__restrict__ vox, const float *msgs, int r,
                                          int count, int lane) {
    const float *Srow = S + (size_t)r * p.M;
    const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
    const float *mrow = msgs ? msgs + (size_t)r * p.M : nullptr;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const int i = ch * WAVE + lane;
        R.sv[ch] = 0.0f; R.mv[ch] = 0.0f; R.pk[ch] = 0;
        if (ch * WAVE < count && i < count) {
            R.sv[ch] = Srow[i];
            R.pk[ch] = load_packed<PACKED>(vrow, i);
            if (mrow) R.mv[ch] = mrow[i];
        }
    }
}
// clip to [1e-5, 1-1e-5] and renormalise over the count (mrf_bp.cu:103-111)
template <int NCH, bool CLIP_IN>
__device__ __forceinline__ void clip_renorm_rows(float (&sv)[NCH], int count, int lane) {
    if (!CLIP_IN) return;          // resident columns are stored clipped + renormalised
    float ssum = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const int i = ch * WAVE + lane;
        float v = 0.0f;
        if (i < count) v = clampf(sv[ch], (float)1e-5, (float)(1 - 1e-5));
        sv[ch] = v;
        ssum += v;
    }
    ssum = wave_sum(ssum);
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) sv[ch] = sv[ch] / ssum;
}
// uniform dispatch on the ray's chunk count: BODY(NB) for the smallest compiled NB >= nch
#define RN_DISPATCH_CHUNKS(NCH, nch, BODY)                  \
    do {                                                    \
        if (NCH >= 1 && nch <= 1) { BODY(1); }              \
        else if (NCH >= 2 && nch <= 2) { BODY(2); }         \
        else if (NCH >= 3 && nch <= 3) { BODY(3); }         \
        else if (NCH >= 4 && nch <= 4) { BODY(4); }         \
        else if (NCH >= 6 && nch <= 6) { BODY(6); }         \
        else if (NCH >= 8 && nch <= 8) { BODY(8); }         \
        else if (NCH >= 12 && nch <= 12) { BODY(12); }      \
        else { BODY(NCH); }                                 \
    } while (0)

// one BP sweep of one ray with NB >= ceil(count / 64) chunks (mrf_bp.cu:88-177)
template <int NB, bool PACKED, bool CLIP_IN>
__device__ __forceinline__ void bp_ray(const Params &p, int r, int count, int lane,
                                       const float *__restrict__ S,
                                       const int32_t *__restrict__ vox,
                                       const float *__restrict__ acc_in, const float *msgs_in,
                                       float *msgs_out, bool uniform_acc) {
    RayRows<NB> cur;
    load_rows<NB, PACKED>(p, cur, S, vox, msgs_in, r, count, lane);
    // accumulator gather (depends on the voxel rows).  uniform_acc: every voxel holds
    // acc_in[0] (the first iteration starts from the prior everywhere) -- nothing to gather,
    // and with zero messages on top the occupancy is one constant for the whole sweep.
    float av[NB];
    const float a0 = uniform_acc ? acc_in[0] : 0.0f;
#pragma unroll
    for (int ch = 0; ch < NB; ch++) {
        const int i = ch * WAVE + lane;
        av[ch] = a0;
        if (!uniform_acc && ch * WAVE < count && i < count)
            av[ch] = gather_acc(acc_in, lin_of<PACKED>(p, cur.pk[ch]));
    }
    const bool const_o = uniform_acc && msgs_in == nullptr;
    const float o_const = occupancy_to_ray(a0, 0.0f);
    float *mout_row = msgs_out + (size_t)r * p.M;
    clip_renorm_rows<NB, CLIP_IN>(cur.sv, count, lane);

    // pass A: occupancy, exclusive cumprod T, w = o*T*s, exclusive cumsum C
    float ov[NB], tsv[NB], cex[NB], wv[NB];
    float carryT = 1.0f, carryC = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NB; ch++) {
        ov[ch] = 0.0f; tsv[ch] = 0.0f; cex[ch] = 0.0f; wv[ch] = 0.0f;
        if (ch * WAVE < count) {
            const int i = ch * WAVE + lane;
            const bool valid = i < count;
            float o = o_const;
            if (!const_o) o = occupancy_to_ray(av[ch], cur.mv[ch]);
            if (!valid) o = 0.0f;
            const float incl = wave_scan_mul(valid ? 1.0f - o : 1.0f);
            const float T = carryT * wave_shift1(incl, 1.0f);
            carryT = carryT * lane63(incl);
            const float ts = T * cur.sv[ch];
            const float w = valid ? o * ts : 0.0f;
            const float inclC = wave_scan_add(w);
            cex[ch] = carryC + wave_shift1(inclC, 0.0f);
            carryC = carryC + lane63(inclC);
            ov[ch] = o;
            tsv[ch] = ts;
            wv[ch] = w;
        }
    }
    // (cumsum1 - cumsum2) of mrf_bp.cu:157 is the suffix sum  sum_{j>i} w_j.  The reference
    // forms it as a difference of two running sums, which is exact-or-zero only because both
    // are the SAME sequential sum; with wave scans that difference could go negative by an
    // ulp (log of a negative number -> NaN), so the suffix is scanned directly.  It is
    // non-negative by construction and free of the reference's cancellation.
    float suf[NB];
    {
        float carryS = 0.0f;
#pragma unroll
        for (int ch = NB - 1; ch >= 0; ch--) {
            suf[ch] = 0.0f;
            if (ch * WAVE < count) {
                float tot;
                suf[ch] = carryS + wave_suffix_excl(wv[ch], lane, tot);
                carryS = carryS + tot;
            }
        }
    }
    // pass B: messages (mrf_bp.cu:136-167); the scatter (:170-176) is a kernel of its own
#pragma unroll
    for (int ch = 0; ch < NB; ch++) {
        if (ch * WAVE < count) {
            const int i = ch * WAVE + lane;
            if (i < count) {
                // log p - log(1 - p) with p = pos / (pos + neg) (mrf_bp.cu:160-165) is
                // log pos - log neg: no normalisation, and no cancellation in 1 - p
                const float pos = cex[ch] + tsv[ch];
                const float neg = cex[ch] + bp_div(suf[ch], 1.0f - ov[ch]);
                const float m = bp_log(pos) - bp_log(neg);
                mout_row[i] = m;
            }
        }
    }
}

template <int NCH, bool PACKED, bool CLIP_IN>
__global__ __launch_bounds__(BLOCK) void k_bp(Params p, int n, const float *__restrict__ S,
                                              const int32_t *__restrict__ vox,
                                              const int32_t *__restrict__ rvc,
                                              const float *__restrict__ acc_in,
                                              const float *msgs_in, float *msgs_out,
                                              int uniform_acc) {
    int lane;
    const int r = ray_of_wave(n, lane);
    if (r < 0) return;
    const int count = min(uniform(rvc[r]), p.M);
    if (count <= 1) return;   // mrf_np.py:300: such rays send nothing
    const int nch = (count + WAVE - 1) / WAVE;
#define RN_BP_BODY(NB) \
    bp_ray<NB, PACKED, CLIP_IN>(p, r, count, lane, S, vox, acc_in, msgs_in, msgs_out, uniform_acc != 0)
    RN_DISPATCH_CHUNKS(NCH, nch, RN_BP_BODY);
#undef RN_BP_BODY
}

// ------------------------------------------------- accumulator scatter, slab-ordered
// mrf_bp.cu:170-176 (acc_out[voxel] += message) for rows in ray-index order: the atomics of
// a tile of 64 CONSECUTIVE rays go through LDS and are issued in order of
// the voxels' coordinate along the tile's dominant travel axis instead of in step order.
// The 64 rays of a tile are neighbouring pixels of one image column: they lie in one
// plane through the camera, so inside one slab of the dominant axis their voxels share
// (nearly) the same column of the grid and differ along z -- consecutive floats.  One
// instruction then touches a handful of cache lines instead of 64 (an L2 float atomic
// costs one request per line: 21 G/s scattered vs 324 G/s coalesced, tools/atomic_bench.hip).
// Any ray order is CORRECT (every element is emitted exactly once; the flush loop takes
// what an unexpected ordering left behind); coherence only buys speed.
#ifdef RN_SCATTER_STATS
__device__ unsigned long long g_scatter_stats[8];   // rounds, emitting lanes, tails, 64B segments, chunks
#endif
#ifndef RN_SLAB_STEPS
#define RN_SLAB_STEPS 32
#endif
constexpr int SLAB_STEPS = RN_SLAB_STEPS;     // steps of a tile: 16, 32 or 64
constexpr int SLAB_PAD = SLAB_STEPS + 1;
template <bool PACKED>
__global__ __launch_bounds__(WAVE) void k_scatter_slab(Params p, int n,
                                                       const float *__restrict__ msgs,
                                                       const int32_t *__restrict__ vox,
                                                       const int32_t *__restrict__ rvc,
                                                       float *acc_out) {
    __shared__ float tile_m[WAVE * SLAB_PAD];
    __shared__ int32_t tile_v[WAVE * SLAB_PAD];
    const int lane = threadIdx.x;
    // one wavefront per (64-ray tile, 32-step chunk): short independent waves keep the
    // launch's tail and the fixed cost on small shards (8-GPU runs) low
    const int nchunks = (p.M + SLAB_STEPS - 1) / SLAB_STEPS;
    const int lb = xcd_block(blockIdx.x, gridDim.x);
    const int r0 = (lb / nchunks) * WAVE;
    const int base = (lb % nchunks) * SLAB_STEPS;
    int cnt = 0;
    if (r0 + lane < n) {
        cnt = min(rvc[r0 + lane], p.M);
        if (cnt <= 1) cnt = 0;        // such rays send no message (mrf_np.py:300)
    }
    int maxc = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
    maxc = uniform(maxc);
    if (base >= maxc) return;

    {
        if (PACKED && (p.M % SLAB_STEPS) == 0) {
            // rows in: 8 rays per instruction, each lane 4 consecutive steps (16 B); all 16
            // loads of the chunk are in flight before the first LDS write
            constexpr int LPR = SLAB_STEPS / 4;      // lanes per row
            constexpr int RPI = WAVE / LPR;          // rows per instruction
            constexpr int NI = WAVE / RPI;           // instructions per array
            const int sub = lane / LPR, q = lane % LPR;
            float4 mv[NI];
            int4 vv[NI];
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const int row = RPI * j + sub;
                const int c = __shfl(cnt, row);      // all lanes take part in the shuffle
                mv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                vv[j] = make_int4(0, 0, 0, 0);
                if (r0 + row < n && base < c) {
                    const size_t off = (size_t)(r0 + row) * p.M + base + 4 * q;
                    mv[j] = *reinterpret_cast<const float4 *>(msgs + off);
                    vv[j] = *reinterpret_cast<const int4 *>(vox + off);
                }
            }
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const int a = (RPI * j + sub) * SLAB_PAD + 4 * q;
                tile_m[a] = mv[j].x; tile_m[a + 1] = mv[j].y;
                tile_m[a + 2] = mv[j].z; tile_m[a + 3] = mv[j].w;
                tile_v[a] = vv[j].x; tile_v[a + 1] = vv[j].y;
                tile_v[a + 2] = vv[j].z; tile_v[a + 3] = vv[j].w;
            }
        } else {
            // generic layout: two rays per instruction, 32 steps each
#pragma unroll 4
            for (int j = 0; j < WAVE; j += WAVE / SLAB_STEPS) {
                const int row = j + lane / SLAB_STEPS;
                const int col = lane % SLAB_STEPS;
                const int c = __shfl(cnt, row);
                float m = 0.0f;
                int32_t v = 0;
                if (base + col < c) {
                    const size_t off = (size_t)(r0 + row) * p.M + base + col;
                    m = msgs[off];
                    if (PACKED) {
                        v = vox[off];
                    } else {
                        const int32_t *t = vox + off * 3;
                        v = pack_voxel(t[0], t[1], t[2]);
                    }
                }
                tile_m[row * SLAB_PAD + col] = m;
                tile_v[row * SLAB_PAD + col] = v;
            }
        }
        wave_sync();
#ifdef RN_SCATTER_STATS
        if (lane == 0) atomicAdd(&g_scatter_stats[4], 1ull);
#endif

        const int nvalid = min(max(cnt - base, 0), SLAB_STEPS);
        int cursor = 0;
        int32_t vcur = nvalid > 0 ? tile_v[lane * SLAB_PAD] : 0;
        // dominant axis / direction of the chunk: sum over rays of (last voxel - first voxel)
        int shift = 0, flip = 0;
        {
            int dx = 0, dy = 0, dz = 0;
            if (nvalid > 1) {
                const int32_t vl = tile_v[lane * SLAB_PAD + nvalid - 1];
                dx = (vl >> 20) - (vcur >> 20);
                dy = ((vl >> 10) & 1023) - ((vcur >> 10) & 1023);
                dz = (vl & 1023) - (vcur & 1023);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                dx += __shfl_xor(dx, o);
                dy += __shfl_xor(dy, o);
                dz += __shfl_xor(dz, o);
            }
            const int ax = abs(dx), ay = abs(dy), az = abs(dz);
            if (ax >= ay && ax >= az) { shift = 20; flip = dx < 0; }
            else if (ay >= az) { shift = 10; flip = dy < 0; }
            else { shift = 0; flip = dz < 0; }
            shift = uniform(shift);
            flip = uniform(flip);
        }
        auto key_of = [&](int32_t v) {
            const int c = (v >> shift) & 1023;
            return flip ? 1023 - c : c;
        };
        // slab range of this chunk (first / last element of every ray; exact when the
        // rays move monotonically along the tile's axis, which is the normal case)
        int kmin = nvalid > 0 ? key_of(vcur) : 1 << 30;
        int kmax = nvalid > 0 ? key_of(tile_v[lane * SLAB_PAD + nvalid - 1]) : -1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmin = min(kmin, __shfl_xor(kmin, o));
            kmax = max(kmax, __shfl_xor(kmax, o));
        }
        kmin = uniform(kmin);
        kmax = uniform(kmax);
        for (int k = kmin; k <= kmax + 1; k++) {
            const int limit = (k > kmax) ? (1 << 30) : k;     // last round flushes everything
            while (true) {
                const bool emit = cursor < nvalid && key_of(vcur) <= limit;
                if (__ballot(emit) == 0) break;
                // Neighbouring rays usually sit in the SAME voxel (ray spacing < voxel size);
                // an instruction with duplicate addresses is serialised by the L2 (x6 for
                // pairs, tools/atomic_bench2.hip).  Runs of equal addresses in adjacent lanes
                // are therefore summed first (segmented scan inside rows of 16 lanes) and only
                // the last lane of each run issues the atomic.
                float val = 0.0f;
                int lin = -2 - lane;                 // unique: a non-emitting lane is its own run
                if (emit) {
                    val = tile_m[lane * SLAB_PAD + cursor];
                    lin = lin_of<PACKED>(p, vcur);
                }
                int head = dpp_i<0x111, 0xf>(0x7fffffff, lin) != lin;     // row start: head
#define RN_SEG_STEP(CTRL)                                             \
    {                                                                 \
        const float vp = dpp_f<CTRL, 0xf>(0.0f, val);                 \
        const int fp = dpp_i<CTRL, 0xf>(1, head);                     \
        if (!head) val += vp;                                         \
        head |= fp;                                                   \
    }
                RN_SEG_STEP(0x111) RN_SEG_STEP(0x112) RN_SEG_STEP(0x114) RN_SEG_STEP(0x118)
#undef RN_SEG_STEP
                const bool tail = dpp_i<0x101, 0xf>(0x7ffffffe, lin) != lin;   // row_shl:1
#ifdef RN_SCATTER_STATS
                {
                    const bool t = emit && tail;
                    const unsigned long long bt = __ballot(t);
                    // distinct 64-byte segments among the issuing lanes (exact count)
                    int seg = t ? (lin >> 4) : -1;
                    int distinct = 0;
                    unsigned long long left = bt;
                    while (left) {
                        const int l = __builtin_ctzll(left);
                        const int sv = __shfl(seg, l);
                        const unsigned long long same = __ballot(t && seg == sv);
                        left &= ~same;
                        distinct++;
                    }
                    if (lane == 0) {
                        atomicAdd(&g_scatter_stats[0], 1ull);
                        atomicAdd(&g_scatter_stats[1], (unsigned long long)__builtin_popcountll(__ballot(emit)));
                        atomicAdd(&g_scatter_stats[2], (unsigned long long)__builtin_popcountll(bt));
                        atomicAdd(&g_scatter_stats[3], (unsigned long long)distinct);
                    }
                }
#endif
                if (emit) {
#ifdef RN_SCATTER_NOATOMIC       // timing experiment only: everything but the atomic
                    asm volatile("" ::"v"(lin), "v"(val));
#else
                    if (tail)
                        __hip_atomic_fetch_add(acc_out + lin, val, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
#endif
                    cursor++;
                    if (cursor < nvalid) vcur = tile_v[lane * SLAB_PAD + cursor];
                }
            }
        }
        wave_sync();
    }
}

// ------------------------------------------------- accumulator scatter, LDS box
// A tile of BOX_RAYS neighbouring rays x BOX_STEPS steps covers a compact block of the grid
// in which every voxel is hit by ~10-20 of the tile's rays (ray spacing << voxel size).
// The tile's messages are therefore summed in a dense LDS image of their bounding box
// in DOUBLE (measured, tools/lds_atomic_bench.hip: ds_add_f64 runs at ~1.7 T lane-ops/s,
// ds_add_f32 at 0.2 T/s whatever the addresses; the sums also become order-independent to
// ~1e-16, i.e. the accumulator is reproducible run to run) and the box is flushed once, so
// there is one global atomic per DISTINCT voxel of the tile instead of one per (ray, voxel).
// One workgroup walks a tile chunk by chunk: all of a chunk's (message, voxel) pairs sit in
// registers (BOX_NB per thread, one round trip), the box is the exact bounding box of those
// voxels -- no assumption on the lists -- and a chunk whose box exceeds the LDS budget (rows
// that are not patch-ordered) goes straight to the global atomics: always correct.
// Tile shapes (rays x steps) and LDS capacity (voxels): 128 x 32 reads 128 B of every row per
// round trip (whole cache lines) and is the default with 4096 voxels (32 KB, 5 workgroups per
// CU); scenes whose bundles do not fit (fine grids, oblique views) first get 6144 voxels, then
// 256 x 16 tiles -- the kernel counts the chunks that overflowed and the launcher looks at
// the previous launches' count (rn_ctx::box_*).
__device__ __forceinline__ int wave_reduce_max(int x) { return lane63i(wave_scan_max(x)); }
__device__ __forceinline__ int wave_reduce_min(int x) { return ~wave_reduce_max(~x); }
// What the scatter sums in.  Default: doubles in LDS, float atomics on the accumulator (the
// reference's float atomicAdd, mrf_bp.cu:170-176).  FIXED: every message becomes a signed
// 31.32 fixed-point integer first and all sums -- LDS, accumulator, and the all-reduce across
// GPUs -- are 64-bit integer additions: associative, so the accumulator is bit-identical from
// run to run and for any number of ranks (SURVEY.md 8e "deterministic mode").
template <bool FIXED>
struct AccSum {
    typedef double box_t;
    typedef float acc_t;
    static __device__ __forceinline__ box_t from_msg(float m) { return (double)m; }
    static __device__ __forceinline__ void direct(acc_t *acc, int lin, float m) {
        __hip_atomic_fetch_add(acc + lin, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ void flush(acc_t *acc, int lin, box_t v) {
        const float f = (float)v;
        if (f != 0.0f) __hip_atomic_fetch_add(acc + lin, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
__device__ __forceinline__ unsigned long long msg_to_fixed(float m) {
    // m * 2^32 is exact in double; saturate what does not fit (non-finite messages)
    const double x = fmin(fmax((double)m * 4294967296.0, -9.2e18), 9.2e18);
    return (unsigned long long)__double2ll_rn(x == x ? x : 0.0);
}
template <>
struct AccSum<true> {
    typedef unsigned long long box_t;
    typedef unsigned long long acc_t;
    static __device__ __forceinline__ box_t from_msg(float m) { return msg_to_fixed(m); }
    static __device__ __forceinline__ void direct(acc_t *acc, int lin, float m) {
        __hip_atomic_fetch_add(acc + lin, msg_to_fixed(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ void flush(acc_t *acc, int lin, box_t v) {
        if (v != 0ull) __hip_atomic_fetch_add(acc + lin, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

template <bool PACKED, int BOX_RAYS, int BOX_STEPS, bool FIXED = false>
__global__ __launch_bounds__(BLOCK) void k_scatter_box(Params p, int n,
                                                       const float *__restrict__ msgs,
                                                       const int32_t *__restrict__ vox,
                                                       const int32_t *__restrict__ rvc,
                                                       void *acc_out_raw,
                                                       unsigned *overflow_stats, int BOX_CAP) {
    typedef AccSum<FIXED> Sum;
    typename Sum::acc_t *acc_out = static_cast<typename Sum::acc_t *>(acc_out_raw);
    constexpr int BOX_NB = BOX_RAYS * BOX_STEPS / BLOCK;     // pairs per thread and chunk
    // BOX_CAP voxels (8 bytes each) of dynamic LDS: the launcher trades capacity for occupancy
    extern __shared__ __attribute__((aligned(16))) unsigned long long box_raw[];
    typename Sum::box_t *box = reinterpret_cast<typename Sum::box_t *>(box_raw);
    __shared__ int red[2][6 * WAVES_PER_BLOCK];
    __shared__ int red_cnt[WAVES_PER_BLOCK];
    __shared__ int cnts[BOX_RAYS];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
    const int r0 = xcd_block(blockIdx.x, gridDim.x) * BOX_RAYS;
    // a wavefront instruction covers RPI rays x BOX_STEPS steps; thread (sub, col) of wave w
    // owns step col of the rays w*RPI + sub + k*STRIDE
    constexpr int RPI = WAVE / BOX_STEPS;
    constexpr int STRIDE = WAVES_PER_BLOCK * RPI;
    const int sub = lane / BOX_STEPS, col = lane % BOX_STEPS;
    const int j0 = w * RPI + sub;
    int maxc = 0;
    for (int j = tid; j < BOX_RAYS; j += BLOCK) {
        int c = r0 + j < n ? min(rvc[r0 + j], p.M) : 0;
        if (c <= 1) c = 0;            // such rays send no message (mrf_np.py:300)
        cnts[j] = c;
        maxc = max(maxc, c);
    }
    maxc = wave_reduce_max(maxc);
    if (lane == 0) red_cnt[w] = maxc;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < WAVES_PER_BLOCK; k++) maxc = max(maxc, red_cnt[k]);
    maxc = uniform(maxc);

    int it = 0;
    // per-thread partial bounding box -> the workgroup's (uniform)
    auto block_bbox = [&](int &lo0, int &lo1, int &lo2, int &hi0, int &hi1, int &hi2) {
        lo0 = wave_reduce_min(lo0); lo1 = wave_reduce_min(lo1); lo2 = wave_reduce_min(lo2);
        hi0 = wave_reduce_max(hi0); hi1 = wave_reduce_max(hi1); hi2 = wave_reduce_max(hi2);
        int *rd = red[it++ & 1];
        if (lane == 0) {
            rd[w] = lo0; rd[WAVES_PER_BLOCK + w] = lo1; rd[2 * WAVES_PER_BLOCK + w] = lo2;
            rd[3 * WAVES_PER_BLOCK + w] = hi0; rd[4 * WAVES_PER_BLOCK + w] = hi1;
            rd[5 * WAVES_PER_BLOCK + w] = hi2;
        }
        __syncthreads();              // also: every thread has finished the previous flush
#pragma unroll
        for (int k = 0; k < WAVES_PER_BLOCK; k++) {
            lo0 = min(lo0, rd[k]); lo1 = min(lo1, rd[WAVES_PER_BLOCK + k]);
            lo2 = min(lo2, rd[2 * WAVES_PER_BLOCK + k]);
            hi0 = max(hi0, rd[3 * WAVES_PER_BLOCK + k]);
            hi1 = max(hi1, rd[4 * WAVES_PER_BLOCK + k]);
            hi2 = max(hi2, rd[5 * WAVES_PER_BLOCK + k]);
        }
        lo0 = uniform(lo0); lo1 = uniform(lo1); lo2 = uniform(lo2);
        hi0 = uniform(hi0); hi1 = uniform(hi1); hi2 = uniform(hi2);
    };
    // box -> accumulator, z fastest; (i0, i1, i2) advance by BLOCK elements without divisions
    auto flush_box = [&](int lo0, int lo1, int lo2, int d0, int d1, int d2) {
        const int V = d0 * d1 * d2;
        int i2 = tid % d2, t = tid / d2;
        int i1 = t % d1, i0 = t / d1;
        const int sz = BLOCK % d2, ty = BLOCK / d2;
        const int sy = ty % d1, sx = ty / d1;
        for (int i = tid; i < V; i += BLOCK) {
            Sum::flush(acc_out, lin_xyz<PACKED>(p, lo0 + i0, lo1 + i1, lo2 + i2), box[i]);
            i2 += sz;
            if (i2 >= d2) { i2 -= d2; i1++; }
            i1 += sy;
            if (i1 >= d1) { i1 -= d1; i0++; }
            i0 += sx;
        }
    };
    // gridDim.y workgroups share a tile, taking every gridDim.y-th chunk: small launches (a
    // rank of a multi-GPU run) still fill the chip
    for (int s0 = blockIdx.y * BOX_STEPS; s0 < maxc; s0 += gridDim.y * BOX_STEPS) {
        const int st = s0 + col;
        // ---- this chunk's pairs into registers, and their bounding box
        float m[BOX_NB];
        int v[BOX_NB];
        unsigned okmask = 0;
#pragma unroll
        for (int k = 0; k < BOX_NB; k++) {
            const bool ok = st < cnts[j0 + k * STRIDE];
            okmask |= (unsigned)ok << k;
            // rows of padding / short rays are read at the tile's first row: valid memory
            const int rr = ok ? r0 + j0 + k * STRIDE : r0, ss = ok ? st : 0;
            m[k] = msgs[(size_t)rr * p.M + ss];
            v[k] = load_packed<PACKED>(vox + (size_t)rr * p.M * (PACKED ? 1 : 3), ss);
        }
        int lo0 = 1 << 30, lo1 = 1 << 30, lo2 = 1 << 30, hi0 = -1, hi1 = -1, hi2 = -1;
#pragma unroll
        for (int k = 0; k < BOX_NB; k++) {
            if (okmask >> k & 1) {
                const int x = v[k] >> 20, y = (v[k] >> 10) & 1023, z = v[k] & 1023;
                lo0 = min(lo0, x); hi0 = max(hi0, x);
                lo1 = min(lo1, y); hi1 = max(hi1, y);
                lo2 = min(lo2, z); hi2 = max(hi2, z);
            }
        }
        block_bbox(lo0, lo1, lo2, hi0, hi1, hi2);
        if (hi0 < 0) continue;        // (cannot happen below maxc; uniform anyway)
        const int d0 = hi0 - lo0 + 1, d1 = hi1 - lo1 + 1, d2 = hi2 - lo2 + 1;
        const int V = d0 * d1 * d2;
#ifdef RN_SCATTER_STATS
        if (tid == 0) {
            atomicAdd(&g_scatter_stats[0], 1ull);
            atomicAdd(&g_scatter_stats[1], V <= BOX_CAP ? 1ull : 0ull);
            atomicAdd(&g_scatter_stats[2], (unsigned long long)V);
        }
#endif
        if (V <= BOX_CAP) {
            for (int i = tid; i < V; i += BLOCK) box[i] = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BOX_NB; k++)
                if (okmask >> k & 1) {
                    const int x = v[k] >> 20, y = (v[k] >> 10) & 1023, z = v[k] & 1023;
                    __hip_atomic_fetch_add(box + ((x - lo0) * d1 + (y - lo1)) * d2 + (z - lo2),
                                           Sum::from_msg(m[k]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            __syncthreads();
            flush_box(lo0, lo1, lo2, d0, d1, d2);
            continue;
        }
        if (tid == 0 && overflow_stats) atomicAdd(overflow_stats + 1, 1u);
        // ---- too big for LDS (rows that are not patch-ordered, very oblique bundles): the
        // chunk again in quarters, pairs re-read (L2-hot) so that this rare path costs the
        // common one no registers; a quarter that still does not fit takes the direct atomics
#pragma unroll 1
        for (int ca = 0; ca < BOX_STEPS; ca += BOX_STEPS / 4) {
            const bool mine = col >= ca && col < ca + BOX_STEPS / 4;
            lo0 = lo1 = lo2 = 1 << 30;
            hi0 = hi1 = hi2 = -1;
#pragma unroll 1
            for (int k = 0; k < BOX_NB; k++) {
                const int j = j0 + k * STRIDE;
                if (mine && st < cnts[j]) {
                    const int pv = load_packed<PACKED>(
                        vox + (size_t)(r0 + j) * p.M * (PACKED ? 1 : 3), st);
                    const int x = pv >> 20, y = (pv >> 10) & 1023, z = pv & 1023;
                    lo0 = min(lo0, x); hi0 = max(hi0, x);
                    lo1 = min(lo1, y); hi1 = max(hi1, y);
                    lo2 = min(lo2, z); hi2 = max(hi2, z);
                }
            }
            block_bbox(lo0, lo1, lo2, hi0, hi1, hi2);
            if (hi0 < 0) continue;
            const int e0 = hi0 - lo0 + 1, e1 = hi1 - lo1 + 1, e2 = hi2 - lo2 + 1;
            const bool fits = e0 * e1 * e2 <= BOX_CAP;
            if (fits) {
                for (int i = tid; i < e0 * e1 * e2; i += BLOCK) box[i] = 0;
                __syncthreads();
            }
#pragma unroll 1
            for (int k = 0; k < BOX_NB; k++) {
                const int j = j0 + k * STRIDE;
                if (mine && st < cnts[j]) {
                    const size_t row = (size_t)(r0 + j) * p.M;
                    const float mm = msgs[row + st];
                    const int pv = load_packed<PACKED>(vox + row * (PACKED ? 1 : 3), st);
                    const int x = pv >> 20, y = (pv >> 10) & 1023, z = pv & 1023;
                    if (fits)
                        __hip_atomic_fetch_add(box + ((x - lo0) * e1 + (y - lo1)) * e2 + (z - lo2),
                                               Sum::from_msg(mm), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                        Sum::direct(acc_out, lin_of<PACKED>(p, pv), mm);
                }
            }
            if (fits) {
                __syncthreads();
                flush_box(lo0, lo1, lo2, e0, e1, e2);
            }
        }
    }
    if (tid == 0 && overflow_stats)
        atomicAdd(overflow_stats,
                  (unsigned)((maxc + BOX_STEPS - 1) / BOX_STEPS + gridDim.y - 1 - blockIdx.y) /
                      gridDim.y);
}

// deterministic scatter for rows the box kernel is not used on: every (ray, voxel) pair adds
// its fixed-point message straight to the 64-bit accumulator (slow, order-independent)
template <bool PACKED>
__global__ __launch_bounds__(BLOCK) void k_scatter_direct_fixed(Params p, int n,
                                                                const float *__restrict__ msgs,
                                                                const int32_t *__restrict__ vox,
                                                                const int32_t *__restrict__ rvc,
                                                                unsigned long long *acc_out) {
    int lane;
    const int r = ray_of_wave(n, lane);
    if (r < 0) return;
    const int count = min(uniform(rvc[r]), p.M);
    if (count <= 1) return;
    const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
    for (int i = lane; i < count; i += WAVE)
        AccSum<true>::direct(acc_out, lin_of<PACKED>(p, load_packed<PACKED>(vrow, i)),
                             msgs[(size_t)r * p.M + i]);
}
// acc_out = prior + fixed-point partial (2^-32 units); the partial is zeroed for the next sweep
__global__ void k_acc_combine_fixed(unsigned long long *part, int64_t G, float prior, float *out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < G;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double s = (double)(long long)part[i] * (1.0 / 4294967296.0);
        part[i] = 0ull;
        out[i] = prior + (float)s;
    }
}

// ------------------------------------------------- K4 / K2 tail: depth estimate
// Writes the distribution (if S_new) and/or the arg-max depth (if depth_map).
// depth distribution of one ray with NB >= ceil(count / 64) chunks (mrf_bp.cu:37-86);
// returns the lane's best (value, index) for the arg-max
template <int NB, bool PACKED, bool CLIP_IN>
__device__ __forceinline__ void depth_ray(const Params &p, int r, int count, int lane,
                                          const float *__restrict__ S,
                                          const int32_t *__restrict__ vox,
                                          const float *__restrict__ acc,
                                          const float *__restrict__ msgs, float *S_new, float &best,
                                          int &best_i) {
    RayRows<NB> cur;
    load_rows<NB, PACKED>(p, cur, S, vox, msgs, r, count, lane);
    float av[NB];
#pragma unroll
    for (int ch = 0; ch < NB; ch++) {
        const int i = ch * WAVE + lane;
        av[ch] = 0.0f;
        if (ch * WAVE < count && i < count) av[ch] = gather_acc(acc, lin_of<PACKED>(p, cur.pk[ch]));
    }
    clip_renorm_rows<NB, CLIP_IN>(cur.sv, count, lane);
    float wv[NB];
    float carryT = 1.0f, wsum = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NB; ch++) {
        wv[ch] = 0.0f;
        if (ch * WAVE < count) {
            const int i = ch * WAVE + lane;
            const bool valid = i < count;
            const float o = valid ? occupancy_to_ray(av[ch], cur.mv[ch]) : 0.0f;
            const float incl = wave_scan_mul(valid ? 1.0f - o : 1.0f);
            const float T = carryT * wave_shift1(incl, 1.0f);
            carryT = carryT * lane63(incl);
            wv[ch] = valid ? o * T * cur.sv[ch] : 0.0f;
            wsum += wv[ch];
        }
    }
    wsum = wave_sum(wsum);
#pragma unroll
    for (int ch = 0; ch < NB; ch++) {
        const int i = ch * WAVE + lane;
        if (ch * WAVE < count && i < count) {
            const float d = bp_div(wv[ch], wsum);
            if (S_new) S_new[(size_t)r * p.M + i] = d;
            if (d > best) {   // ascending i per lane: keeps the first maximum
                best = d;
                best_i = i;
            }
        }
    }
}

template <int NCH, bool PACKED, bool CLIP_IN>
__global__ __launch_bounds__(BLOCK) void k_depth(Params p, int n, const float *S,
                                                 const int32_t *__restrict__ vox,
                                                 const int32_t *__restrict__ rvc,
                                                 const float *__restrict__ acc,
                                                 const float *__restrict__ msgs,
                                                 const float *__restrict__ axes,
                                                 const float *__restrict__ cc, float *S_new,
                                                 float *depth_map, int rays_per_center) {
    int lane;
    const int r = ray_of_wave(n, lane);
    if (r < 0) return;
    if (rays_per_center > 0 && cc) cc += 4 * (r / rays_per_center);
    const int count = min(uniform(rvc[r]), p.M);
    float best = -INFINITY;
    int best_i = 0;
    if (count > 1) {
        const int nch = (count + WAVE - 1) / WAVE;
#define RN_DE_BODY(NB) \
    depth_ray<NB, PACKED, CLIP_IN>(p, r, count, lane, S, vox, acc, msgs, S_new, best, best_i)
        RN_DISPATCH_CHUNKS(NCH, nch, RN_DE_BODY);
#undef RN_DE_BODY
    } else if (S_new) {
        // mrf_np.py:370-377: skipped rays keep an all-zero row (first `count` entries)
        for (int i = lane; i < count; i += WAVE) S_new[(size_t)r * p.M + i] = 0.0f;
    }
    if (!depth_map) return;
    // raynet_fp.py:193-226.  Entries beyond count are zero in the reference's zero-filled
    // buffer and every d_i > 0, so the arg-max lies in [0, count); for count <= 1 the row is
    // all zeros and index 0 wins.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(best_i, o);
        if (ob > best || (ob == best && oi < best_i)) {
            best = ob;
            best_i = oi;
        }
    }
    if (lane == 0) {
        const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
        int x = 0, y = 0, z = 0;
        if (count > 0) load_voxel<PACKED>(vrow, count > 1 ? best_i : 0, x, y, z);
        const float pt[3] = {axes[x], axes[p.gx + y], axes[p.gx + p.gy + z]};
        float sum = 0.0f;
        for (int i = 0; i < 3; i++) {
            const float d = pt[i] - cc[i];
            sum += d * d;
        }
        depth_map[r] = sqrtf(sum);
    }
}

// K12 tail: arg-max of the mapped (not BP-refined) voxel column -> depth
template <bool PACKED>
__global__ __launch_bounds__(BLOCK) void k_argmax_depth(Params p, int n, const float *S_voxel,
                                                        const int32_t *__restrict__ vox,
                                                        const int32_t *__restrict__ rvc,
                                                        const float *__restrict__ axes,
                                                        const float *__restrict__ cc,
                                                        float *depth_map) {
    int lane;
    const int r = ray_of_wave(n, lane);
    if (r < 0) return;
    const int count = min(uniform(rvc[r]), p.M);
    const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
    float best = -INFINITY;
    int best_i = 0;
    for (int i = lane; i < count; i += WAVE) {
        const float v = S_voxel[(size_t)r * p.M + i];
        if (v > best) {
            best = v;
            best_i = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(best_i, o);
        if (ob > best || (ob == best && oi < best_i)) {
            best = ob;
            best_i = oi;
        }
    }
    if (lane == 0) {
        // a zero-filled tail (value 0) beats only a non-positive head; mapped
        // values are positive, so the winner is inside [0, count) when count > 0
        int x = 0, y = 0, z = 0;
        if (count > 0 && best > 0.0f) load_voxel<PACKED>(vrow, best_i, x, y, z);
        else if (count > 0) load_voxel<PACKED>(vrow, 0, x, y, z);
        const float pt[3] = {axes[x], axes[p.gx + y], axes[p.gx + p.gy + z]};
        float sum = 0.0f;
        for (int i = 0; i < 3; i++) {
            const float d = pt[i] - cc[i];
            sum += d * d;
        }
        depth_map[r] = sqrtf(sum);
    }
}

}  // namespace

// =============================================================== host side
struct rn_ctx {
    rn_config cfg;
    Params p;
    float *axes;          // device, gx+gy+gz
    bool have_axes;
    int scatter_mode;     // A/B knob RAYNET_HIP_SCATTER_MODE: -1 by row layout (default), 0 slab, 2 LDS box
    // LDS-box scatter: level in use (launch_bp), {chunks, overflowed chunks} of the previous
    // launches on the device and its pinned host mirror
    int box_level, box_level0;
    bool box_pin;         // RAYNET_HIP_BOX_PIN: stay at the starting level (A/B runs)
    unsigned *box_stats, *box_stats_host;
    hipEvent_t ev0, ev1;
    // per-launch profiling (rn_prof_begin / rn_prof_end)
    bool prof_on;
    int prof_cap, prof_n;
    hipEvent_t *prof_ev;      // 2 * prof_cap
    int32_t *prof_id, *prof_rays;
    char err[512];
};

// Brackets one kernel launch with two events on its stream when profiling is on.
struct ProfScope {
    rn_ctx *c;
    hipStream_t st;
    int slot;
    ProfScope(rn_ctx *ctx, int id, int n_rays, hipStream_t s) : c(ctx), st(s), slot(-1) {
        if (c->prof_on && c->prof_n < c->prof_cap) {
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
        if (e_ != hipSuccess)                                                          \
            return fail(ctx, RN_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int ray_blocks(int n) { return (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK; }
inline int thread_blocks(int n) { return (n + BLOCK - 1) / BLOCK; }
inline int fill_blocks(int64_t n) {
    int64_t b = (n + BLOCK - 1) / BLOCK;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
inline size_t sweep_lds(const Params &p) {
    return sizeof(float) * ((size_t)((p.gx + p.gy + p.gz + 3) & ~3) +
                            (size_t)WAVES_PER_BLOCK * (p.D + p.M));
}

FeatureViews stacked_views(const Params &p, const float *features) {
    FeatureViews fv;
    const size_t dim = (size_t)p.Hf * p.Wf * p.F;
    for (int v = 0; v < MAX_VIEWS; v++) fv.v[v] = v < p.N ? features + dim * v : nullptr;
    return fv;
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
};

template <int SIM, int NV, int LPS, int MAPMODE, bool PACKED>
void launch_sweep_t(rn_ctx *ctx, const SweepArgs &a, hipStream_t st) {
    ProfScope prof(ctx, RN_K_SWEEP_MAP, a.n * a.n_images, st);
    hipLaunchKernelGGL((k_sweep_map<SIM, NV, LPS, MAPMODE, PACKED>),
                       dim3(ray_blocks(a.n), a.n_images), dim3(BLOCK), sweep_lds(ctx->p), st,
                       ctx->p, a.n, a.ray_idxs, a.fv, a.P, a.P_inv, a.cc, a.starts, a.ends, a.S_in,
                       ctx->axes, a.vox, a.rvc, a.S_planes, a.S_voxel, a.depth_from_planes,
                       a.points, a.order, a.fv_table, a.cam_stride, a.rows_per_image, a.seg);
}

// pick the plane-sweep flavour: cooperative for F=32 and 2..9 views, generic otherwise
template <int MAPMODE, bool PACKED>
void launch_sweep(rn_ctx *ctx, const SweepArgs &a, bool have_features, hipStream_t st) {
    const Params &p = ctx->p;
    if (!have_features) {
        launch_sweep_t<0, 1, 8, MAPMODE, PACKED>(ctx, a, st);
        return;
    }
    if (p.F == 32 && getenv("RAYNET_HIP_GENERIC_SWEEP") == nullptr) {
        switch (p.N) {
#define RN_CASE(NV_)                                                  \
    case NV_:                                                         \
        launch_sweep_t<2, NV_, 8 / RN_SWEEP_V4, MAPMODE, PACKED>(ctx, a, st);       \
        return;
            RN_CASE(2) RN_CASE(3) RN_CASE(4) RN_CASE(5) RN_CASE(6) RN_CASE(7) RN_CASE(8) RN_CASE(9)
#undef RN_CASE
            default: break;
        }
    }
    launch_sweep_t<1, 1, 8, MAPMODE, PACKED>(ctx, a, st);
}

// workgroups per box-scatter tile (grid.y): enough of them for ~16 per CU
inline int box_split(int n, int tile_rays) {
    const int tiles = (n + tile_rays - 1) / tile_rays;
    return max(1, min(4, 4096 / max(tiles, 1)));
}

// One BP sweep: k_bp (messages) + the accumulator scatter that fits the row layout.
template <bool PACKED, bool CLIP_IN>
int launch_bp(rn_ctx *ctx, int n, const float *Sv, const int32_t *vox, const int32_t *rvc,
              const float *acc_in, const float *msgs_in, void *acc_out, float *msgs_out,
              hipStream_t st, bool patch_rows = false, bool fixed = false,
              bool uniform_acc = false) {
    const int nch = (ctx->p.M + WAVE - 1) / WAVE;
    {
        ProfScope prof(ctx, RN_K_BP, n, st);
#define RN_BP(NCH_)                                                                            \
    hipLaunchKernelGGL((k_bp<NCH_, PACKED, CLIP_IN>), dim3(ray_blocks(n)), dim3(BLOCK), 0, st, \
                       ctx->p, n, Sv, vox, rvc, acc_in, msgs_in, msgs_out, uniform_acc ? 1 : 0)
        if (nch <= 2) RN_BP(2);
        else if (nch <= 4) RN_BP(4);
        else if (nch <= 6) RN_BP(6);
        else if (nch <= 8) RN_BP(8);
        else if (nch <= 12) RN_BP(12);
        else RN_BP(16);
#undef RN_BP
    }
    RN_LAUNCH_CHECK(ctx);
    ProfScope prof(ctx, RN_K_SCATTER, n, st);
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
        const unsigned per = ctx->box_level < LAST - 1 ? 50u : 4u;
        if (!ctx->box_pin && ctx->box_level < LAST && ctx->box_stats_host[0] > 0 &&
            ctx->box_stats_host[1] * per > ctx->box_stats_host[0]) {
            ctx->box_level++;
            ctx->box_stats_host[0] = ctx->box_stats_host[1] = 0;
        }
        level = ctx->box_level;
    }
#define RN_BOX(RAYS, STEPS, FIXED_, CAP)                                                          \
    hipLaunchKernelGGL((k_scatter_box<PACKED, RAYS, STEPS, FIXED_>),                              \
                       dim3((n + RAYS - 1) / RAYS, box_split(n, RAYS)), dim3(BLOCK),              \
                       (CAP) * sizeof(double), st, ctx->p, n, msgs_out, vox, rvc, acc_out,        \
                       ctx->box_stats, CAP)
    if (level == 0) {
        if (fixed) RN_BOX(128, 32, true, 4096); else RN_BOX(128, 32, false, 4096);
    } else if (level == 1) {
        if (fixed) RN_BOX(256, 16, true, 6144); else RN_BOX(256, 16, false, 6144);
    } else if (fixed) {
        hipLaunchKernelGGL((k_scatter_direct_fixed<PACKED>), dim3(ray_blocks(n)), dim3(BLOCK), 0, st,
                           ctx->p, n, msgs_out, vox, rvc,
                           static_cast<unsigned long long *>(acc_out));
    } else {
        hipLaunchKernelGGL((k_scatter_slab<PACKED>),
                           dim3(((n + WAVE - 1) / WAVE) *
                                ((ctx->p.M + SLAB_STEPS - 1) / SLAB_STEPS)),
                           dim3(WAVE), 0, st, ctx->p, n, msgs_out, vox, rvc,
                           static_cast<float *>(acc_out));
    }
#undef RN_BOX
    if (level < LAST) {
        (void)hipMemcpyAsync(ctx->box_stats_host, ctx->box_stats, 2 * sizeof(unsigned),
                             hipMemcpyDeviceToHost, st);
        (void)hipMemsetAsync(ctx->box_stats, 0, 2 * sizeof(unsigned), st);
    }
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

template <bool PACKED, bool CLIP_IN>
int launch_depth(rn_ctx *ctx, int n, const float *Sv, const int32_t *vox, const int32_t *rvc,
                 const float *acc, const float *msgs, const float *cc, float *S_new,
                 float *depth_map, hipStream_t st, int rays_per_center = 0) {
    const int nch = (ctx->p.M + WAVE - 1) / WAVE;
    ProfScope prof(ctx, RN_K_DEPTH, n, st);
#define RN_DE(NCH_)                                                                             \
    hipLaunchKernelGGL((k_depth<NCH_, PACKED, CLIP_IN>), dim3(ray_blocks(n)), dim3(BLOCK), 0, st, \
                       ctx->p, n, Sv, vox, rvc, acc, msgs, ctx->axes, cc, S_new, depth_map,           \
                       rays_per_center)
    if (nch <= 2) RN_DE(2);
    else if (nch <= 4) RN_DE(4);
    else if (nch <= 6) RN_DE(6);
    else if (nch <= 8) RN_DE(8);
    else if (nch <= 12) RN_DE(12);
    else RN_DE(16);
#undef RN_DE
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

// floats of one resident (bricked) accumulator: every axis padded to a multiple of 4
inline int64_t acc_floats(const rn_ctx *ctx) {
    return (int64_t)((ctx->p.gx + 3) / 4) * ctx->p.nby * ctx->p.nbz * 64;
}

int need_axes(rn_ctx *ctx) {
    if (!ctx->have_axes)
        return fail(ctx, RN_ERR_STATE, "rn_set_voxel_grid must be called before this entry point");
    return RN_OK;
}

}  // namespace

extern "C" {

const char *rn_version(void) { return "raynet_hip 0.1 (gfx950)"; }

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
    p.Hf = cfg->H + cfg->padding + 1;
    p.Wf = cfg->W + cfg->padding + 1;
    for (int i = 0; i < 6; i++) p.bbox[i] = cfg->bbox[i];
    const char *sm = getenv("RAYNET_HIP_SCATTER_MODE");
    ctx->scatter_mode = sm ? atoi(sm) : -1;
    const char *bs = getenv("RAYNET_HIP_BOX_LEVEL");      // A/B knob: start at this tile shape
    ctx->box_level = ctx->box_level0 = bs ? max(0, min(2, atoi(bs))) : 0;
    ctx->box_pin = getenv("RAYNET_HIP_BOX_PIN") != nullptr;
    if (hipMalloc(&ctx->axes, sizeof(float) * (p.gx + p.gy + p.gz)) != hipSuccess ||
        hipMalloc(&ctx->box_stats, 2 * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc(&ctx->box_stats_host, 2 * sizeof(unsigned)) != hipSuccess ||
        hipMemset(ctx->box_stats, 0, 2 * sizeof(unsigned)) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return RN_ERR_HIP;
    }
    ctx->box_stats_host[0] = ctx->box_stats_host[1] = 0;
    if (sweep_lds(p) > 160 * 1024) {
        hipFree(ctx->axes);
        delete ctx;
        return RN_ERR_INVALID;
    }
    *out = ctx;
    return RN_OK;
}

void rn_destroy(rn_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->cfg.device);
    if (ctx->axes) hipFree(ctx->axes);
    if (ctx->box_stats) hipFree(ctx->box_stats);
    if (ctx->box_stats_host) hipHostFree(ctx->box_stats_host);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    for (int i = 0; i < 2 * ctx->prof_cap; i++) hipEventDestroy(ctx->prof_ev[i]);
    delete[] ctx->prof_ev;
    delete[] ctx->prof_id;
    delete[] ctx->prof_rays;
    delete ctx;
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
    ctx->box_stats_host[0] = ctx->box_stats_host[1] = 0;
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

int rn_scene_prepare_all(rn_ctx *ctx, int32_t n_images, int32_t n, int64_t rows_per_image,
                         const int32_t *ray_idxs, const float *const *features_views,
                         const float *cameras, const int32_t *order, int32_t *vox, int32_t *rvc,
                         float *Sr, float *ray_segments, void *stream) {
    if (!ctx || n_images < 1 || n < 0 || rows_per_image < n || !ray_idxs || !features_views ||
        !cameras || !vox || !rvc || !Sr)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    if (n == 0) return RN_OK;
    const int N = ctx->p.N;
    const int cam_stride = 12 * N + 12 + 4;
    const float *P = cameras, *P_inv = cameras + 12 * N, *cc = cameras + 12 * N + 12;
    {
        ProfScope prof(ctx, RN_K_TRAVERSE, n * n_images, S(stream));
        hipLaunchKernelGGL((k_traverse<true>), dim3((n + WAVE - 1) / WAVE, n_images), dim3(WAVE), 0,
                           S(stream), ctx->p, n, ray_idxs, P_inv, cc, (const float *)nullptr,
                           (const float *)nullptr, vox, rvc, cam_stride, rows_per_image,
                           ray_segments);
    }
    RN_LAUNCH_CHECK(ctx);
    SweepArgs a{n, ray_idxs, FeatureViews{}, P, P_inv, cc, nullptr, nullptr, nullptr, vox, rvc,
                nullptr, Sr, nullptr, nullptr};
    a.order = order;
    a.fv_table = features_views;
    a.cam_stride = cam_stride;
    a.rows_per_image = rows_per_image;
    a.n_images = n_images;
    a.seg = ray_segments;
    launch_sweep<2, true>(ctx, a, true, S(stream));
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
                                  (first_sweep & RN_SWEEP_UNIFORM_ACC) != 0);
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
                                  true, (first_sweep & RN_SWEEP_UNIFORM_ACC) != 0);
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

int rn_prof_offsets(rn_ctx *ctx, float *start_ms_host) {
    if (!ctx || !start_ms_host) return fail(ctx, RN_ERR_INVALID, "bad argument");
    for (int i = 0; i < ctx->prof_n; i++)
        RN_HIP(ctx, hipEventElapsedTime(start_ms_host + i, ctx->prof_ev[0], ctx->prof_ev[2 * i]));
    return RN_OK;
}

#ifdef RN_SCATTER_STATS
int rn_debug_scatter_stats(unsigned long long *out_host, int reset) {
    if (out_host &&
        hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_scatter_stats), sizeof(unsigned long long) * 8) !=
            hipSuccess)
        return RN_ERR_HIP;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_scatter_stats), z, sizeof(z)) != hipSuccess)
            return RN_ERR_HIP;
    }
    return RN_OK;
}
#endif

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
