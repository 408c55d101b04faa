// raynet_kernels.h -- device code of the RayNet forward_pass hot path for gfx950.
//
// Execution model (MI355X-first, not the reference's one-thread-per-ray):
//   * one 64-lane wavefront owns one ray; lanes stride the D depth planes and
//     the <= M traversed voxels, so every per-ray column (similarities, voxel
//     list, messages) is read and written as coalesced rows;
//   * the sequential cumulative product / sum of the ray potential
//     (mrf_bp.cu:115-167) are DPP wave scans with a carried prefix per 64-voxel
//     chunk; the occupancy term is evaluated once per voxel instead of twice;
//   * the D-deep cost column and the mapped voxel column live in LDS;
//   * the only serial piece, the 3-D DDA (ray_tracing.pyx:64-199), runs one
//     thread per ray in its own kernel (k_traverse) and hands its list over in HBM.
// Built with -ffp-contract=off: index maps are bit-exact w.r.t. the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

namespace rn {

constexpr int WAVE = 64;
constexpr int MAX_VIEWS = 16;

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

struct Params {
    int M, D, N, F, H, W, padding;
    int gx, gy, gz;
    int nby, nbz;        // 4x4x4 bricks along y and z (resident accumulator layout)
    int Hf, Wf;          // feature map extent: H+padding+1, W+padding+1
    float plane_step;    // (1.0f - 0.0f) / (D - 1) of planes_voxels_mapping.cu, divided on the host
    float bbox[6];
};

struct FeatureViews {
    const float *v[MAX_VIEWS];   // one [Hf][Wf][F] map per view
};

// ----------------------------------------------------------------- wave ops
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                           __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf,
                                           false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// row_shr:1,2,4,8 then row_bcast:15 / row_bcast:31 -- the gfx9 wave64 scan.
// Written as `v_op_dpp x, x, x` WITHOUT bound_ctrl: a lane whose DPP source is out of range
// (or whose row is masked) is simply not written, i.e. keeps x -- the identity of any scan
// operator for free.  Through __builtin_amdgcn_update_dpp the compiler needs three
// instructions per step for mul / max and for the two row_bcast steps of add (materialise
// the identity, v_mov_dpp, op).  `s_nop 1`
// covers the VALU-write -> DPP-read hazard, which the assembler does not insert inside asm.
#define RN_SCAN_STEP(OP, X, CTRL) \
    asm("s_nop 1\n\t" OP " %0, %0, %0 " CTRL : "+v"(X))
#define RN_WAVE_SCAN(OP, X)                                             \
    RN_SCAN_STEP(OP, X, "row_shr:1 row_mask:0xf bank_mask:0xf");        \
    RN_SCAN_STEP(OP, X, "row_shr:2 row_mask:0xf bank_mask:0xf");        \
    RN_SCAN_STEP(OP, X, "row_shr:4 row_mask:0xf bank_mask:0xf");        \
    RN_SCAN_STEP(OP, X, "row_shr:8 row_mask:0xf bank_mask:0xf");        \
    RN_SCAN_STEP(OP, X, "row_bcast:15 row_mask:0xa bank_mask:0xf");     \
    RN_SCAN_STEP(OP, X, "row_bcast:31 row_mask:0xc bank_mask:0xf")
__device__ __forceinline__ float wave_scan_add(float x) {
    RN_WAVE_SCAN("v_add_f32_dpp", x);
    return x;
}
__device__ __forceinline__ float wave_scan_mul(float x) {
    RN_WAVE_SCAN("v_mul_f32_dpp", x);
    return x;
}
__device__ __forceinline__ int wave_scan_max(int x) {
    RN_WAVE_SCAN("v_max_i32_dpp", x);
    return x;
}
// value of the previous lane (wave_shr:1); lane 0 receives `first`
__device__ __forceinline__ float wave_shift1(float x, float first) {
    return dpp_f<0x138, 0xf>(first, x);
}
// suffix[i] = sum_{j>i} x[j] within the wavefront (exclusive reverse scan); also returns
// the wave total through `total`
__device__ __forceinline__ float wave_suffix_excl(float x, int lane, float &total) {
    // reverse lane order: ds_bpermute with the byte address of lane 63 - i (__shfl would
    // rebuild the lane id with two v_mbcnt and mask it for its width argument every time)
    const int rev = (63 - lane) << 2;
    const float y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rev, __builtin_bit_cast(int, x)));
    const float incl = wave_scan_add(y);
    total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
    const float excl = dpp_f<0x138, 0xf>(0.0f, incl);
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rev, __builtin_bit_cast(int, excl)));
}
__device__ __forceinline__ float lane63(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ int lane63i(int x) { return __builtin_amdgcn_readlane(x, 63); }
// Wave-wide reductions on the VALU only (no LDS permutes): xor-1 / xor-2 inside the quads,
// 7-i inside the half rows, 15-i inside the rows (every lane then holds its row's result),
// then the two row broadcasts of the scan; lane 63 ends up with the result of the whole
// wave and hands it out through an SGPR.  All 64 lanes must be active.
#define RN_WAVE_REDUCE(OP, X)                                                    \
    RN_SCAN_STEP(OP, X, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");       \
    RN_SCAN_STEP(OP, X, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");       \
    RN_SCAN_STEP(OP, X, "row_half_mirror row_mask:0xf bank_mask:0xf");           \
    RN_SCAN_STEP(OP, X, "row_mirror row_mask:0xf bank_mask:0xf");                \
    RN_SCAN_STEP(OP, X, "row_bcast:15 row_mask:0xa bank_mask:0xf");              \
    RN_SCAN_STEP(OP, X, "row_bcast:31 row_mask:0xc bank_mask:0xf")
__device__ __forceinline__ float wave_sum(float x) {
    RN_WAVE_REDUCE("v_add_f32_dpp", x);
    return lane63(x);
}
__device__ __forceinline__ float wave_max(float x) {
    RN_WAVE_REDUCE("v_max_f32_dpp", x);
    return lane63(x);
}
__device__ __forceinline__ int wave_max_i(int x) {
    RN_WAVE_REDUCE("v_max_i32_dpp", x);
    return lane63i(x);
}
__device__ __forceinline__ int wave_min_i(int x) {
    RN_WAVE_REDUCE("v_min_i32_dpp", x);
    return lane63i(x);
}
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// LDS hand-over between lanes of ONE wavefront: the hardware keeps a wave's DS
// operations in order; this only stops the compiler from moving them.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// utils.cu:1-3, min(max(x, a), b) with a < b: the median of the three, one v_med3_f32 -- which for
// a NaN x returns min3 = a, what fmaxf(NaN, a) passes on (fminf / fmaxf each canonicalise their
// argument first: three instructions; only a SIGNALLING NaN would differ, and every value the
// path clamps is the result of an arithmetic instruction, which quiets it).
__device__ __forceinline__ float clampf(float x, float a, float b) {
    return __builtin_amdgcn_fmed3f(x, a, b);
}

// ------------------------------------------------- exact arithmetic shortcuts
// Bit-identical, cheaper forms of expressions whose results feed INDEX maps (DESIGN.md
// section 6).  Nothing here is an approximation.  (Sharing one refined reciprocal between
// quotients of the same divisor -- x/n and y/n, k*(e-s)/(D-1), sum/|ray|^2 -- is also exact
// inside a range check, saves ~130 VALU instructions per ray and was measured SLOWER with
// its fallback branch in place: the kernel waits on its gathers, not on VALU issue.)
//
// roundf(x) == trunc(x + copysign(pred(0.5), x)) for every one of the 2^32 floats
// (tools/verify_round_trick.c walks them all): 3 instructions instead of 6.
__device__ __forceinline__ float round_half_away(float x) {
    return __builtin_truncf(x + __builtin_copysignf(0x1.fffffep-2f, x));
}

// round_half_away(x / n) without the division, for the pixel a projection lands on.  Only the
// rounded quotient is used, never the quotient: q' = x * rcp(n) is within 2^-22 |q| of the
// correctly rounded x / n (v_rcp_f32: 1 ulp = 2^-23 relative; the product: 2^-24; RN(x / n)
// itself: 2^-24), so the two round to the same integer unless q' lies within that distance of
// a rounding boundary k + 1/2.  |q' - round(q')| <= 1/2 always; `sure` = it stays 2^-21 |q'|
// short of 1/2 (false for a non-finite q' and for |q'| >= 2^20).  The bound on rcp needs a
// NORMAL reciprocal (a divisor beyond 2^126 has a denormal one, flushed to zero; a denormal
// divisor an infinite one): reciprocal_is_normal, one v_cmp_class per divisor.  The caller
// takes the IEEE division when it is not sure; rn_selftest_quotient puts the two side by side.
__device__ __forceinline__ bool reciprocal_is_normal(float rcp_n) {
    return __builtin_amdgcn_classf(rcp_n, 0x108);    // -normal | +normal (v_cmp_class_f32)
}
// (Where it is sure, q' is not within 2^-21 |q'| of a half-way point, so ANY round-to-nearest
// gives the integer roundf gives: v_rndne_f32, one instruction, stands in for the
// three-instruction half-away form; an exact half-way q' has |q' - rndne(q')| = 1/2: not sure.)
__device__ __forceinline__ float round_quotient_fast(float x, float rcp_n, bool &sure) {
    const float q = x * rcp_n;
    const float r = __builtin_rintf(q);
    sure = __builtin_fabsf(q - r) < __builtin_fmaf(__builtin_fabsf(q), -0x1p-21f, 0.5f);
    return r;
}

// ------------------------------------------------------------------- a1
// sampling_schemes.cu:44-90; arithmetic identical to oracle rno_sample_in_bbox
__device__ __forceinline__ void sample_in_bbox(const Params &p, int ray_idx,
                                               const float *__restrict__ P_inv,
                                               const float *__restrict__ cc, float s[3],
                                               float e[3]) {
    const float px = (float)(ray_idx / p.H);
    const float py = (float)(ray_idx % p.H);
    double o[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double a = 0.0;
        a += (double)(P_inv[3 * r + 0] * px);
        a += (double)(P_inv[3 * r + 1] * py);
        a += (double)P_inv[3 * r + 2] * 1.0;
        o[r] = a;
    }
    float dir[3];
#pragma unroll
    for (int i = 0; i < 3; i++) dir[i] = (float)(o[i] / o[3] - (double)cc[i]);
    float t_near = -INFINITY, t_far = INFINITY;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float t1 = (float)(((double)p.bbox[i] - (double)cc[i]) / (double)dir[i]);
        const float t2 = (float)(((double)p.bbox[3 + i] - (double)cc[i]) / (double)dir[i]);
        t_near = fmaxf(fminf(t1, t2), t_near);
        t_far = fminf(fmaxf(t1, t2), t_far);
    }
    const float near_mask = (fabsf(t_near) < fabsf(t_far)) ? 1.0f : 0.0f;
    const float tn = t_near * near_mask + t_far * (1 - near_mask);
    const float tf = (1 - near_mask) * t_near + near_mask * t_far;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        s[i] = cc[i] + tn * dir[i];
        e[i] = cc[i] + tf * dir[i];
    }
}

// ------------------------------------------------------------------- a2
// feature_similarities.cu:10-61: project plane k of the ray into one view and
// return the element offset of its feature vector inside that view's map
__device__ __forceinline__ int feature_offset(const Params &p, const float *__restrict__ Pv,
                                              const float point[3]) {
    float x = 0.0f, y = 0.0f, n = 0.0f;
    x += Pv[0] * point[0]; x += Pv[1] * point[1]; x += Pv[2] * point[2]; x += Pv[3] * 1;
    y += Pv[4] * point[0]; y += Pv[5] * point[1]; y += Pv[6] * point[2]; y += Pv[7] * 1;
    n += Pv[8] * point[0]; n += Pv[9] * point[1]; n += Pv[10] * point[2]; n += Pv[11] * 1;
    x = x / n;
    y = y / n;
    const int half = (p.padding - 1) / 2;
    int fx = (int)(roundf(x) + p.padding - half);   // v_cvt_i32_f32 saturates, NaN -> 0
    int fy = (int)(roundf(y) + p.padding - half);
    fx = min(max(fx, 0), p.W);
    fy = min(max(fy, 0), p.H);
    if (fx == 0 || fy == 0) fx = fy = 0;
    return (fy * p.Wf + fx) * p.F;
}

// The same map for the cooperative sweep (F*4 = 1 << LOG2_VEC_BYTES bytes per vector), as a
// BYTE offset and with fewer instructions, every step exact: the 3-instruction round,
// `+ padding - half` as one addition (exact on integer-valued floats; beyond 2^24 the clamp
// decides either way), the clamp as one v_med3_f32 before the conversion (NaN -> 0 like the
// saturating v_cvt_i32_f32), a 24-bit multiply and a shift instead of two 32-bit multiplies,
// and the two IEEE divisions only for the wavefronts in which some lane's quotient is too close
// to a rounding boundary for round_quotient_fast.
// (in three pieces, so that a wavefront can project into ALL views first and decide once
// whether any of them needs the IEEE quotients: sweep_coop)
// 1: the matrix product and the quotients' fast form; `sure`: lanes whose x AND y quotients
//    are certainly the IEEE division's (a lane mask)
__device__ __forceinline__ void project_fast(const float *__restrict__ Pv, const float point[3],
                                             float &x, float &y, float &n, float &rx, float &ry,
                                             unsigned long long &sure) {
    // The reference starts each sum at 0.0f (`x = 0; x += P0 * X; ...`).  0.0f + a is a for every
    // a but -0 (and keeps NaN a NaN), so a sum started at its first product can differ from the
    // literal one only in the sign of a zero result.  For the numerators that sign is lost:
    // x = +-0 -> quotient +-0 (or NaN for n = 0 either way) -> rounds to +-0 -> + padding.  The
    // DENOMINATOR keeps the literal form: the sign of n = 0 is the sign of an infinite quotient.
    x = Pv[0] * point[0]; y = Pv[4] * point[0]; n = 0.0f;
    x += Pv[1] * point[1]; x += Pv[2] * point[2]; x += Pv[3] * 1;
    y += Pv[5] * point[1]; y += Pv[6] * point[2]; y += Pv[7] * 1;
    n += Pv[8] * point[0]; n += Pv[9] * point[1]; n += Pv[10] * point[2]; n += Pv[11] * 1;
    const float rn = __builtin_amdgcn_rcpf(n);
    bool sure_x, sure_y;
    rx = round_quotient_fast(x, rn, sure_x);
    ry = round_quotient_fast(y, rn, sure_y);
    // (the class test straight into a lane mask: through ballot(bool) the compiler turns the
    // v_cmp_class result into 0 / 1 and compares that again, two instructions per view)
    unsigned long long normal_rcp;
    asm("v_cmp_class_f32_e64 %0, %1, %2" : "=s"(normal_rcp) : "v"(rn), "v"(0x108));
    sure = __builtin_amdgcn_ballot_w64(sure_x) & __builtin_amdgcn_ballot_w64(sure_y) & normal_rcp;
}
// 2: the reference's own quotients (feature_similarities.cu:31-36)
__device__ __forceinline__ void project_ieee(float x, float y, float n, float &rx, float &ry) {
    rx = round_half_away(x / n);
    ry = round_half_away(y / n);
}
// 3: padding shift, clamp, the (0, 0) rule, byte offset of the vector in the view's map
template <int LOG2_VEC_BYTES>
__device__ __forceinline__ int project_offset(const Params &p, float rx, float ry, float pad_shift) {
    rx += pad_shift;
    ry += pad_shift;
    const unsigned fx = (unsigned)(int)__builtin_amdgcn_fmed3f(rx, 0.0f, (float)p.W);
    const unsigned fy = (unsigned)(int)__builtin_amdgcn_fmed3f(ry, 0.0f, (float)p.H);
    const unsigned off = (__umul24(fy, (unsigned)p.Wf) + fx) << LOG2_VEC_BYTES;
    return min(fx, fy) == 0u ? 0 : (int)off;
}
template <int LOG2_VEC_BYTES>
__device__ __forceinline__ int feature_offset_bytes(const Params &p,
                                                    const float *__restrict__ Pv,
                                                    const float point[3], float pad_shift) {
    float x, y, n, rx, ry;
    unsigned long long sure;
    project_fast(Pv, point, x, y, n, rx, ry, sure);
    if (sure != __builtin_amdgcn_read_exec())         // ~4 % of the wavefronts per view at config 2
        project_ieee(x, y, n, rx, ry);
    return project_offset<LOG2_VEC_BYTES>(p, rx, ry, pad_shift);
}

__device__ __forceinline__ void plane_point(const float s[3], const float e[3], int k, int D,
                                            float point[3]) {
#pragma unroll
    for (int a = 0; a < 3; a++) point[a] = s[a] + k * (e[a] - s[a]) / (D - 1);
}

// Generic plane sweep: lane = depth plane, the F-long dot is walked serially in
// the reference's order (pairs i<j, then f), so it tracks the oracle to the
// last bit before expf.  Any N, any F.  Writes raw pair sums / pairs to Sl[D].
__device__ __forceinline__ void sweep_generic(const Params &p, const FeatureViews &fv,
                                              const float *const *__restrict__ tbl,
                                              const float *__restrict__ P, const float s[3],
                                              const float e[3], int lane, float *Sl) {
    const int pairs = (p.N * (p.N - 1)) / 2;
    for (int base = 0; base < p.D; base += WAVE) {
        const int k = base + lane;
        if (k < p.D) {
            float point[3];
            plane_point(s, e, k, p.D, point);
            float acc = 0.0f;
            for (int i = 0; i < p.N; i++) {
                const float *fi = (tbl ? tbl[i] : fv.v[i]) + feature_offset(p, P + 12 * i, point);
                for (int j = i + 1; j < p.N; j++) {
                    const float *fj =
                        (tbl ? tbl[j] : fv.v[j]) + feature_offset(p, P + 12 * j, point);
                    float dot = 0.0f;
                    for (int f = 0; f < p.F; f++) dot += fi[f] * fj[f];
                    acc += dot;
                }
            }
            Sl[k] = acc / pairs;
        }
    }
}

// Cooperative plane sweep for F = 4*V4*LPS: LPS lanes fetch one feature vector as one
// contiguous segment (a whole 128-B line for F=32), 16*V4 bytes per lane, 64/LPS planes per
// load round.  Each lane multiplies its 4*V4 channels for all view pairs, the LPS partial
// sums are folded with an xor butterfly.
constexpr int SWEEP_V4 = 1;                 // float4s per lane and view
constexpr int SWEEP_UNROLL2_MAX_VIEWS = 6;  // two load rounds in flight up to this many views
// x + (x of the lane the DPP control names), one VALU instruction, no LDS round trip.
// quad_perm:[1,0,3,2] / [2,3,0,1] = xor 1 / xor 2; row_half_mirror pairs lane i with 7-i of
// its group of 8, which sums the two quads once they are quad-uniform.
#define RN_ADD_DPP(OUT, MOVED, STAY, CTRL)                                                   \
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf"           \
        : "=v"(OUT) : "v"(MOVED), "v"(STAY))
#define RN_DPP_XOR1 "quad_perm:[1,0,3,2]"
#define RN_DPP_XOR2 "quad_perm:[2,3,0,1]"
#define RN_DPP_MIRROR8 "row_half_mirror"

// one load round of the cooperative sweep: this lane's 4*V4 channels of plane `src` (within
// the chunk) in every view, multiplied out over the view pairs; returns the lane's partial
// REF_HELD: view 0's vector is the same for every plane of the chunk (see sweep_coop) and
// comes in `ref` instead of being loaded again
template <int NV, int V4, bool REF_HELD = false>
__device__ __forceinline__ float sweep_round(const float *const (&vbase)[NV], const int (&offb)[NV],
                                             int src, unsigned part_bytes,
                                             const float2v (&ref)[2 * V4]) {
    // lane's 16*V4 bytes of every view's vector, as channel pairs: the packed FMAs below
    // then work on the register pairs exactly as the loads deliver them
    float2v f2[NV][2 * V4];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        if (REF_HELD && v == 0) {
#pragma unroll
            for (int c = 0; c < 2 * V4; c++) f2[0][c] = ref[c];
            continue;
        }
        // byte offset from a uniform GLOBAL base: loads with a 32-bit register offset
        // (global_load ... s[base]), no 64-bit address arithmetic per lane
        const unsigned ob = (unsigned)__shfl(offb[v], src) + part_bytes;
        typedef const __attribute__((address_space(1))) char *gptr;
        typedef const __attribute__((address_space(1))) float4v *gptr4;
#pragma unroll
        for (int q = 0; q < V4; q++) {
            const float4v f = *(gptr4)((gptr)vbase[v] + ob + 16u * q);
            f2[v][2 * q] = float2v{f.x, f.y};
            f2[v][2 * q + 1] = float2v{f.z, f.w};
        }
    }
    // sum over view pairs i<j of <f_i, f_j>, as  sum_j <f_0 + ... + f_{j-1}, f_j>  on this
    // lane's channels: NV-1 packed FMAs and NV-2 packed adds per channel pair instead
    // of NV(NV-1)/2 products.  This is the only place where multiply-adds may fuse and
    // where the summation order departs from the reference's serial pair loop (the
    // kernels are VALU-issue bound; tolerance-tested against the oracle).
    float acc;
    {
#pragma clang fp contract(fast)
        float2v run[2 * V4], a2[2 * V4];
#pragma unroll
        for (int c = 0; c < 2 * V4; c++) {
            run[c] = f2[0][c];
            a2[c] = float2v{0.f, 0.f};
        }
#pragma unroll
        for (int j = 1; j < NV; j++) {
#pragma unroll
            for (int c = 0; c < 2 * V4; c++) {
                a2[c] = run[c] * f2[j][c] + a2[c];
                if (j + 1 < NV) run[c] += f2[j][c];
            }
        }
        float2v t = a2[0];
#pragma unroll
        for (int c = 1; c < 2 * V4; c++) t += a2[c];
        acc = t.x + t.y;
    }
    return acc;
}

// the LPS load rounds of one 64-sample chunk; returns the pair sum of the sample this lane ends up
// holding, `mine_round` says which (sample = mine_round * SPL + sub).
// RPW rays share the wavefront (sweep_coop): sample j is plane j % DPAD of ray j / DPAD, DPAD =
// 64 / RPW; a load round's 64 / LPS samples never straddle two rays (DPAD >= 16), so the held
// vector of view 0 is ref[ray of the round].  `live` = planes of a ray that exist (D, or what is
// left of D in this chunk when RPW = 1), `nrays` = rays of this wavefront that exist.  PARTIAL: rounds
// whose samples are all beyond either are skipped (wave-uniform branches) -- the cost of a sweep is
// proportional to D as the reference's loop is (feature_similarities.cu:84), not to 64; a full
// chunk takes the branch-free instantiation, whose loads the compiler schedules across rounds.
template <int NV, int LPS, int V4, bool REF_HELD, int RPW, bool PARTIAL>
__device__ __forceinline__ float sweep_rounds(const float *const (&vbase)[NV], const int (&offb)[NV],
                                              int sub, int part, unsigned part_bytes,
                                              const float2v (&ref)[RPW][2 * V4], int live, int nrays,
                                              int &mine_round) {
    constexpr int SPL = WAVE / LPS;
    constexpr int DPAD = WAVE / RPW;
    static_assert(DPAD % (2 * SPL) == 0 || LPS != 8, "a pair of load rounds stays within one ray");
    float mine = 0.0f;
    // The LPS partial sums of a plane are folded across its lanes with DPP adds (no LDS
    // permutes).
    if (LPS == 8 && NV <= SWEEP_UNROLL2_MAX_VIEWS) {
        // two rounds of loads in flight while the view count leaves registers for it, and
        // their two reductions transposed: "odd" lanes fold round 2t+1, the others round
        // 2t, so the xor-1 step serves both rounds at once.  The last step, lane i with
        // 7-i, is the only 8-lane exchange DPP has on gfx9; it flips the lane parity, so
        // "odd" is flipped in the upper quad to meet it.
        const bool odd = (part ^ (part >> 2)) & 1;
        mine_round = (part & ~1) | (int)odd;
#pragma unroll
        for (int t = 0; t < LPS / 2; t++) {
            const int first = (2 * t) * SPL;             // first sample of the pair of rounds
            if (PARTIAL && (first % DPAD >= live || first / DPAD >= nrays)) continue;
            const float2v(&rf)[2 * V4] = ref[RPW == 1 ? 0 : first / DPAD];
            const float a0 = sweep_round<NV, V4, REF_HELD>(vbase, offb, first + sub, part_bytes, rf);
            const float a1 = sweep_round<NV, V4, REF_HELD>(vbase, offb, first + SPL + sub, part_bytes, rf);
            const float stay = odd ? a1 : a0, moved = odd ? a0 : a1;
            float r;
            RN_ADD_DPP(r, moved, stay, RN_DPP_XOR1);
            RN_ADD_DPP(r, r, r, RN_DPP_XOR2);
            RN_ADD_DPP(r, r, r, RN_DPP_MIRROR8);
            if ((part >> 1) == t) mine = r;
        }
    } else {
        static_assert(LPS <= 8, "sweep_coop folds at most 8 lanes per plane");
        mine_round = part;
#pragma unroll
        for (int it = 0; it < LPS; it++) {
            const int first = it * SPL;
            if (PARTIAL && (first % DPAD >= live || first / DPAD >= nrays)) continue;
            float acc = sweep_round<NV, V4, REF_HELD>(vbase, offb, first + sub, part_bytes,
                                                      ref[RPW == 1 ? 0 : first / DPAD]);
            if (LPS >= 2) RN_ADD_DPP(acc, acc, acc, RN_DPP_XOR1);
            if (LPS >= 4) RN_ADD_DPP(acc, acc, acc, RN_DPP_XOR2);
            if (LPS >= 8) RN_ADD_DPP(acc, acc, acc, RN_DPP_MIRROR8);
            if (part == it) mine = acc;
        }
    }
    return mine;
}

// RPW = 1: the wavefront's ray has the segment s -> e (wave-uniform), lane k projects plane
// base + k, Sl[D] receives the column.  RPW = 2 / 4 (D <= 32 / 16, k_sweep_map_packed): the
// wavefront takes RPW rays at once, lane l = plane l % DPAD of ray l / DPAD with that ray's
// segment in ITS s / e; ray q's column goes to Sl[q * D ...]; `nrays` of them exist (the lanes of
// a missing ray shadow the last one).  The arithmetic per (ray, plane) sample is the same
// instruction sequence either way: the columns are the same bits.
template <int NV, int LPS, bool FAST = false, int RPW = 1>
__device__ __forceinline__ void sweep_coop(const Params &p, const FeatureViews &fv,
                                           const float *const *__restrict__ tbl,
                                           const float *__restrict__ P, const float s[3],
                                           const float e[3], int lane, float *Sl, int nrays = 1) {
    constexpr int SPL = WAVE / LPS;       // planes per load round
    constexpr int V4 = SWEEP_V4;          // float4s per lane and view
    constexpr int DPAD = WAVE / RPW;      // lanes (= plane slots) per ray
    const float *vbase[NV];               // uniform per-view bases (kernel argument or table)
#pragma unroll
    for (int v = 0; v < NV; v++) vbase[v] = tbl ? tbl[v] : fv.v[v];
    const int sub = lane / LPS;           // which sample of the load round
    const int part = lane % LPS;          // which float4 of the vector
    const unsigned part_bytes = (16u * V4) * (unsigned)part;
    const int pairs = (NV * (NV - 1)) / 2;
    // F = 4*V4*LPS floats per vector
    constexpr int LOG2_VEC_BYTES = LPS * V4 == 8 ? 7 : LPS * V4 == 4 ? 6 : LPS * V4 == 2 ? 5 : 4;
    static_assert((16 * V4 * LPS) == (1 << LOG2_VEC_BYTES), "vector bytes must be a power of two");
    const float pad_shift = (float)(p.padding - (p.padding - 1) / 2);
    const int q_lane = RPW == 1 ? 0 : lane / DPAD;
    for (int base = 0; base < (RPW == 1 ? p.D : 1); base += WAVE) {
        // lane k projects plane base+k into every view
        int offb[NV];           // BYTE offset of the plane's feature vector in every view
        {
            const int k = min(RPW == 1 ? base + lane : lane % DPAD, p.D - 1);
            float point[3];
            plane_point(s, e, k, p.D, point);
            // The 12 entries of a view's matrix are wave-uniform: scalar loads into SGPRs.  (While
            // the class test below went through ballot(bool) the compiler hoisted all NV x 12 of
            // them out of the chunk loop and at 9 views spilled 125 SGPRs, ~300 v_writelane /
            // v_readlane in the kernel; with the test written as a lane mask it reloads the
            // matrices per chunk and spills none -- 5.08 -> 4.11 G VALU instructions per launch
            // at config 4.  What was tried against the spills before that:)
            // Every way of not spilling them was measured SLOWER at config 4 (17.0 -> 18.5 - 20.5
            // ms, with 11 % fewer VALU instructions): reloading a view's matrix where it is used
            // (the pointer made opaque by an empty asm),
            // requesting it one view ahead, staging the matrices in LDS -- the reloads' latency,
            // or the asm barriers between the views' projections, cost more than the spill code
            // (profiles/r03_exp_view_matrix_sgprs.txt).
#pragma unroll
            for (int v = 0; v < NV; v++)
                offb[v] = feature_offset_bytes<LOG2_VEC_BYTES>(p, P + 12 * v, point, pad_shift);
        }
        // View 0 is the reference image itself: every plane of a ray projects onto the
        // ray's own pixel there (up to the rounding of the projection, which is checked, not
        // assumed), so its feature vector is fetched once per ray and chunk instead of once per
        // load round -- a fifth of the sweep's gathers at 5 views.
        int off0[RPW];
#pragma unroll
        for (int q = 0; q < RPW; q++) off0[q] = __builtin_amdgcn_readlane(offb[0], q * DPAD);
        int off0_mine = off0[0];
#pragma unroll
        for (int q = 1; q < RPW; q++) off0_mine = q_lane == q ? off0[q] : off0_mine;
        const bool ref_held = __all(offb[0] == off0_mine);
        float2v ref[RPW][2 * V4];
#pragma unroll
        for (int q = 0; q < RPW; q++)
#pragma unroll
            for (int c = 0; c < 2 * V4; c++) ref[q][c] = float2v{0.f, 0.f};
        if (ref_held) {
            typedef const __attribute__((address_space(1))) char *gptr;
            typedef const __attribute__((address_space(1))) float4v *gptr4;
#pragma unroll
            for (int q = 0; q < RPW; q++)
#pragma unroll
                for (int c = 0; c < V4; c++) {
                    const float4v f = *(gptr4)((gptr)vbase[0] + (unsigned)off0[q] + part_bytes + 16u * c);
                    ref[q][2 * c] = float2v{f.x, f.y};
                    ref[q][2 * c + 1] = float2v{f.z, f.w};
                }
        }
        float mine = 0.0f;
        int mine_round;         // which load round's sample this lane ends up holding
        const int live = RPW == 1 ? p.D - base : p.D;
        const bool full = live >= DPAD && nrays == RPW;
        if (ref_held && full)
            mine = sweep_rounds<NV, LPS, V4, true, RPW, false>(vbase, offb, sub, part, part_bytes, ref, live, nrays, mine_round);
        else if (ref_held)
            mine = sweep_rounds<NV, LPS, V4, true, RPW, true>(vbase, offb, sub, part, part_bytes, ref, live, nrays, mine_round);
        else
            mine = sweep_rounds<NV, LPS, V4, false, RPW, true>(vbase, offb, sub, part, part_bytes, ref, live, nrays, mine_round);
        const int j = mine_round * SPL + sub;           // the sample this lane holds
        const int k = RPW == 1 ? base + j : j % DPAD;
        const int q = RPW == 1 ? 0 : j / DPAD;
        // FAST (resident path): a constant factor of the softmax's input, value-only
        if (k < p.D && q < nrays) Sl[q * p.D + k] = FAST ? mine * (1.0f / pairs) : mine / pairs;
    }
}

// every lane receives the maximum / the sum over its group of G = 16, 32 or 64 lanes (G = 64:
// wave_max / wave_sum).  The butterfly is the one RN_WAVE_REDUCE runs, cut off after the rows for
// G = 16 and after the first row broadcast for G = 32 -- a column of D <= G values padded with the
// operator's identity gets the very bits the 64-lane reduction gives it.
#define RN_GROUP_REDUCE16(OP, X)                                                 \
    RN_SCAN_STEP(OP, X, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");       \
    RN_SCAN_STEP(OP, X, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");       \
    RN_SCAN_STEP(OP, X, "row_half_mirror row_mask:0xf bank_mask:0xf");           \
    RN_SCAN_STEP(OP, X, "row_mirror row_mask:0xf bank_mask:0xf")
template <int G>
__device__ __forceinline__ float group_pick(float x, int lane) {
    if (G == 16) return x;            // every lane of a row holds the row's result
    // G = 32: after row_bcast:15 the rows 1 and 3 hold the results of lanes 0-31 / 32-63
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 31));
    const float hi = lane63(x);
    return lane < 32 ? lo : hi;
}
template <int G>
__device__ __forceinline__ float group_max(float x, int lane) {
    static_assert(G == 16 || G == 32, "groups of 16 or 32 lanes");
    RN_GROUP_REDUCE16("v_max_f32_dpp", x);
    if (G == 32) RN_SCAN_STEP("v_max_f32_dpp", x, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    return group_pick<G>(x, lane);
}
template <int G>
__device__ __forceinline__ float group_sum(float x, int lane) {
    static_assert(G == 16 || G == 32, "groups of 16 or 32 lanes");
    RN_GROUP_REDUCE16("v_add_f32_dpp", x);
    if (G == 32) RN_SCAN_STEP("v_add_f32_dpp", x, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    return group_pick<G>(x, lane);
}

// a / b for VALUE arithmetic only (never feeds an index): hardware reciprocal when FAST
template <bool FAST>
__device__ __forceinline__ float vdiv(float a, float b) {
    return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b;
}
// expf(x) for x <= 0: the library's own sequence (extended-precision x*log2(e), v_exp_f32,
// ldexp, underflow select) minus its overflow select and range constants -- x <= 0 cannot
// overflow.  Bit-identical to expf for every x <= 0, including -inf (0) and NaN.
__device__ __forceinline__ float exp_nonpos(float x) {
    const float log2e_hi = 0x1.715476p+0f, log2e_lo = 0x1.4ae0bep-26f;   // 0x3fb8aa3b, 0x32a5705f
    const float ph = x * log2e_hi;
    float pl = __builtin_fmaf(x, log2e_hi, -ph);
    pl = __builtin_fmaf(x, log2e_lo, pl);
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    const float r = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
    return x < -0x1.9d1da00000000p+6f ? 0.0f : r;       // 0xc2ce8ed0, the library's underflow bound
}
// feature_similarities.cu:109-123 on the LDS column
template <bool FAST = false>
__device__ __forceinline__ void softmax_column(int D, int lane, float *Sl) {
    float mx = -INFINITY;
    for (int k = lane; k < D; k += WAVE) mx = fmaxf(mx, Sl[k]);
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int k = lane; k < D; k += WAVE) {
        const float d = Sl[k] - mx;
        // FAST (resident path, value-only): v_exp_f32 on the rounded product, as the occupancy's
        // exponential: relative error <= |d| * 1.44 * 2^-24 on a term that
        // is exp(d) of the column's sum -- below 1e-7 of it for any d
#if !defined(RN_EXACT_OCC_EXP) && !defined(RN_EXACT_SOFTMAX_EXP)
        const float v = FAST ? __builtin_amdgcn_exp2f(d * 0x1.715476p+0f)
                             : (d <= 0.0f ? exp_nonpos(d) : expf(d));
#else
        const float v = d <= 0.0f ? exp_nonpos(d) : expf(d);     // d > 0 only for NaN / inf input
#endif
        Sl[k] = v;
        sum += v;
    }
    sum = wave_sum(sum);
    for (int k = lane; k < D; k += WAVE) Sl[k] = vdiv<FAST>(Sl[k], sum);
}

// ... and on the RPW columns of a wavefront that sweeps RPW rays at once (sweep_coop): lane l owns
// plane l % DPAD of ray l / DPAD; same expressions per element, the reductions per group of lanes
template <bool FAST, int RPW>
__device__ __forceinline__ void softmax_columns(int D, int lane, float *Sl) {
    constexpr int DPAD = WAVE / RPW;
    const int k = lane % DPAD;
    float *x = Sl + (lane / DPAD) * D + k;
    const bool mine = k < D;
    const float mx = group_max<DPAD>(mine ? *x : -INFINITY, lane);
    float v = 0.0f;
    if (mine) {
        const float d = *x - mx;
#if !defined(RN_EXACT_OCC_EXP) && !defined(RN_EXACT_SOFTMAX_EXP)
        v = FAST ? __builtin_amdgcn_exp2f(d * 0x1.715476p+0f) : (d <= 0.0f ? exp_nonpos(d) : expf(d));
#else
        v = d <= 0.0f ? exp_nonpos(d) : expf(d);
#endif
    }
    const float sum = group_sum<DPAD>(v, lane);
    if (mine) *x = vdiv<FAST>(v, sum);
}

// ------------------------------------------------------------------- a3
// (the DDA itself lives in k_traverse, raynet_hip.hip)
// voxel list element access: the reference's [M][3] triples or the packed form
template <bool PACKED>
__device__ __forceinline__ void load_voxel(const int32_t *__restrict__ row, int i, int &x,
                                           int &y, int &z) {
    if (PACKED) {
        const int v = row[i];
        x = v >> 20;
        y = (v >> 10) & 1023;
        z = v & 1023;
    } else {
        x = row[3 * i];
        y = row[3 * i + 1];
        z = row[3 * i + 2];
    }
}
__device__ __forceinline__ int pack_voxel(int x, int y, int z) {
    return (x << 20) | (y << 10) | z;
}

// Slab boxes: per block of 64 consecutive rows and per 16 steps, the bounding box of the
// voxels those rays visit there (k_traverse writes them, k_scatter_box merges them).
constexpr int SLAB_BOX_STEPS = 16;
__host__ __device__ __forceinline__ int slab_box_count(int M) {
    return (M + SLAB_BOX_STEPS - 1) / SLAB_BOX_STEPS;
}

// ------------------------------------------------------------------- a4
// planes_voxels_mapping.cu:6-92 for one ray, wave-parallel.
//   * t_i is computed per lane;
//   * the reference's monotone (left, right) walk equals
//       left_i = max_{j<=i} L(t_j),  L(t) = first l with !(t-l*step>0 && t-(l+1)*step>0),
//     evaluated with the same fp32 expressions, so the plane indices are exact;
//   * vals[] (LDS, M floats) receives the un-normalised interpolation, the sum
//     is returned wave-uniform.
//   * STAGED: the caller parked the ray's packed ids in vals[0 .. count) (k_sweep_map's
//     LDS-DMA); a run-time choice between that and the global row turns the load into a flat
//     load of a selected pointer.
// TABLE (the resident path): the plane positions `0.0f + l * step`, l = 0 .. D, come from an LDS
// table `pos` holding exactly those fp32 values, and
//   * the walk becomes a count.  fl(t - a) > 0 iff t > a, and the positions increase with l, so
//     the walk's answer is L* = #{l >= 1 : pos[l] < t}.  With X = t (D - 1): pos[l] < t iff
//     l < X (1 + e), |e| < 2^-22 (the roundings of step, l * step), and g = (int)fl(t (D - 1))
//     is floor(X (1 + e')), |e'| <= 2^-24; X <= D <= 4096 (rn_create), so both L* and g are
//     floor(X) unless X is within 2^-10 of an integer m, and then both lie in {m - 1, m}:
//     |L* - g| <= 1 always, i.e. L* = (a - 1) + [pos[a] < t] + [pos[a + 1] < t] with
//     a = max(g, 1) -- two table entries decide, no loop (the reference's loop costs a
//     divergent ~2 x 9 instructions here);
//   * t = sum / |ray|^2 is formed as Markstein's correctly rounded quotient from ONE IEEE
//     reciprocal per ray: y = RN(1 / b), q0 = RN(a y), r = a - b q0 (exact in the FMA),
//     q = RN(q0 + r y) = RN(a / b) whenever nothing over- or underflows -- which `div_ok`
//     (|ray|^2 within 2^-60 .. 2^60 and every lane's dividend below 2^60; quotients below 5e-5
//     clamp to eps anyway) guarantees; the IEEE division otherwise.  rn_selftest_mapping puts both next to the reference forms.
__device__ __forceinline__ float markstein_div(float a, float b, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}
__device__ __forceinline__ bool markstein_ok(float b) {
    return b >= 0x1p-60f && b <= 0x1p60f;
}
// ... and the dividend: below 2^60 the quotient cannot overflow (an infinite q0 makes r NaN)
__device__ __forceinline__ bool markstein_ok_dividend(float a) {
    return __builtin_fabsf(a) < 0x1p60f;
}
// first plane index of the walk (planes_voxels_mapping.cu:60-67) from the table, see above
__device__ __forceinline__ int plane_index_from_table(const float *pos, float t, int D) {
    const int a = max((int)(t * (D - 1)), 1);
    return (a - 1) + (pos[a] < t ? 1 : 0) + (pos[a + 1] < t ? 1 : 0);
}
// ... and the reference's own walk, from anywhere at or below its answer
__device__ __forceinline__ int plane_index_walk(float t, int D, float step) {
    int L = max(0, (int)(t * (D - 1)) - 1);
    while ((t - (0.0f + L * step) > 0) && (t - (0.0f + (L + 1) * step) > 0)) L++;
    return L;
}

template <bool PACKED, bool FAST = false, bool STAGED = false, bool TABLE = false>
__device__ __forceinline__ float map_planes_to_voxels(const Params &p,
                                                      const float *__restrict__ axes,
                                                      const int32_t *__restrict__ vrow,
                                                      int count, const float s[3],
                                                      const float e[3], const float *Sl,
                                                      float *vals, int lane, int n_staged = 0,
                                                      const float *pos = nullptr) {
    const float eps = 1e-4f;
    float ray[3], ray_norm = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) ray[i] = e[i] - s[i];
#pragma unroll
    for (int i = 0; i < 3; i++) ray_norm += ray[i] * ray[i];
    const float step = p.plane_step;       // the same IEEE quotient, once per context
    const bool div_ok = TABLE && markstein_ok(ray_norm);      // (wave-uniform)
    const float rcp_norm = TABLE ? 1.0f / ray_norm : 0.0f;    // RN(1 / b): an IEEE division
    int carry = 0;
    float total = 0.0f;
    for (int base = 0; base < count; base += WAVE) {
        const int i = base + lane;
        const bool valid = i < count;
        int L = 0;
        float t = 0.0f;
        if (valid) {
            int x, y, z;
            if (PACKED && STAGED) {
                const int v = __builtin_bit_cast(int, vals[i]);
                x = v >> 20;
                y = (v >> 10) & 1023;
                z = v & 1023;
            } else {
                load_voxel<PACKED>(vrow, i, x, y, z);
            }
            float sum = 0.0f;
            float vd = axes[x];
            vd -= s[0];
            sum += ray[0] * vd;
            vd = axes[p.gx + y];
            vd -= s[1];
            sum += ray[1] * vd;
            vd = axes[p.gx + p.gy + z];
            vd -= s[2];
            sum += ray[2] * vd;
            if (TABLE) {
                float q;
                if (div_ok && __all(markstein_ok_dividend(sum))) q = markstein_div(sum, ray_norm, rcp_norm);
                else q = sum / ray_norm;
                t = clampf(q, eps, 1 - eps);
                L = plane_index_from_table(pos, t, p.D);
            } else {
                t = clampf(sum / ray_norm, eps, 1 - eps);
                // The walk's answer L* is the first L at which (t - L*step > 0 && t - (L+1)*step > 0)
                // fails; both hold for every L below it, so the walk may start anywhere at or below
                // L*.  With q = t / step, L* is ceil(q) - 1 up to the rounding of these fp32
                // expressions when q is within ~1e-5 of an integer k (then k - 1 or k), and
                // floor(t * (D - 1)) is floor(q) up to the same (k - 1 or k): one below it is
                // never above L*, and is L* - 1 for all but those boundary cases -- one loop
                // iteration instead of two (round 1 started two below).
                L = plane_index_walk(t, p.D, step);
            }
        }
        int left = max(wave_scan_max(valid ? L : 0), carry);
        carry = lane63i(left);
        float val = 0.0f;
        if (valid) {
            const int right = left + 1;
            float left_d = fabsf(t - (TABLE ? pos[left] : 0.0f + left * step));
            float right_d = fabsf(t - (TABLE ? pos[right] : 0.0f + right * step));
            const float c1 = 1.0f - vdiv<FAST>(left_d, left_d + right_d);
            const float c2 = 1.0f - vdiv<FAST>(right_d, left_d + right_d);
            val = c1 * Sl[left] + c2 * Sl[right];
            vals[i] = val;
        }
        total += val;
    }
    return wave_sum(total);
}

// ------------------------------------------------------------------ a5/a6
// Value arithmetic of the BP / depth kernels (never index-producing): hardware
// reciprocal and log2 instead of the ~10-instruction IEEE division and ~15-instruction
// logf sequences.  Each is within ~2 ulp; the messages' own fp32 conditioning
// (eps * exp(|m|)) dominates that by far.  The occupancy's exponential
// is v_exp_f32 on the rounded product (below); -DRN_EXACT_OCC_EXP: the library sequence.
#ifndef RN_EXACT_BP_MATH
__device__ __forceinline__ float bp_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float bp_log(float x) {
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;    // v_log_f32 is log2
}
// log pos - log neg (the message, mrf_bp.cu:160-165): one multiplication by ln 2 for both
__device__ __forceinline__ float bp_log_ratio(float pos, float neg) {
    return (__builtin_amdgcn_logf(pos) - __builtin_amdgcn_logf(neg)) * 0.6931471805599453f;
}
#else
__device__ __forceinline__ float bp_div(float a, float b) { return a / b; }
__device__ __forceinline__ float bp_log(float x) { return logf(x); }
__device__ __forceinline__ float bp_log_ratio(float pos, float neg) { return logf(pos) - logf(neg); }
#endif
__device__ __forceinline__ float occupancy_to_ray(float acc, float msg) {
    // mrf_bp.cu:12-35
    // t1 = exp(0 - max(0,mu)), t2 = exp(mu - max(0,mu)): one of the two is exp(0) = 1
    // exactly, the other exp(-|mu|) -- one exponential gives both, bit for bit
    const float mu = acc - msg;
#ifndef RN_EXACT_OCC_EXP
    // t2 / (t1 + t2) is 1 / (1 + exp(-mu)) on either side of 0; the reference forms it from
    // exp(-|mu|) so that nothing overflows -- v_exp_f32 and v_rcp_f32 saturate instead (exp(-mu)
    // = inf -> 0 -> the clamp's 1e-4, as the two-term form gives), NaN stays NaN.  v_exp_f32 on
    // the rounded product mu * log2(e) makes the exponential wrong by <= |mu| * 1.44 * 2^-24
    // relative: 8e-7 where the occupancy is not clamped anyway (|mu| <= 9.21; CUDA's own expf,
    // what the reference runs, is specified to 2 ulp = 2.4e-7), an order of magnitude below what
    // the fp32 subtraction 1 - o already costs the transmittance next to the clamp (6e-8 / 1e-4).
    // Full-size parity with the C oracle is unchanged by it.
    return clampf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mu * -0x1.715476p+0f)), 1e-4f,
                  (float)(1 - 1e-4));
#else
    // the exact-arithmetic build (tests/test_exact_build_gpu.py): the library's own exponential
    const float e = exp_nonpos(0 - fabsf(mu));
    const float t1 = mu > 0.0f ? e : 1.0f;
    const float t2 = mu > 0.0f ? 1.0f : e;
    return clampf(bp_div(t2, t1 + t2), 1e-4f, (float)(1 - 1e-4));
#endif
}

}  // namespace rn
