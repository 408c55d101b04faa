/*
 * raynet_oracle.c -- CPU restatement of RayNet's forward_pass hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under raynet_amd/ may import, link or
 * call this file.  It is used by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py, always as the checker / the timed CPU
 * baseline, never as the product path.
 *
 * Every function restates one reference function in plain C99, scalar, fp32
 * unless the reference promotes to double.  Build with
 *     gcc -O2 -ffp-contract=off -fno-fast-math
 * so that no multiply-add is fused: the HIP kernels are built the same way and
 * the integer outputs (feature indices, voxel index maps, plane indices) are
 * bit-exact between the two.
 *
 * Reference files restated (paths relative to /root/reference/raynet):
 *   cuda_implementations/sampling_schemes.cu:5-90      -> rno_sample_in_bbox
 *   cuda_implementations/feature_similarities.cu:10-124 -> rno_similarities
 *   ray_marching/ray_tracing.pyx:64-199                 -> rno_voxel_traversal
 *       (Cython flavour: run-time fp32 bbox; the CUDA twin ray_tracing.cu:9-143
 *        differs only by double-promoted bbox literals)
 *   cuda_implementations/planes_voxels_mapping.cu:6-92  -> rno_planes_to_voxels
 *   cuda_implementations/mrf_bp.cu:3-35                 -> occupancy_to_ray
 *   cuda_implementations/mrf_bp.cu:88-177 + mrf/mrf_np.py:243-330 -> rno_bp_ray
 *   cuda_implementations/mrf_bp.cu:37-86 + mrf/mrf_np.py:333-385  -> rno_depth_ray
 *   cuda_implementations/raynet_fp.py:55-227            -> rno_fused_bp / rno_fused_depth
 *
 * How each piece is pinned to the reference (tests/test_oracle_golden.py,
 * tests/test_sampling_reference.py, tests/test_saturated_golden.py):
 *   rno_sample_in_bbox    the reference's NumPy sampling scheme on 11 cameras (1.1e-6)
 *   rno_voxel_traversal   the reference's compiled Cython traversal (bit-exact)
 *   rno_planes_to_voxels  the reference's NumPy `li` / `li_2` mappings
 *   rno_bp_ray / rno_depth_ray   the reference's mrf_np.py, both message forms on the small
 *                         scenes; the robust form on the saturated 96,000-ray scene
 *   rno_similarities      the reference's own batch_compute_similarities, its .cu text compiled
 *                         unchanged for gfx950 (oracle/build_ref_cu.py) and run on an MI355X:
 *                         fixture tests/golden/ref_cu_gfx950.npz, <= 1e-7
 *                         (tests/test_reference_kernels.py); its geometry also by the reference's
 *                         NumPy `project` (tests/test_projection_reference.py)
 *   rno_fused_bp / _depth compositions of the above
 *
 * Decisions where the reference's flavours disagree:
 *   Q3  the D-plane column is zero-initialised before accumulation;
 *   Q4  rays with count <= 1 send no message and get an all-zero distribution
 *       (mrf_np.py:300, :376), instead of the +inf of mrf_bp.cu:157-165;
 *   Q9  traversal follows the Cython arithmetic (the one with an exact golden);
 *   Q11 count is written as 0 when the first voxel is outside the grid.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t M;        /* max marched voxels per ray          */
    int32_t D;        /* depth planes                        */
    int32_t N;        /* views (reference + neighbours)      */
    int32_t F;        /* feature channels                    */
    int32_t H;        /* image height                        */
    int32_t W;        /* image width                         */
    int32_t padding;  /* generation_params.padding           */
    int32_t grid[3];  /* voxel grid shape                    */
    float bbox[6];    /* min xyz, max xyz                    */
} rno_config;

static inline float clampf(float x, float a, float b) {
    /* utils.cu:1-3 */
    return fminf(fmaxf(x, a), b);
}

/* float -> int the way the device does it (saturating, NaN -> 0); plain C
 * leaves out-of-range conversions undefined. */
static inline int f2i_sat(float v) {
    if (!(v == v)) return 0;
    if (v >= 2147483520.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

/* ------------------------------------------------------------------ a1 -- */
/* sampling_schemes.cu:5-8 (unravel), :15-39 (dot_m43v3), :44-90 */
void rno_sample_in_bbox(const rno_config *c, int ray_idx, const float *P_inv,
                        const float *center, float *ray_start, float *ray_end) {
    float px = (float)(ray_idx / c->H);
    float py = (float)(ray_idx % c->H);

    /* 4x3 matrix times (px, py, 1): fp32 products summed in double */
    double o[4];
    for (int r = 0; r < 4; r++) {
        double a = 0.0;
        a += (double)(P_inv[3 * r + 0] * px);
        a += (double)(P_inv[3 * r + 1] * py);
        a += (double)P_inv[3 * r + 2] * 1.0;
        o[r] = a;
    }
    float dir[3];
    for (int i = 0; i < 3; i++) {
        double p = o[i] / o[3];
        dir[i] = (float)(p - (double)center[i]);
    }

    /* slab test; the bbox literals are doubles in the reference, so each
     * quotient is formed in double and rounded once to float (:65-77) */
    float t_near = -INFINITY, t_far = INFINITY;
    for (int i = 0; i < 3; i++) {
        float t1 = (float)(((double)c->bbox[i] - (double)center[i]) / (double)dir[i]);
        float t2 = (float)(((double)c->bbox[3 + i] - (double)center[i]) / (double)dir[i]);
        t_near = fmaxf(fminf(t1, t2), t_near);
        t_far = fminf(fmaxf(t1, t2), t_far);
    }
    /* :81-83 swap by |t| */
    float near_mask = (fabsf(t_near) < fabsf(t_far)) ? 1.0f : 0.0f;
    float tn = t_near * near_mask + t_far * (1 - near_mask);
    float tf = (1 - near_mask) * t_near + near_mask * t_far;
    for (int i = 0; i < 3; i++) {
        ray_start[i] = center[i] + tn * dir[i];
        ray_end[i] = center[i] + tf * dir[i];
    }
}

/* ------------------------------------------------------------------ a2 -- */
/* feature_similarities.cu:10-32 */
static inline void project34(const float *m, const float *v, float *out) {
    float x = 0.0f, y = 0.0f, n = 0.0f;
    x += m[0] * v[0]; x += m[1] * v[1]; x += m[2] * v[2]; x += m[3] * 1;
    y += m[4] * v[0]; y += m[5] * v[1]; y += m[6] * v[2]; y += m[7] * 1;
    n += m[8] * v[0]; n += m[9] * v[1]; n += m[10] * v[2]; n += m[11] * 1;
    out[0] = x / n;
    out[1] = y / n;
}

/* feature_similarities.cu:42-61; returns the (fx, fy) feature-map index */
static inline void pixel_to_features(const float *x, int *f_idx, int padding,
                                     int h, int w) {
    /* round() is half-away-from-zero; the sum is formed in float and
     * truncated on assignment to int */
    f_idx[0] = f2i_sat(roundf(x[0]) + padding - (padding - 1) / 2);
    f_idx[1] = f2i_sat(roundf(x[1]) + padding - (padding - 1) / 2);
    f_idx[0] = f_idx[0] > 0 ? f_idx[0] : 0;
    f_idx[0] = f_idx[0] < w ? f_idx[0] : w;
    f_idx[1] = f_idx[1] > 0 ? f_idx[1] : 0;
    f_idx[1] = f_idx[1] < h ? f_idx[1] : h;
    if (f_idx[0] == 0 || f_idx[1] == 0) f_idx[0] = f_idx[1] = 0;
}

/* Exposed for the index-parity tests: feature index of depth plane k in view v */
void rno_feature_index(const rno_config *c, const float *P, const float *ray_start,
                       const float *ray_end, int v, int k, int *f_idx) {
    float point[3], pix[2];
    for (int a = 0; a < 3; a++)
        point[a] = ray_start[a] + k * (ray_end[a] - ray_start[a]) / (c->D - 1);
    project34(P + 12 * v, point, pix);
    pixel_to_features(pix, f_idx, c->padding, c->H, c->W);
}

/* feature_similarities.cu:66-124.  S has D entries and is overwritten. */
void rno_similarities(const rno_config *c, const float *features, const float *P,
                      const float *ray_start, const float *ray_end, float *S) {
    const int fh = c->H + c->padding + 1, fw = c->W + c->padding + 1;
    const size_t dim_x = (size_t)fh * fw * c->F, dim_y = (size_t)fw * c->F;
    const int F = c->F, D = c->D, N = c->N;

    for (int k = 0; k < D; k++) S[k] = 0.0f; /* Q3 */
    for (int i = 0; i < N; i++) {
        for (int j = i + 1; j < N; j++) {
            for (int k = 0; k < D; k++) {
                float point[3], pi[2], pj[2];
                int fi[2], fj[2];
                for (int a = 0; a < 3; a++)
                    point[a] = ray_start[a] + k * (ray_end[a] - ray_start[a]) / (D - 1);
                project34(P + 12 * i, point, pi);
                project34(P + 12 * j, point, pj);
                pixel_to_features(pi, fi, c->padding, c->H, c->W);
                pixel_to_features(pj, fj, c->padding, c->H, c->W);
                const float *a = features + dim_x * i + dim_y * fi[1] + (size_t)F * fi[0];
                const float *b = features + dim_x * j + dim_y * fj[1] + (size_t)F * fj[0];
                float dot = 0.0f;
                for (int f = 0; f < F; f++) dot += a[f] * b[f];
                S[k] += dot;
            }
        }
    }
    /* :105-107 integer pair count */
    int pairs = (N * (N - 1)) / 2;
    for (int k = 0; k < D; k++) S[k] /= pairs;
    /* :109-123 stable softmax */
    float mx = -INFINITY;
    for (int k = 0; k < D; k++) mx = fmaxf(mx, S[k]);
    float sum = 0.0f;
    for (int k = 0; k < D; k++) {
        S[k] = expf(S[k] - mx);
        sum += S[k];
    }
    for (int k = 0; k < D; k++) S[k] /= sum;
}

/* ------------------------------------------------------------------ a3 -- */
/* ray_tracing.pyx:64-199.  voxels has room for M triples; returns the count. */
int rno_voxel_traversal(const float *bbox, const int32_t *grid, int M,
                        const float *ray_start, const float *ray_end,
                        int32_t *voxels) {
    const float EPS = 1e-2f;
    float s[3], e[3], bin[3], ray[3], tmax[3], tdelta[3];
    int step[3], cur[3], last[3];
    for (int i = 0; i < 3; i++) {
        s[i] = ray_start[i] - bbox[i];
        e[i] = ray_end[i] - bbox[i];
        bin[i] = (bbox[3 + i] - bbox[i]) / grid[i];
    }
    for (int i = 0; i < 3; i++) {
        ray[i] = e[i] - s[i];
        step[i] = ray[i] >= 0 ? 1 : -1;
    }
    for (int i = 0; i < 3; i++) {
        s[i] += step[i] * bin[i] * EPS;
        e[i] -= step[i] * bin[i] * EPS;
    }
    for (int i = 0; i < 3; i++) {
        cur[i] = (int)floorf(s[i] / bin[i]);
        last[i] = (int)floorf(e[i] / bin[i]);
    }
    for (int i = 0; i < 3; i++)
        if (cur[i] < 0 || cur[i] >= grid[i]) return 0;

    for (int i = 0; i < 3; i++) {
        tmax[i] = FLT_MAX;
        if (ray[i] != 0) {
            float cc = cur[i] * bin[i];
            if (step[i] < 0 && cc < s[i])
                tmax[i] = cc;
            else
                tmax[i] = cc + step[i] * bin[i];
            tmax[i] = (tmax[i] - s[i]) / ray[i];
        }
    }
    for (int i = 0; i < 3; i++)
        tdelta[i] = ray[i] != 0 ? step[i] * bin[i] / ray[i] : FLT_MAX;

    int ii = 0;
    voxels[0] = cur[0]; voxels[1] = cur[1]; voxels[2] = cur[2];
    ii = 1;
    while (!(cur[0] == last[0] && cur[1] == last[1] && cur[2] == last[2]) && ii < M) {
        int axis;
        if (tmax[0] < tmax[1])
            axis = (tmax[0] < tmax[2]) ? 0 : 2;
        else
            axis = (tmax[1] < tmax[2]) ? 1 : 2;
        cur[axis] += step[axis];
        if (cur[axis] < 0 || cur[axis] >= grid[axis]) return ii;
        tmax[axis] += tdelta[axis];
        voxels[3 * ii] = cur[0]; voxels[3 * ii + 1] = cur[1]; voxels[3 * ii + 2] = cur[2];
        ii++;
    }
    return ii;
}

/* ------------------------------------------------------------------ a4 -- */
/* planes_voxels_mapping.cu:6-92.  voxel_grid is [gx][gy][gz][3]. */
void rno_planes_to_voxels(const rno_config *c, const float *voxel_grid,
                          const int32_t *rvi, int count, const float *ray_start,
                          const float *ray_end, const float *S, float *S_new) {
    const float eps = 1e-4f;
    float ray[3], ray_norm = 0.0f;
    for (int i = 0; i < 3; i++) ray[i] = ray_end[i] - ray_start[i];
    for (int i = 0; i < 3; i++) ray_norm += ray[i] * ray[i];

    const float start = 0.0f, end = 1.0f;
    const float step = (end - start) / (c->D - 1);
    int left = 0, right = 1;
    const size_t dim_x = (size_t)3 * c->grid[1] * c->grid[2], dim_y = (size_t)3 * c->grid[2];
    float srsum = 0.0f;
    for (int i = 0; i < count; i++) {
        float sum = 0.0f;
        const float *vc = voxel_grid + rvi[3 * i] * dim_x + rvi[3 * i + 1] * dim_y +
                          (size_t)rvi[3 * i + 2] * 3;
        for (int j = 0; j < 3; j++) {
            float vd = vc[j];
            vd -= ray_start[j];
            sum += ray[j] * vd;
        }
        float t = clampf(sum / ray_norm, eps, 1 - eps);
        float left_d = t - (start + left * step);
        float right_d = t - (start + right * step);
        while (left_d > 0 && right_d > 0) {
            left++;
            right++;
            left_d = t - (start + left * step);
            right_d = t - (start + right * step);
        }
        left_d = fabsf(left_d);
        right_d = fabsf(right_d);
        float c1 = (float)(1.0 - (double)(left_d / (left_d + right_d)));
        float c2 = (float)(1.0 - (double)(right_d / (left_d + right_d)));
        S_new[i] = c1 * S[left] + c2 * S[right];
        srsum += S_new[i];
    }
    for (int i = 0; i < count; i++) S_new[i] = S_new[i] / srsum;
}

/* Exposed for index parity: the (left) plane index the walk assigns per voxel */
void rno_plane_indices(const rno_config *c, const float *voxel_grid,
                       const int32_t *rvi, int count, const float *ray_start,
                       const float *ray_end, int32_t *left_out) {
    const float eps = 1e-4f;
    float ray[3], ray_norm = 0.0f;
    for (int i = 0; i < 3; i++) ray[i] = ray_end[i] - ray_start[i];
    for (int i = 0; i < 3; i++) ray_norm += ray[i] * ray[i];
    const float step = (1.0f - 0.0f) / (c->D - 1);
    int left = 0, right = 1;
    const size_t dim_x = (size_t)3 * c->grid[1] * c->grid[2], dim_y = (size_t)3 * c->grid[2];
    for (int i = 0; i < count; i++) {
        float sum = 0.0f;
        const float *vc = voxel_grid + rvi[3 * i] * dim_x + rvi[3 * i + 1] * dim_y +
                          (size_t)rvi[3 * i + 2] * 3;
        for (int j = 0; j < 3; j++) {
            float vd = vc[j];
            vd -= ray_start[j];
            sum += ray[j] * vd;
        }
        float t = clampf(sum / ray_norm, eps, 1 - eps);
        float left_d = t - (0.0f + left * step);
        float right_d = t - (0.0f + right * step);
        while (left_d > 0 && right_d > 0) {
            left++;
            right++;
            left_d = t - (0.0f + left * step);
            right_d = t - (0.0f + right * step);
        }
        left_out[i] = left;
    }
}

/* ------------------------------------------------------------ a5 / a6 -- */
static inline size_t grid_index(const rno_config *c, const int32_t *v) {
    /* mrf_bp.cu:3-10 */
    return (size_t)c->grid[1] * c->grid[2] * v[0] + (size_t)c->grid[2] * v[1] + v[2];
}

static inline float occupancy_to_ray(float acc, float msg) {
    /* mrf_bp.cu:12-35 */
    float mu = acc - msg;
    float mx = fmaxf(0.0f, mu);
    float t1 = expf(0 - mx);
    float t2 = expf(mu - mx);
    return clampf(t2 / (t1 + t2), 1e-4f, (float)(1 - 1e-4));
}

static void clip_and_renorm(const float *S, int count, float *Sr) {
    /* mrf_bp.cu:103-111 == mrf_np.py:4-8 */
    float sum = 0.0f;
    for (int i = 0; i < count; i++) {
        Sr[i] = clampf(S[i], (float)1e-5, (float)(1 - 1e-5));
        sum += Sr[i];
    }
    for (int i = 0; i < count; i++) Sr[i] = Sr[i] / sum;
}

/* One BP sweep for one ray.  msgs_in and msgs_out may alias (the reference
 * always aliases them).  acc_out += message, element-wise, non-atomically
 * unless atomic != 0.  Sr is count floats of scratch. */
/* 0 (default): the reference's message arithmetic, mrf_bp.cu:136-167, to the letter.
 * 1: the numerically robust form the HIP kernels use -- the suffix sum
 *    sum_{j>i} w_j accumulated directly instead of (cumsum1 - cumsum2), and the message as
 *    log(pos) - log(neg) instead of log(p) - log(1 - p).  Same mathematics; it exists so that
 *    full-size runs, where the literal form overflows to +inf in some voxels, still have a
 *    finite CPU statement to be compared with (tools/fullsize_parity.py). */
static int g_robust_messages = 0;
void rno_set_robust_messages(int on) { g_robust_messages = on; }

void rno_bp_ray(const rno_config *c, const float *S, const int32_t *rvi, int count,
                const float *acc_in, const float *msgs_in, float *acc_out,
                float *msgs_out, float *Sr, int atomic) {
    if (count <= 1) return; /* Q4: mrf_np.py:300 */
    clip_and_renorm(S, count, Sr);
    if (g_robust_messages) {
        float *w = (float *)malloc(sizeof(float) * 3 * (size_t)count);
        float *ov = w + count, *Tp = w + 2 * count;
        float cumprod = 1.0f;
        for (int i = 0; i < count; i++) {
            ov[i] = occupancy_to_ray(acc_in[grid_index(c, rvi + 3 * i)], msgs_in[i]);
            Tp[i] = cumprod;
            w[i] = ov[i] * cumprod * Sr[i];
            cumprod *= (1.0f - ov[i]);
        }
        float suffix = 0.0f, prefix = 0.0f;
        for (int i = count - 1; i >= 0; i--) {       /* msgs_out[i] <- suffix, for now */
            msgs_out[i] = suffix;
            suffix += w[i];
        }
        for (int i = 0; i < count; i++) {
            float pos = prefix + Tp[i] * Sr[i];
            float neg = prefix + msgs_out[i] / (1.0f - ov[i]);
            prefix += w[i];
            msgs_out[i] = logf(pos) - logf(neg);
        }
        free(w);
        for (int i = 0; i < count; i++) {
            float *dst = acc_out + grid_index(c, rvi + 3 * i);
            if (atomic) {
#pragma omp atomic
                *dst += msgs_out[i];
            } else {
                *dst += msgs_out[i];
            }
        }
        return;
    }

    /* pass 1 (mrf_bp.cu:115-133): total of o_j * prod_{k<j}(1-o_k) * s_j */
    float cumsum1 = 0.0f, cumprod = 1.0f, cumprod_prev;
    for (int i = 0; i < count; i++) {
        float o = occupancy_to_ray(acc_in[grid_index(c, rvi + 3 * i)], msgs_in[i]);
        cumprod_prev = cumprod;
        cumprod *= (1.0f - o);
        cumsum1 += o * cumprod_prev * Sr[i];
    }
    /* pass 2 (mrf_bp.cu:136-167) */
    float cumsum2 = 0.0f, cumsum2_prev;
    cumprod = 1.0f;
    for (int i = 0; i < count; i++) {
        float o = occupancy_to_ray(acc_in[grid_index(c, rvi + 3 * i)], msgs_in[i]);
        cumprod_prev = cumprod;
        cumprod *= (1.0f - o);
        cumsum2_prev = cumsum2;
        cumsum2 += o * cumprod_prev * Sr[i];
        float pos = cumsum2_prev + cumprod_prev * Sr[i];
        float neg = cumsum2_prev + (cumsum1 - cumsum2) / (1.0f - o);
        pos = pos / (pos + neg);
        msgs_out[i] = logf(pos) - logf(1.0f - pos);
    }
    /* pass 3 (mrf_bp.cu:170-176) */
    for (int i = 0; i < count; i++) {
        float *dst = acc_out + grid_index(c, rvi + 3 * i);
        if (atomic) {
#pragma omp atomic
            *dst += msgs_out[i];
        } else {
            *dst += msgs_out[i];
        }
    }
}

/* mrf_bp.cu:37-86.  S_new gets M entries (zero beyond count, as the
 * reference's zero-filled buffers do). */
void rno_depth_ray(const rno_config *c, const float *S, const int32_t *rvi, int count,
                   const float *acc, const float *msgs, float *S_new, float *Sr) {
    for (int i = 0; i < c->M; i++) S_new[i] = 0.0f;
    if (count <= 1) return; /* Q4: mrf_np.py:376 */
    clip_and_renorm(S, count, Sr);
    float cumprod = 1.0f, cumprod_prev, sum = 0.0f;
    for (int i = 0; i < count; i++) {
        float o = occupancy_to_ray(acc[grid_index(c, rvi + 3 * i)], msgs[i]);
        cumprod_prev = cumprod;
        cumprod *= (1.0f - o);
        S_new[i] = o * cumprod_prev * Sr[i];
        sum += S_new[i];
    }
    for (int i = 0; i < count; i++) S_new[i] = S_new[i] / sum;
}

/* raynet_fp.py:193-226: first arg-max over all M entries, distance to camera */
float rno_depth_from_distribution(const rno_config *c, const float *S_new,
                                  const int32_t *rvi, const float *voxel_grid,
                                  const float *center) {
    float mx = -INFINITY;
    int idx = 0;
    for (int i = 0; i < c->M; i++) {
        if (S_new[i] > mx) {
            idx = i;
            mx = S_new[i];
        }
    }
    const size_t dim_x = (size_t)3 * c->grid[1] * c->grid[2], dim_y = (size_t)3 * c->grid[2];
    const float *p = voxel_grid + rvi[3 * idx] * dim_x + rvi[3 * idx + 1] * dim_y +
                     (size_t)rvi[3 * idx + 2] * 3;
    float sum = 0.0f;
    for (int i = 0; i < 3; i++) {
        float d = p[i] - center[i];
        sum += d * d;
    }
    return sqrtf(sum);
}

/* ------------------------------------------------ batch entry points -- */
static int pick_threads(int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

int rno_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* K8-prefix: start/end for a list of rays */
void rno_batch_sample(const rno_config *c, int n, const int32_t *ray_idxs,
                      const float *P_inv, const float *center, float *starts,
                      float *ends) {
    for (int r = 0; r < n; r++)
        rno_sample_in_bbox(c, ray_idxs[r], P_inv, center, starts + 3 * r, ends + 3 * r);
}

/* K7 */
void rno_batch_similarities(const rno_config *c, int n, const float *features,
                            const float *P, const float *starts, const float *ends,
                            float *S, int threads) {
    threads = pick_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64)
    for (int r = 0; r < n; r++)
        rno_similarities(c, features, P, starts + 3 * r, ends + 3 * r, S + (size_t)c->D * r);
}

/* K5: rvi [n,M,3] and rvc [n]; rows are zero-filled first like the driver does */
void rno_batch_traversal(const rno_config *c, int n, const float *starts,
                         const float *ends, int32_t *rvi, int32_t *rvc) {
    for (int r = 0; r < n; r++) {
        int32_t *row = rvi + (size_t)3 * c->M * r;
        memset(row, 0, sizeof(int32_t) * 3 * c->M);
        rvc[r] = rno_voxel_traversal(c->bbox, c->grid, c->M, starts + 3 * r, ends + 3 * r, row);
    }
}

/* K6: S [n,D] -> S_new [n,M] (zero beyond count) */
void rno_batch_planes_to_voxels(const rno_config *c, int n, const float *voxel_grid,
                                const int32_t *rvi, const int32_t *rvc,
                                const float *starts, const float *ends, const float *S,
                                float *S_new) {
    for (int r = 0; r < n; r++) {
        float *row = S_new + (size_t)c->M * r;
        memset(row, 0, sizeof(float) * c->M);
        rno_planes_to_voxels(c, voxel_grid, rvi + (size_t)3 * c->M * r, rvc[r], starts + 3 * r,
                             ends + 3 * r, S + (size_t)c->D * r, row);
    }
}

/* K3: one sweep over n rays */
void rno_batch_bp(const rno_config *c, int n, const float *S, const int32_t *rvi,
                  const int32_t *rvc, const float *acc_in, float *msgs, float *acc_out,
                  int threads) {
    threads = pick_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        float *Sr = (float *)malloc(sizeof(float) * c->M);
#pragma omp for schedule(dynamic, 64)
        for (int r = 0; r < n; r++) {
            float *m = msgs + (size_t)c->M * r;
            rno_bp_ray(c, S + (size_t)c->M * r, rvi + (size_t)3 * c->M * r, rvc[r], acc_in, m,
                       acc_out, m, Sr, threads > 1);
        }
        free(Sr);
    }
}

/* K4 */
void rno_batch_depth(const rno_config *c, int n, const float *S, const int32_t *rvi,
                     const int32_t *rvc, const float *acc, const float *msgs, float *S_new,
                     int threads) {
    threads = pick_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        float *Sr = (float *)malloc(sizeof(float) * c->M);
#pragma omp for schedule(dynamic, 64)
        for (int r = 0; r < n; r++)
            rno_depth_ray(c, S + (size_t)c->M * r, rvi + (size_t)3 * c->M * r, rvc[r], acc,
                          msgs + (size_t)c->M * r, S_new + (size_t)c->M * r, Sr);
        free(Sr);
    }
}

/* raynet_fp.py:55-104: sample -> similarities -> traversal -> mapping for one ray.
 * Writes rvi row (zero-filled first), returns count, fills S_voxel row [M]. */
static int prefix_ray(const rno_config *c, int ray_idx, const float *features,
                      const float *P, const float *P_inv, const float *center,
                      const float *voxel_grid, int32_t *rvi_row, float *Sv_row, float *Sd) {
    float rs[3], re[3];
    rno_sample_in_bbox(c, ray_idx, P_inv, center, rs, re);
    rno_similarities(c, features, P, rs, re, Sd);
    memset(rvi_row, 0, sizeof(int32_t) * 3 * c->M);
    memset(Sv_row, 0, sizeof(float) * c->M);
    int count = rno_voxel_traversal(c->bbox, c->grid, c->M, rs, re, rvi_row);
    rno_planes_to_voxels(c, voxel_grid, rvi_row, count, rs, re, Sd, Sv_row);
    return count;
}

/* Multi-threaded K1: where the threads' messages are summed.  0 (default): into the caller's ONE
 * accumulator with `omp atomic` adds, the way the reference's kernel adds into its one array
 * (mrf_bp.cu:170-176) -- its cache lines bounce between the cores.  1: every thread adds into a
 * private zero-started copy, the copies are summed into the caller's array at the end (what an
 * "all host cores" CPU figure should be measured with: bench.py's cpu_baseline reports both). */
static int g_private_accumulators = 0;
void rno_set_private_accumulators(int on) { g_private_accumulators = on; }

/* K1 (raynet_fp.py:106-149) */
void rno_fused_bp(const rno_config *c, int n, const int32_t *ray_idxs, const float *features,
                  const float *P, const float *P_inv, const float *center,
                  const float *voxel_grid, int32_t *rvi, int32_t *rvc, float *S_voxel,
                  const float *acc_in, float *msgs, float *acc_out, int threads) {
    threads = pick_threads(threads);
    const size_t G = (size_t)c->grid[0] * c->grid[1] * c->grid[2];
    const int private_acc = g_private_accumulators && threads > 1;
    float **parts = private_acc ? (float **)calloc((size_t)threads, sizeof(float *)) : NULL;
#pragma omp parallel num_threads(threads)
    {
        float *Sr = (float *)malloc(sizeof(float) * c->M);
        float *Sd = (float *)malloc(sizeof(float) * c->D);
        float *mine = acc_out;
#ifdef _OPENMP
        if (private_acc) mine = parts[omp_get_thread_num()] = (float *)calloc(G, sizeof(float));
#endif
#pragma omp for schedule(dynamic, 64)
        for (int r = 0; r < n; r++) {
            int32_t *row = rvi + (size_t)3 * c->M * r;
            float *sv = S_voxel + (size_t)c->M * r;
            float *m = msgs + (size_t)c->M * r;
            rvc[r] = prefix_ray(c, ray_idxs[r], features, P, P_inv, center, voxel_grid, row, sv, Sd);
            rno_bp_ray(c, sv, row, rvc[r], acc_in, m, mine, m, Sr, threads > 1 && !private_acc);
        }
        free(Sr);
        free(Sd);
        if (private_acc) {
#pragma omp for schedule(static)
            for (size_t i = 0; i < G; i++) {
                float sum = 0.0f;
                for (int t = 0; t < threads; t++)
                    if (parts[t]) sum += parts[t][i];
                acc_out[i] += sum;
            }
        }
    }
    if (private_acc) {
        for (int t = 0; t < threads; t++) free(parts[t]);
        free(parts);
    }
}

/* K2 (raynet_fp.py:151-227); S_voxel receives the final distribution */
void rno_fused_depth(const rno_config *c, int n, const int32_t *ray_idxs,
                     const float *features, const float *P, const float *P_inv,
                     const float *center, const float *voxel_grid, int32_t *rvi,
                     int32_t *rvc, float *S_voxel, const float *acc, const float *msgs,
                     float *depth_map, int threads) {
    threads = pick_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        float *Sr = (float *)malloc(sizeof(float) * c->M);
        float *Sd = (float *)malloc(sizeof(float) * c->D);
        float *Sn = (float *)malloc(sizeof(float) * c->M);
#pragma omp for schedule(dynamic, 64)
        for (int r = 0; r < n; r++) {
            int32_t *row = rvi + (size_t)3 * c->M * r;
            float *sv = S_voxel + (size_t)c->M * r;
            rvc[r] = prefix_ray(c, ray_idxs[r], features, P, P_inv, center, voxel_grid, row, sv, Sd);
            rno_depth_ray(c, sv, row, rvc[r], acc, msgs + (size_t)c->M * r, Sn, Sr);
            memcpy(sv, Sn, sizeof(float) * c->M);
            depth_map[r] = rno_depth_from_distribution(c, sv, row, voxel_grid, center);
        }
        free(Sr);
        free(Sd);
        free(Sn);
    }
}
