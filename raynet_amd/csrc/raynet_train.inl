// raynet_train.inl -- differentiable MRF block for training: the
// analytic backward of one BP sweep and of the depth distribution.  Included at the end of
// raynet_hip.hip (same translation unit: rn_ctx, launch helpers, wave primitives in scope).
//
// The reference differentiates the unrolled graph with TensorFlow autodiff
// (raynet/tf_implementations/forward_backward_pass.py:214-246, raynet/mrf/mrf_tf.py); there
// is no hand-written backward to restate.  k_ray_bwd is the reverse-mode derivative of
// k_bp / k_depth; oracle/mrf_backward.py states the same in float64
// and is checked against finite differences.
//
// One wavefront per ray (one per workgroup); per-voxel quantities live in LDS rows of M
// floats, the prefix / suffix recurrences are chunked wave scans over those rows.  Training
// batches are a few thousand rays: clarity over the last percent.

namespace {

// out[i] = sum_{j<i} in[j]  (REVERSE: sum_{j>i}); returns the total.  in may alias out.
template <bool REVERSE>
__device__ __forceinline__ float lds_excl_sum(const float *in, float *out, int c, int lane) {
    float carry = 0.0f;
    const int nchunk = (c + WAVE - 1) / WAVE;
    for (int k = 0; k < nchunk; k++) {
        const int ch = REVERSE ? nchunk - 1 - k : k;
        const int i = ch * WAVE + lane;
        const float x = i < c ? in[i] : 0.0f;
        float excl, tot;
        if (REVERSE) {
            excl = wave_suffix_excl(x, lane, tot);
        } else {
            const float incl = wave_scan_add(x);
            excl = wave_shift1(incl, 0.0f);
            tot = lane63(incl);
        }
        if (i < c) out[i] = carry + excl;
        carry += tot;
    }
    return carry;
}
// out[i] = prod_{k<i} in[k]
__device__ __forceinline__ void lds_excl_prod(const float *in, float *out, int c, int lane) {
    float carry = 1.0f;
    for (int base = 0; base < c; base += WAVE) {
        const int i = base + lane;
        const float incl = wave_scan_mul(i < c ? in[i] : 1.0f);
        if (i < c) out[i] = carry * wave_shift1(incl, 1.0f);
        carry *= lane63(incl);
    }
}

constexpr int TRAIN_ROWS = 14;   // LDS rows of M floats per wavefront

// MODE 0: backward of one BP sweep;  MODE 1: backward of the depth distribution.
//   s          [n][M]  clipped + renormalised column (forward input)
//   acc        [G]     accumulator the forward read;  msgs [n][M] messages the forward read
//   g_out      [n][M]  MODE 0: dL/d(new messages) from their later direct use
//                      MODE 1: dL/d(depth distribution)
//   g_acc_next [G]     MODE 0: dL/d(accumulator built from the new messages); may be null
//   g_s        [n][M]  += dL/ds        g_acc [G] += dL/d(acc) (atomic)
//   g_msgs     [n][M]  =  dL/d(msgs the forward read)
template <int MODE, bool PACKED>
__global__ __launch_bounds__(WAVE) void k_ray_bwd(Params p, int n, const float *__restrict__ s_in,
                                                  const int32_t *__restrict__ vox,
                                                  const int32_t *__restrict__ rvc,
                                                  const float *__restrict__ acc,
                                                  const float *__restrict__ msgs,
                                                  const float *__restrict__ g_out,
                                                  const float *__restrict__ g_acc_next, float *g_s,
                                                  float *g_acc, float *g_msgs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    if (r >= n) return;
    const int M = p.M;
    float *S_ = smem, *O_ = smem + M, *Q_ = smem + 2 * M, *T_ = smem + 3 * M, *W_ = smem + 4 * M,
          *DS_ = smem + 5 * M, *G_ = smem + 6 * M, *C_ = smem + 7 * M, *U_ = smem + 8 * M,
          *CB_ = smem + 9 * M, *UB_ = smem + 10 * M, *TB_ = smem + 11 * M, *SB_ = smem + 12 * M,
          *QB_ = smem + 13 * M;
    const int c = min(uniform(rvc[r]), M);
    float *gm_row = g_msgs + (size_t)r * M;
    if (c <= 1) {            // skipped ray (mrf_np.py:300): no message, no gradient
        for (int i = lane; i < c; i += WAVE) gm_row[i] = 0.0f;
        return;
    }
    const int32_t *vrow = vox + (size_t)r * M * (PACKED ? 1 : 3);

    // ---- forward quantities of this ray
    for (int i = lane; i < c; i += WAVE) {
        const int lin = lin_of<false>(p, load_packed<PACKED>(vrow, i));
        const float mu = acc[lin] - msgs[(size_t)r * M + i];
        const float e = expf(0 - fabsf(mu));
        const float sig = (mu > 0.0f ? 1.0f : e) / (1.0f + e);
        const float o = clampf(sig, 1e-4f, (float)(1 - 1e-4));
        S_[i] = s_in[(size_t)r * M + i];
        O_[i] = o;
        Q_[i] = 1.0f - o;
        // d(o)/d(mu): sigma'(mu) inside the clamp, 0 where the clamp is active
        DS_[i] = (sig > 1e-4f && sig < (float)(1 - 1e-4)) ? sig * (1.0f - sig) : 0.0f;
        float g = g_out[(size_t)r * M + i];
        if (MODE == 0 && g_acc_next) g += g_acc_next[lin];
        G_[i] = g;
    }
    wave_sync();
    lds_excl_prod(Q_, T_, c, lane);
    wave_sync();
    for (int i = lane; i < c; i += WAVE) W_[i] = O_[i] * T_[i] * S_[i];
    wave_sync();

    // ---- dL/dw into C_, partial dL/dT, dL/ds, dL/dq into TB_, SB_, QB_
    if (MODE == 0) {
        lds_excl_sum<false>(W_, C_, c, lane);          // C_i = sum_{j<i} w_j
        lds_excl_sum<true>(W_, U_, c, lane);           // U_i = sum_{j>i} w_j
        wave_sync();
        for (int i = lane; i < c; i += WAVE) {
            const float q = Q_[i], g = G_[i];
            const float pos = C_[i] + T_[i] * S_[i];
            const float neg = C_[i] + U_[i] / q;
            const float pbar = g / pos, nbar = -g / neg;
            CB_[i] = pbar + nbar;
            UB_[i] = nbar / q;
            TB_[i] = pbar * S_[i];
            SB_[i] = pbar * T_[i];
            QB_[i] = -nbar * U_[i] / (q * q);
        }
        wave_sync();
        lds_excl_sum<true>(CB_, C_, c, lane);          // sum_{i>j} Cbar_i
        lds_excl_sum<false>(UB_, U_, c, lane);         // sum_{i<j} Ubar_i
        wave_sync();
        for (int i = lane; i < c; i += WAVE) C_[i] = C_[i] + U_[i];      // dL/dw
    } else {
        // d = w / W:  dL/dw_i = (g_i - sum_j g_j d_j) / W
        float wsum = 0.0f, dot = 0.0f;
        for (int i = lane; i < c; i += WAVE) {
            wsum += W_[i];
            dot += G_[i] * W_[i];
        }
        wsum = wave_sum(wsum);
        dot = wave_sum(dot) / wsum;
        for (int i = lane; i < c; i += WAVE) {
            C_[i] = (G_[i] - dot) / wsum;
            TB_[i] = 0.0f;
            SB_[i] = 0.0f;
            QB_[i] = 0.0f;
        }
    }
    wave_sync();

    // ---- w = o T s;  T_j = prod_{k<j} q_k;  q = 1 - o;  o = clamp(sigmoid(mu))
    for (int i = lane; i < c; i += WAVE) {
        const float wbar = C_[i];
        const float tb = TB_[i] + wbar * O_[i] * S_[i];
        SB_[i] = SB_[i] + wbar * O_[i] * T_[i];
        CB_[i] = wbar * T_[i] * S_[i];                 // dL/do from w
        UB_[i] = tb * T_[i];                           // Tbar_j T_j
    }
    wave_sync();
    lds_excl_sum<true>(UB_, U_, c, lane);              // sum_{j>k} Tbar_j T_j
    wave_sync();
    for (int i = lane; i < c; i += WAVE) {
        const float qbar = QB_[i] + U_[i] / Q_[i];
        const float mubar = (CB_[i] - qbar) * DS_[i];
        const size_t off = (size_t)r * M + i;
        g_s[off] = g_s[off] + SB_[i];
        gm_row[i] = -mubar;
        const int lin = lin_of<false>(p, load_packed<PACKED>(vrow, i));
        __hip_atomic_fetch_add(g_acc + lin, mubar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// left plane index and the two interpolation weights of every traversed voxel, i.e. the
// data-independent part of planes_voxels_mapping.cu:48-84 (the differentiable part,
// S_voxel = normalise(c1 S[left] + c2 S[left+1]), is dense glue done by the framework)
template <bool PACKED>
__global__ __launch_bounds__(BLOCK) void k_plane_weights(Params p, int n,
                                                         const float *__restrict__ axes_g,
                                                         const int32_t *__restrict__ vox,
                                                         const int32_t *__restrict__ rvc,
                                                         const float *__restrict__ starts,
                                                         const float *__restrict__ ends,
                                                         int32_t *left_out, float *c1_out,
                                                         float *c2_out) {
    int lane;
    const int r = ray_of_wave(n, lane);
    if (r < 0) return;
    const int count = min(uniform(rvc[r]), p.M);
    const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
    float s[3], e[3], ray[3], ray_norm = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        s[i] = starts[3 * r + i];
        e[i] = ends[3 * r + i];
        ray[i] = e[i] - s[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) ray_norm += ray[i] * ray[i];
    const float eps = 1e-4f;
    const float step = (1.0f - 0.0f) / (p.D - 1);
    int carry = 0;
    for (int base = 0; base < count; base += WAVE) {
        const int i = base + lane;
        const bool valid = i < count;
        int L = 0;
        float t = 0.0f;
        if (valid) {
            int x, y, z;
            load_voxel<PACKED>(vrow, i, x, y, z);
            float sum = 0.0f;
            float vd = axes_g[x];
            vd -= s[0];
            sum += ray[0] * vd;
            vd = axes_g[p.gx + y];
            vd -= s[1];
            sum += ray[1] * vd;
            vd = axes_g[p.gx + p.gy + z];
            vd -= s[2];
            sum += ray[2] * vd;
            t = clampf(sum / ray_norm, eps, 1 - eps);
            L = max(0, (int)(t * (p.D - 1)) - 1);      // (see map_planes_to_voxels)
            while ((t - (0.0f + L * step) > 0) && (t - (0.0f + (L + 1) * step) > 0)) L++;
        }
        const int left = max(wave_scan_max(valid ? L : 0), carry);
        carry = lane63i(left);
        if (valid) {
            const float left_d = fabsf(t - (0.0f + left * step));
            const float right_d = fabsf(t - (0.0f + (left + 1) * step));
            const size_t off = (size_t)r * p.M + i;
            left_out[off] = left;
            c1_out[off] = 1.0f - (left_d / (left_d + right_d));
            c2_out[off] = 1.0f - (right_d / (left_d + right_d));
        }
    }
}

}  // namespace

extern "C" {

/* Training forward on a pre-normalised column: rn_bp_sweep / rn_depth_estimation without
 * their internal clip_and_renorm (the framework does that step differentiably). */
int rn_train_bp_sweep(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                      const int32_t *rvc, const float *acc_in, const float *msgs_in,
                      float *acc_out, float *msgs_out, void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !Sr || !rvi || !rvc || !acc_in || !acc_out || !msgs_out)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    return launch_bp<false, false>(ctx, n, Sr, rvi, rvc, acc_in, msgs_in, acc_out, msgs_out,
                                   S(stream));
}

int rn_train_depth(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                   const int32_t *rvc, const float *acc, const float *msgs, float *S_new,
                   void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !Sr || !rvi || !rvc || !acc || !msgs || !S_new)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    return launch_depth<false, false>(ctx, n, Sr, rvi, rvc, acc, msgs, nullptr, S_new, nullptr,
                                      S(stream));
}

int rn_train_bp_sweep_bwd(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                          const int32_t *rvc, const float *acc_in, const float *msgs_in,
                          const float *g_msgs_out, const float *g_acc_out, float *g_Sr,
                          float *g_acc_in, float *g_msgs_in, void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !Sr || !rvi || !rvc || !acc_in || !msgs_in || !g_msgs_out || !g_Sr ||
        !g_acc_in || !g_msgs_in)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    const size_t lds = sizeof(float) * TRAIN_ROWS * ctx->p.M;
    if (lds > 64 * 1024) return fail(ctx, RN_ERR_INVALID, "M > 1170 not supported by the training kernels");
    hipLaunchKernelGGL((k_ray_bwd<0, false>), dim3(n), dim3(WAVE), lds, S(stream), ctx->p, n, Sr, rvi,
                       rvc, acc_in, msgs_in, g_msgs_out, g_acc_out, g_Sr, g_acc_in, g_msgs_in);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_train_depth_bwd(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                       const int32_t *rvc, const float *acc, const float *msgs,
                       const float *g_S_new, float *g_Sr, float *g_acc, float *g_msgs,
                       void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !Sr || !rvi || !rvc || !acc || !msgs || !g_S_new || !g_Sr || !g_acc ||
        !g_msgs)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    const size_t lds = sizeof(float) * TRAIN_ROWS * ctx->p.M;
    if (lds > 64 * 1024) return fail(ctx, RN_ERR_INVALID, "M > 1170 not supported by the training kernels");
    hipLaunchKernelGGL((k_ray_bwd<1, false>), dim3(n), dim3(WAVE), lds, S(stream), ctx->p, n, Sr, rvi,
                       rvc, acc, msgs, g_S_new, (const float *)nullptr, g_Sr, g_acc, g_msgs);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_plane_weights(rn_ctx *ctx, int32_t n, const int32_t *rvi, const int32_t *rvc,
                     const float *ray_start, const float *ray_end, int32_t *left, float *c1,
                     float *c2, void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || !rvi || !rvc || !ray_start || !ray_end || !left || !c1 || !c2)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    int rc = need_axes(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL((k_plane_weights<false>), dim3(ray_blocks(n)), dim3(BLOCK), 0, S(stream),
                       ctx->p, n, ctx->axes, rvi, rvc, ray_start, ray_end, left, c1, c2);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

}  // extern "C"
