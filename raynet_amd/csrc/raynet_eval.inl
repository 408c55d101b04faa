// raynet_eval.inl -- depth maps -> point cloud -> accuracy / completeness (SURVEY.md 8f
// row 3): the consumers of the hot path's output.  Included at the end of raynet_hip.hip.
//
//   k_depth_points     raynet/pointcloud.py:121-147  back-projection of a depth map
//   k_consistency_tau  raynet/pointcloud.py:205-245  multi-view consistency of the points
//   k_nn               raynet/pointcloud.py:63-72 + metrics.py:155-236: the nearest
//                      neighbour search behind Accuracy / Completeness.  The reference
//                      builds a KD-tree on the host; here every query scans all reference
//                      points from LDS tiles -- exact, branch-free and at MI355X's fp32
//                      rate quicker than building any index for the ~10^6-point clouds
//                      five depth maps give.
// The first two are float64 like the NumPy code they restate (the inputs are float64
// camera matrices and float32 depths); k_nn works on float32 copies.

namespace {

// points[0..2][i] for pixel i = u*H + v (column-major, common/image.py:252-255), all pixels;
// the caller selects (borders, ground-truth mask) afterwards
__global__ __launch_bounds__(BLOCK) void k_depth_points(int H, int W,
                                                        const double *__restrict__ P_pinv,
                                                        const double *__restrict__ center,
                                                        const float *__restrict__ depth,
                                                        double *__restrict__ points) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int n = H * W;
    if (i >= n) return;
    const int u = i / H, v = i % H;
    // rays = project(P_pinv, (u, v, 1)): (4x3) x pixel, normalised by the last coordinate
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        r[k] = P_pinv[3 * k] * (double)u + P_pinv[3 * k + 1] * (double)v + P_pinv[3 * k + 2] * 1.0;
    double d[4], norm = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d[k] = r[k] / r[3] - center[k];
        norm += d[k] * d[k];
    }
    norm = sqrt(norm);
    const double t = (double)depth[(size_t)v * W + u];
#pragma unroll
    for (int k = 0; k < 3; k++) points[(size_t)k * n + i] = center[k] + t * d[k] / norm;
}

__device__ __forceinline__ double np_maximum(double a, double b) {
    return (a != a || b != b) ? (a + b) : (a > b ? a : b);     // NaN-propagating, like np.maximum
}

// one neighbour view of the consistency check: tau = max(tau, |depth_i(pixel) - dist|),
// inf where the point projects outside the view (pointcloud.py:218-242)
__global__ __launch_bounds__(BLOCK) void k_consistency_tau(int n, int H, int W, int first,
                                                           const double *__restrict__ points,
                                                           const double *__restrict__ P,
                                                           const double *__restrict__ center,
                                                           const float *__restrict__ depth,
                                                           double *tau) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double p[4] = {points[i], points[(size_t)n + i], points[2 * (size_t)n + i], 1.0};
    double h[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        h[k] = P[4 * k] * p[0] + P[4 * k + 1] * p[1] + P[4 * k + 2] * p[2] + P[4 * k + 3] * p[3];
    // np.round (half to even) then int32
    const int x = (int)rint(h[0] / h[2]), y = (int)rint(h[1] / h[2]);
    const bool valid = 0 <= x && x < W && 0 <= y && y < H;
    const double predicted = (double)depth[valid ? (size_t)y * W + x : 0];
    double dist = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double d = p[k] - center[k];
        dist += d * d;
    }
    dist = sqrt(dist);
    const double diff = fabs(predicted - dist);
    double t = first ? diff : np_maximum(diff, tau[i]);
    if (!valid) t = INFINITY;
    tau[i] = t;
}

// exact nearest neighbour of every query among n_ref reference points, brute force:
// reference points stream through LDS tiles, every lane keeps NN_Q queries in registers
constexpr int NN_TILE = 2048;
constexpr int NN_Q = 2;
__global__ __launch_bounds__(BLOCK) void k_nn(int n_ref, const float4 *__restrict__ ref, int n_q,
                                              const float4 *__restrict__ query, float *dist,
                                              int32_t *idx) {
    __shared__ float4 tile[NN_TILE];
    const int tid = threadIdx.x;
    float4 me[NN_Q];
    float best[NN_Q];
    int bi[NN_Q];
#pragma unroll
    for (int q = 0; q < NN_Q; q++) {
        const int i = (blockIdx.x * NN_Q + q) * BLOCK + tid;
        me[q] = i < n_q ? query[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        best[q] = INFINITY;
        bi[q] = -1;
    }
    for (int base = 0; base < n_ref; base += NN_TILE) {
        const int cnt = min(NN_TILE, n_ref - base);
        __syncthreads();
        for (int j = tid; j < cnt; j += BLOCK) tile[j] = ref[base + j];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < cnt; j++) {
            const float4 r = tile[j];             // same address in every lane: LDS broadcast
#pragma unroll
            for (int q = 0; q < NN_Q; q++) {
                const float dx = me[q].x - r.x, dy = me[q].y - r.y, dz = me[q].z - r.z;
                const float d2 = dx * dx + dy * dy + dz * dz;
                if (d2 < best[q]) {               // strict: the first of equal distances wins
                    best[q] = d2;
                    bi[q] = base + j;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NN_Q; q++) {
        const int i = (blockIdx.x * NN_Q + q) * BLOCK + tid;
        if (i < n_q) {
            if (dist) dist[i] = sqrtf(best[q]);
            if (idx) idx[i] = bi[q];
        }
    }
}

}  // namespace

extern "C" {

int rn_depthmap_points(rn_ctx *ctx, int32_t H, int32_t W, const double *P_pinv,
                       const double *camera_center, const float *depth_map, double *points,
                       void *stream) {
    if (!ctx || H < 1 || W < 1 || !P_pinv || !camera_center || !depth_map || !points)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_depth_points, dim3(thread_blocks(H * W)), dim3(BLOCK), 0, S(stream), H, W,
                       P_pinv, camera_center, depth_map, points);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_consistency_tau(rn_ctx *ctx, int32_t n, int32_t H, int32_t W, int32_t first,
                       const double *points, const double *P, const double *camera_center,
                       const float *depth_map, double *tau, void *stream) {
    if (ctx && n == 0) return RN_OK;
    if (!ctx || n < 0 || H < 1 || W < 1 || !points || !P || !camera_center || !depth_map || !tau)
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    hipLaunchKernelGGL(k_consistency_tau, dim3(thread_blocks(n)), dim3(BLOCK), 0, S(stream), n, H, W,
                       first, points, P, camera_center, depth_map, tau);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

int rn_nearest_neighbors(rn_ctx *ctx, int32_t n_ref, const float *ref_xyzw, int32_t n_query,
                         const float *query_xyzw, float *dist, int32_t *idx, void *stream) {
    if (ctx && n_query == 0) return RN_OK;
    if (!ctx || n_ref < 1 || n_query < 0 || !ref_xyzw || !query_xyzw || (!dist && !idx))
        return fail(ctx, RN_ERR_INVALID, "bad argument");
    const int blocks = (n_query + BLOCK * NN_Q - 1) / (BLOCK * NN_Q);
    hipLaunchKernelGGL(k_nn, dim3(blocks), dim3(BLOCK), 0, S(stream), n_ref,
                       reinterpret_cast<const float4 *>(ref_xyzw), n_query,
                       reinterpret_cast<const float4 *>(query_xyzw), dist, idx);
    RN_LAUNCH_CHECK(ctx);
    return RN_OK;
}

}  // extern "C"
