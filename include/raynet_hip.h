/*
 * raynet_hip.h -- C ABI of libraynet_hip.so: RayNet's forward_pass hot path on
 * MI355X (gfx950).  This is the drop-in boundary.
 *
 * The reference has no C ABI: its operator interface is the set of PyCUDA
 * closure factories in raynet/cuda_implementations/ (one .py per kernel family), each of which wraps one
 * `prepared_call` of a __global__ kernel.  Every entry point below replaces one
 * of those launches (kernel numbers K1..K12 as in SURVEY.md section 2.2); the
 * comment above each function names the reference launcher (file:line) it
 * stands in for.  INTEGRATION.md shows the ctypes stub a maintainer would put
 * in place of each PyCUDA closure.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the name ends in _host;
 *   - the caller allocates every buffer and keeps it alive; the library never
 *     allocates per call and never frees caller memory (reference: `to_gpu` in
 *     the driver, raynet/forward_pass.py:515-538);
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as
 *     void*; NULL = the null stream); no hidden global state, one rn_ctx per
 *     device;
 *   - return value: RN_OK or a negative rn_status; no exceptions cross the
 *     boundary.  The Python mirror re-raises shape/dtype problems as
 *     AssertionError like the reference (raynet_fp.py:291-301);
 *   - msgs_in and msgs_out may alias (the reference always aliases them,
 *     raynet_fp.py:321-323, mrf_cuda.py:73-75);
 *   - there is NO CPU fallback in this library.  rn_create fails with
 *     RN_ERR_NO_DEVICE when no gfx950 device is visible.
 *
 * Array layouts are the reference's:
 *   features   [N][H+padding+1][W+padding+1][F] f32   (forward_pass.py:622-641)
 *   P          [N][3][4] f32, P_inv [4][3] f32, camera_center [4] f32
 *   voxel_grid [gx][gy][gz][3] f32                     (forward_pass.py:573-575)
 *   ray_idxs   [n] i32, ray_idx = x*H + y              (sampling_schemes.cu:5-8)
 *   rvi        [n][M][3] i32, rvc [n] i32
 *   S (planes) [n][D] f32, S_voxel / msgs / S_new [n][M] f32
 *   acc        [gx][gy][gz] f32 (log-odds)
 */
#ifndef RAYNET_HIP_H
#define RAYNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RN_OK = 0,
    RN_ERR_INVALID = -1,     /* bad argument / unsupported size          */
    RN_ERR_HIP = -2,         /* a HIP runtime call failed                */
    RN_ERR_NO_DEVICE = -3,   /* no usable gfx950 device                  */
    RN_ERR_STATE = -4        /* call order (e.g. voxel grid not set)     */
} rn_status;

/* The values the reference bakes into its kernels as $literals
 * (cuda_implementations/raynet_fp.py:230-248); here they are run-time. */
typedef struct {
    int32_t M;        /* generation_params.max_number_of_marched_voxels */
    int32_t D;        /* depth_planes                                    */
    int32_t N;        /* neighbors + 1                                   */
    int32_t F;        /* feature channels                                */
    int32_t H;        /* image height                                    */
    int32_t W;        /* image width                                     */
    int32_t padding;  /* generation_params.padding                       */
    int32_t grid[3];  /* voxel grid shape                                */
    float bbox[6];    /* scene.bbox: min xyz, max xyz                    */
    int32_t device;   /* HIP device ordinal                              */
} rn_config;

typedef struct rn_ctx rn_ctx;

/* perform_raynet_fp(M,D,N,F,H,W,padding,bbox,grid_shape,"sample_in_bbox")
 * (raynet_fp.py:10-21): where the reference JIT-compiles, this validates the
 * configuration and creates a context. */
int rn_create(const rn_config *cfg, rn_ctx **out);
void rn_destroy(rn_ctx *ctx);

/* Behaviour options of a context -- schedule and A/B knobs; no result depends on them beyond
 * the summation order of the accumulator scatter.  The Python mirror is
 * raynet_amd.hip_implementations.options.PathOptions; the reference has no counterpart (its
 * only launch knob is `threads=2048`, raynet_fp.py:288).  rn_create starts from the defaults
 * below, overridden by the environment variables named here (read there and nowhere else). */
typedef struct {
    int32_t scatter_mode;   /* -1 by row layout (default), 0 slab scatter, 2 LDS-box scatter
                               [RAYNET_HIP_SCATTER_MODE] */
    int32_t box_level;      /* tile shape the adaptive box scatter starts from: 0 (default) 128
                               rays x 32 steps, 1: 256 x 16, 2: slab scatter [RAYNET_HIP_BOX_LEVEL] */
    int32_t box_pin;        /* != 0: stay at box_level [RAYNET_HIP_BOX_PIN] */
    int32_t overlap;        /* second stream (scatter || BP halves, traversal || plane sweep): 0 off,
                               1 on, 2 (default) when the scatter runs at level 1 [RAYNET_HIP_OVERLAP] */
    int32_t generic_sweep;  /* != 0: the reference-order plane sweep even for F = 32
                               [RAYNET_HIP_GENERIC_SWEEP] */
    int32_t sweep_rays_per_wave;  /* rays one wavefront of the cooperative plane sweep takes: 0
                               (default) by D -- 4 for D <= 16, 2 for D <= 32, else 1; 1: always one
                               (the same bits, slower for D <= 32) [RAYNET_HIP_SWEEP_RAYS_PER_WAVE] */
} rn_options;
int rn_get_options(const rn_ctx *ctx, rn_options *out);
/* also restarts the adaptive scatter (rn_scatter_reset) */
int rn_set_options(rn_ctx *ctx, const rn_options *opt);
const char *rn_last_error(const rn_ctx *ctx);
const char *rn_version(void);

/* Voxel centres are read by K1/K2/K6/K11/K12 from the [gx][gy][gz][3] array
 * (planes_voxels_mapping.cu:52-57).  The grid is separable, so the context keeps
 * the three per-axis centre tables extracted from the caller's array (bit-equal
 * values, 1.5 KB instead of 25 MB of gathers).  Call once per scene. */
int rn_set_voxel_grid(rn_ctx *ctx, const float *voxel_grid, void *stream);

/* GPUArray.fill (forward_pass.py:646-648, :678) */
int rn_fill_f32(rn_ctx *ctx, float *dst, int64_t count, float value, void *stream);
int rn_fill_i32(rn_ctx *ctx, int32_t *dst, int64_t count, int32_t value, void *stream);

/* ---- stand-alone stages ------------------------------------------------ */

/* sample_in_bbox for a list of rays -> ray_start/ray_end [n][3]
 * (sampling_schemes.cu:44-90; what every fused kernel does first). */
int rn_sample_rays(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *P_inv,
                   const float *camera_center, float *ray_start, float *ray_end,
                   void *stream);

/* K8 batch_sample_points_in_bbox, launcher sample_points.py:12-54:
 * points [n][D][4] */
int rn_sample_points(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *P_inv,
                     const float *camera_center, float *points, void *stream);

/* K7 batch_compute_similarities (feature_similarities.cu:126-146): S [n][D] */
int rn_compute_similarities(rn_ctx *ctx, int32_t n, const float *features, const float *P,
                            const float *ray_start, const float *ray_end, float *S,
                            void *stream);

/* K5 batch_voxel_traversal, launcher ray_tracing_cuda.py:35-63 */
int rn_voxel_traversal(rn_ctx *ctx, int32_t n, const float *ray_start, const float *ray_end,
                       int32_t *rvi, int32_t *rvc, void *stream);

/* K6 batch_planes_voxels_mapping, launcher planes_voxels_mapping_cuda.py:28-65 */
int rn_planes_to_voxels(rn_ctx *ctx, int32_t n, const int32_t *rvi, const int32_t *rvc,
                        const float *ray_start, const float *ray_end, const float *S,
                        float *S_new, void *stream);

/* K3 batch_belief_propagation, launcher mrf_cuda.py:37-79.
 * msgs_in may be NULL: all-zero messages (then msgs_out is only written). */
int rn_bp_sweep(rn_ctx *ctx, int32_t n, const float *S, const int32_t *rvi, const int32_t *rvc,
                const float *acc_in, const float *msgs_in, float *acc_out, float *msgs_out,
                void *stream);

/* K4 batch_depth_estimation, launcher mrf_cuda.py:81-122 */
int rn_depth_estimation(rn_ctx *ctx, int32_t n, const float *S, const int32_t *rvi,
                        const int32_t *rvc, const float *acc, const float *msgs, float *S_new,
                        void *stream);

/* ---- fused kernels of the forward-pass drivers -------------------------- */

/* K9 batch_multi_view_cnn_forward_pass, launcher similarities.py:101-130:
 * sample + plane sweep -> S [n][D] */
int rn_mvcnn_similarities(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                          const float *P, const float *P_inv, const float *camera_center,
                          float *S, void *stream);

/* K10 batch_multi_view_cnn_forward_pass_with_depth, launcher similarities.py:252-285:
 * K9 + arg-max plane -> depth_map [n]; points [n][D][4] is filled too */
int rn_mvcnn_depth(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                   const float *P, const float *P_inv, const float *camera_center, float *S,
                   float *points, float *depth_map, void *stream);

/* K11 batch_mvcnn_planes_voxels_with_ray_marching, launcher
 * mvcnn_with_ray_marching_and_voxels_mapping.py:137-174 */
int rn_mvcnn_voxel_space(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                         const float *P, const float *P_inv, const float *camera_center,
                         int32_t *rvi, int32_t *rvc, float *S_voxel, void *stream);

/* K12 ..._with_depth, launcher mvcnn_with_ray_marching_and_voxels_mapping.py:339-378 */
int rn_mvcnn_voxel_space_depth(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs,
                               const float *features, const float *P, const float *P_inv,
                               const float *camera_center, int32_t *rvi, int32_t *rvc,
                               float *S_voxel, float *depth_map, void *stream);

/* K1 batch_raynet_fp, launcher raynet_fp.py:274-326 */
int rn_fused_bp_sweep(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                      const float *P, const float *P_inv, const float *camera_center,
                      int32_t *rvi, int32_t *rvc, float *S_voxel, const float *acc_in,
                      const float *msgs_in, float *acc_out, float *msgs_out, void *stream);

/* K2 batch_complete_depth_estimation, launcher raynet_fp.py:328-376 */
int rn_fused_depth(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs, const float *features,
                   const float *P, const float *P_inv, const float *camera_center,
                   int32_t *rvi, int32_t *rvc, float *S_voxel, const float *acc,
                   const float *msgs, float *depth_map, void *stream);

/* ---- resident-scene path (what RayNetForwardPass runs on MI355X) -------- *
 * The reference recomputes features, similarities, traversal and mapping in
 * each of its 3 BP sweeps and again in the depth sweep (forward_pass.py:593-664,
 * :682-736) and round-trips messages through host memory.  With 288 GB of HBM
 * the per-ray state stays resident instead: rn_scene_prepare runs the
 * K1-prefix once per reference image and keeps, per ray, the packed voxel list
 * and the clipped+renormalised voxel-space column; the sweeps then stream it.
 * Results are those of K1/K2 called with the same inputs.
 *   vox  [n][M] i32  packed (x<<20 | y<<10 | z)
 *   Sr   [n][M] f32  clip_and_renorm(S_voxel) (mrf_bp.cu:103-111); the row of a ray with
 *                    rvc <= 1 is NOT written (such a ray sends no message and its depth is
 *                    that of its only voxel, mrf_np.py:300: nothing reads its column)
 * features_views: N device pointers (HOST array), one [Hf][Wf][F] map per view,
 * so a bank of per-view feature maps needs no re-stacking per reference image.
 * order (optional, may be NULL): a permutation of 0..n-1; wavefront i of the plane
 * sweep works on ray order[i].  It changes nothing but the schedule: walking the
 * rays along the direction of the epipolar lines keeps the neighbour views'
 * feature rows in L2.
 * ray_segments (optional scratch, may be NULL): [n][8] f32; the traversal leaves every
 * ray's bbox entry / exit point there and the plane sweep reads it back instead of
 * repeating the double-precision back-projection (sampling_schemes.cu:44-90) per wavefront. */
int rn_scene_prepare(rn_ctx *ctx, int32_t n, const int32_t *ray_idxs,
                     const float *const *features_views_host, const float *P,
                     const float *P_inv, const float *camera_center, const int32_t *order,
                     int32_t *vox, int32_t *rvc, float *Sr, float *ray_segments, void *stream);

/* The same for ALL reference images of a scene in two launches (one grid row per image).
 * Image g owns rows [g*rows_per_image, g*rows_per_image + n) of vox / rvc / Sr; all images
 * share ray_idxs [n] and the optional schedule `order` [n].
 *   cameras        [n_images][12N + 12 + 4] f32: P of the N views, P_inv and camera centre
 *                  of the reference view (device)
 *   features_views [n_images][N] device pointers, stored in DEVICE memory
 *   ray_segments   optional scratch [n_images * rows_per_image][8] f32 (see above) */
int rn_scene_prepare_all(rn_ctx *ctx, int32_t n_images, int32_t n, int64_t rows_per_image,
                         const int32_t *ray_idxs, const float *const *features_views,
                         const float *cameras, const int32_t *order, int32_t *vox, int32_t *rvc,
                         float *Sr, float *ray_segments, void *stream);

/* Slab boxes (optional).  The accumulator scatter of RN_ROWS_PATCHES rows sums a tile's
 * messages in an LDS image of the bounding box of the voxels the tile's rays visit in a chunk
 * of steps.  That box depends on the rays alone, not on the messages: bound to a list buffer,
 * rn_scene_prepare_all leaves, per 64 consecutive rows and per 16 steps, the box of the voxels
 * those rays visit there (a DDA is monotone along every axis: the extremes are the first and
 * the last voxel), and the three scatters of a pass merge boxes instead of scanning the lists.
 *   vox    the buffer rn_scene_prepare_all writes its lists to ([rows][M] i32)
 *   boxes  rn_slab_boxes_size(rows) i32 of scratch owned by the caller
 * Contract: while bound, the rows of `vox` are written by rn_scene_prepare_all only
 * (rn_scene_prepare into the buffer switches the table off until the next prepare_all);
 * rn_scene_bp_sweep* use the table for list pointers inside `vox` whose rows it covers, and
 * scan the lists as before for any other pointer.  boxes == NULL unbinds. */
int64_t rn_slab_boxes_size(const rn_ctx *ctx, int64_t rows);
int rn_scene_bind_slab_boxes(rn_ctx *ctx, const int32_t *vox, int64_t rows, int32_t *boxes);

/* Work list of the LDS-box scatter for the rows [0, rows) of the list buffer `vox` at tile level
 * `level` (0: 128 rays x 32 steps, 1: 256 x 16): items[i] = tile << 12 | first chunk << 6 | chunks
 * -- one workgroup per item, in the list's order (the caller sorts longest first and never lists
 * a chunk no ray of its tile reaches; every (tile, chunk) below the tile's longest ray must be in
 * exactly one item).  A scatter launch over exactly these rows at that level takes the list;
 * any other launch -- and every launch after items == NULL -- deals tiles x chunks out itself.
 * The list depends on the rays' voxel counts only (ray_tracing.pyx:64-199), not on messages: it
 * is built once per scene and shard.  Speed only: the sums are the same.  The tile field has 19
 * bits: rows beyond 2^19 tiles of the level are refused (RN_ERR_INVALID). */
int rn_scene_bind_scatter_items(rn_ctx *ctx, const int32_t *vox, int64_t rows, int32_t level,
                                const int32_t *items, int32_t count);

/* What the adaptive accumulator scatter of the resident path last saw (diagnostics): the tile
 * shape in use (0: 128 rays x 32 steps, 1: 256 x 16, 2: slab scatter) and, of the launches
 * between the launcher's last two looks at the counters that have arrived on the host (they are
 * copied out for the first dozen scatters after a reset only), the number of tile chunks and how
 * many of them did not fit the LDS box. */
int rn_scatter_state(const rn_ctx *ctx, int32_t *level, uint32_t *chunks, uint32_t *overflowed);

/* 1 once the scatter no longer copies its overflow counters out (a dozen launches after
 * rn_create / rn_scatter_reset / rn_set_options): from then on no launch of the resident path
 * touches the host and the tile shape stays as it is -- the precondition for recording a step's
 * launches into a HIP graph (the reference has no counterpart: PyCUDA launches one kernel at a
 * time from the interpreter, cuda_implementations/raynet_fp.py:303-326). */
int rn_scatter_settled(const rn_ctx *ctx);

/* Voxel counts only (the traversal of rn_scene_prepare_all without its lists): rvc
 * [n_images][n] i32, row g*n + i = number of voxels ray ray_idxs[i] crosses in reference image
 * g (<= M).  The multi-GPU driver balances its ray shards by these counts (the cost of the BP
 * sweep, the scatter and the depth sweep is per traversed voxel, ray_tracing.pyx:64-199 decides
 * how many there are); cameras as in rn_scene_prepare_all. */
int rn_scene_count_voxels(rn_ctx *ctx, int32_t n_images, int32_t n, const int32_t *ray_idxs,
                          const float *cameras, int32_t *rvc, void *stream);

/* Accumulators of the resident path are stored as 4x4x4 bricks,
 * [ceil(gx/4)][ceil(gy/4)][ceil(gz/4)][4][4][4] f32 = rn_acc_size() floats (a ray stays
 * inside a brick for ~4 steps, so a wavefront's gather touches ~4x fewer cache lines than in
 * the [gx][gy][gz] array).  rn_acc_to_grid / rn_acc_from_grid convert to / from the
 * reference's [gx][gy][gz] layout (mrf_bp.cu:3-10); padding voxels are never read back.
 * acc_part: [rn_acc_copies()][rn_acc_size()] f32 partial accumulator(s) the sweep scatters
 * into, zero before the first sweep of an iteration (rn_acc_copies() is 1: per-XCD copies
 * were measured and bought nothing). */
int64_t rn_acc_size(const rn_ctx *ctx);
int rn_acc_to_grid(rn_ctx *ctx, const float *acc, float *grid_out, void *stream);
int rn_acc_from_grid(rn_ctx *ctx, const float *grid, float *acc_out, void *stream);
int rn_acc_copies(const rn_ctx *ctx);
/* first_sweep is a set of rn_sweep_flags:
 * RN_SWEEP_ZERO_MSGS   the messages are taken as zero and `msgs` is only written, so it needs
 *                      no zero-fill (the reference zero-fills for iteration 0,
 *                      forward_pass.py:613-615);
 * RN_SWEEP_UNIFORM_ACC every voxel of acc_in holds acc_in[0] (iteration 0 starts from the
 *                      prior everywhere, forward_pass.py:533-538): only that element is read,
 *                      nothing is gathered.
 * row_layout says how consecutive rows relate in the image, which only selects the
 * accumulator-scatter kernel (any value is correct for any input, it is a speed hint):
 * RN_ROWS_LINEAR  -- consecutive rows run along an image column / row (ray-index order);
 * RN_ROWS_PATCHES -- every 256 consecutive rows are a compact pixel patch (e.g. 16x16). */
typedef enum { RN_ROWS_LINEAR = 0, RN_ROWS_PATCHES = 1 } rn_row_layout;
typedef enum { RN_SWEEP_ZERO_MSGS = 1, RN_SWEEP_UNIFORM_ACC = 2 } rn_sweep_flags;
/* The scatter for RN_ROWS_PATCHES adapts its tile shape to the scene from what previous
 * sweeps measured (LDS overflow counts); call this when the scene / cameras change so that
 * the next sweeps start from the default again. */
int rn_scatter_reset(rn_ctx *ctx);
int rn_scene_bp_sweep(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *vox,
                      const int32_t *rvc, const float *acc_in, float *msgs, float *acc_part,
                      int32_t first_sweep, int32_t row_layout, void *stream);
/* Deterministic mode: the same sweep, but every message is turned into a
 * signed 31.32 fixed-point integer and summed with 64-bit integer additions in LDS and in
 * acc_part_fixed [rn_acc_size()] i64 (zero before the first sweep of an iteration).  Integer
 * addition is associative: the accumulator is bit-identical from run to run, and -- with the
 * partials of several GPUs summed as int64 -- for any number of ranks.
 * rn_acc_combine_fixed: acc_out = prior + acc_part_fixed * 2^-32, and zeroes the partial. */
int rn_scene_bp_sweep_fixed(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *vox,
                            const int32_t *rvc, const float *acc_in, float *msgs,
                            int64_t *acc_part_fixed, int32_t first_sweep, int32_t row_layout,
                            void *stream);
int rn_acc_combine_fixed(rn_ctx *ctx, int64_t *acc_part_fixed, float prior, float *acc_out,
                         void *stream);
/* The same on `count` elements of any slab of the accumulator: what a rank runs on ITS 1 / N of
 * the voxels between a reduce-scatter of the fixed-point partials and the all-gather of the
 * float accumulator (sharded combine; 3/4 of the all-reduce's bytes on the wire). */
int rn_acc_combine_fixed_range(rn_ctx *ctx, int64_t *acc_part_fixed, int64_t count, float prior,
                               float *acc_out, void *stream);
/* acc_out = prior + sum over copies (+ optionally `extra`, e.g. nothing or a
 * peer's partial); the copies are zeroed for the next iteration. */
int rn_acc_combine(rn_ctx *ctx, float *acc_part, float prior, float *acc_out, void *stream);
/* Only the local sum (no prior), written to acc_out; used before an all-reduce. */
int rn_acc_reduce_local(rn_ctx *ctx, float *acc_part, float *acc_out, void *stream);
int rn_acc_add_prior(rn_ctx *ctx, float *acc, float prior, void *stream);

/* depth_map [n] and, when S_new != NULL, the per-ray distribution [n][M].
 * rays_per_center > 0: the n rays are consecutive groups of rays_per_center rays (one
 * group per reference image) and group g measures its distances from
 * camera_center[4*g .. 4*g+3]; 0: one centre for all rays. */
int rn_scene_depth(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *vox,
                   const int32_t *rvc, const float *acc, const float *msgs,
                   const float *camera_center, int32_t rays_per_center, float *S_new,
                   float *depth_map, void *stream);

/* ---- one pass as a PLAN (what RayNetForwardPass.forward_pass enqueues per step) ----------
 * forward_pass.py:579-748 in the resident form: the caller describes the scene's buffers once
 * and then runs whole phases of a pass with one call each -- the K1 prefix of all images, one BP
 * iteration over all images, the depth sweep -- instead of one call per kernel.  Between the
 * SWEEP phases of two iterations the caller runs its exchange (the all-reduce across GPUs of
 * acc[iteration & 1], or of acc_fixed), nothing else.
 *
 * Accumulators.  Float mode: acc[0], acc[1] (rn_acc_size() floats each, bricked) hold the
 * SUM of the messages only; the prior is added where an accumulator is read (`prior + sum`,
 * the value rn_acc_combine would have stored -- bit for bit), so there is no combine kernel,
 * and the sweep of iteration t clears acc[t & 1] itself before its scatter adds into it: no
 * buffer has to be zeroed or refilled by the caller, ever (forward_pass.py:676-678's swap +
 * fill(prior) costs nothing here).  Iteration 0 reads no accumulator (the prior everywhere).
 * Deterministic mode: the scatter adds 31.32 fixed-point integers into acc_fixed (cleared by
 * RN_RUN_PREPARE and by every RN_RUN_COMBINE), RN_RUN_COMBINE turns it into acc[t & 1] =
 * prior + sum (a full log-odds accumulator), which iteration t + 1 reads as it is.
 * The depth sweep reads acc[(iterations - 1) & 1]; with iterations == 0 that is acc[1], which
 * the caller then sets itself (zeros in float mode, the prior in deterministic mode). */
typedef struct {
    int32_t n_images;              /* reference images of the pass                       */
    int32_t n;                     /* rays per image (this rank's rows of every image)   */
    int64_t rows_per_image;        /* image g owns rows [g*rows_per_image, +n); % 256 == 0 */
    const int32_t *ray_idxs;       /* [n], shared by the images                          */
    const int32_t *order;          /* optional schedule of the plane sweep, see rn_scene_prepare */
    const float *const *features_views;  /* DEVICE [n_images][N] feature-map pointers    */
    const float *cameras;          /* [n_images][12N + 16], see rn_scene_prepare_all     */
    int32_t *vox;                  /* [n_images*rows_per_image][M]                       */
    int32_t *rvc;                  /* [n_images*rows_per_image]                          */
    float *Sr;                     /* [n_images*rows_per_image][M]                       */
    float *msgs;                   /* [n_images*rows_per_image][M]                       */
    float *ray_segments;           /* optional scratch [n_images*rows_per_image][8]      */
    float *acc[2];                 /* see above                                          */
    int64_t *acc_fixed;            /* deterministic mode only, else NULL                 */
    float *depth;                  /* [n_images*rows_per_image]                          */
    float prior;                   /* log(gamma / (1 - gamma)), forward_pass.py:533-538  */
    int32_t row_layout;            /* rn_row_layout                                      */
    float *depth_image;            /* optional [n_images][depth_image_stride]: the depth sweeps
                                      write image g's map in RAY-INDEX (pixel) order,
                                      depth_image[g*stride + ray_idxs[row]], instead of depth[]
                                      in row order -- what forward_pass.py:744 hands out, with
                                      no reordering pass behind the sweep.  Entries no ray of the
                                      list maps to are left as they are                      */
    int64_t depth_image_stride;    /* floats between two images' maps (>= max ray index + 1) */
    int32_t sweep_xcd_chunk;       /* plane sweep: consecutive wavefronts (entries of `order`) in
                                      groups of this many RAYS per XCD, the groups dealt round the
                                      8 XCDs -- one group is what an XCD's private L2 sees side by
                                      side (a multiple of 4; 0: the library's default, 2048)    */
} rn_scene_plan;
typedef enum {
    RN_RUN_PREPARE = 1,   /* traversal + plane sweep + mapping of all images (rn_scene_prepare_all) */
    RN_RUN_SWEEP = 2,     /* BP iteration `iteration` over all images: messages + scatter          */
    RN_RUN_COMBINE = 4,   /* deterministic mode: acc[iteration & 1] = prior + acc_fixed            */
    RN_RUN_DEPTH = 8,     /* depth sweep of image `image` (all images in one launch if < 0) after
                             `iteration` BP iterations                                             */
    RN_RUN_DEPTH_RANGE = 16 /* depth sweep of `count` consecutive images from `first` in ONE launch,
                             `image` = first | count << 16: a single GPU decodes all images but
                             the last together (no launch tails between them) and the last one on
                             its own, under which the others' maps leave                         */
} rn_run_phase;
/* Runs the phases named in `phases` in the order PREPARE, SWEEP, COMBINE, DEPTH_RANGE / DEPTH.
 * (The first PREPARE | SWEEP of iteration 0 with a given prior synchronises `stream` once: the one
 * occupancy of that iteration is evaluated on the device and kept with the context.) */
int rn_scene_run(rn_ctx *ctx, const rn_scene_plan *plan, int32_t phases, int32_t iteration,
                 int32_t image, void *stream);

/* out[i] = rows[index[i]] for i < n: the depth rows of a pass (one rank's, or the ranks' blocks
 * side by side after the all-gather) into the ray-index order forward_pass.py:744 hands out.
 * `out` (16-byte aligned, like `index`) may be PAGE-LOCKED HOST memory (hipHostMalloc, torch's
 * pin_memory): the kernel then writes the map across PCIe itself -- the reordering pass and the
 * device-to-host copy (`.get()`, forward_pass.py:739-744) are one launch on `stream`. */
int rn_stitch_rows(rn_ctx *ctx, int64_t n, const float *rows, const int32_t *index, float *out,
                   void *stream);

/* ---- measurement -------------------------------------------------------- */
/* Per-launch hipEvent timing on the stream each kernel runs on.  Between
 * rn_prof_begin and rn_prof_end every kernel launch made through this context
 * is bracketed by two events; rn_prof_end synchronises and returns, per launch,
 * the kernel family (rn_kernel_id), its ray count and its duration. */
typedef enum {
    RN_K_TRAVERSE = 1,   /* voxel traversal (thread per ray)                 */
    RN_K_SWEEP_MAP = 2,  /* plane sweep + softmax (+ planes->voxels mapping) */
    RN_K_BP = 3,         /* BP sweep                                         */
    RN_K_DEPTH = 4,      /* depth estimation / arg-max                       */
    RN_K_ACC = 5,        /* accumulator combine / fill                       */
    RN_K_OTHER = 6,
    RN_K_SCATTER = 7     /* accumulator scatter of the BP messages (tile-transposed) */
} rn_kernel_id;
int rn_prof_begin(rn_ctx *ctx, int32_t capacity);
/* Which families the next rn_prof_begin brackets: bit (1 << rn_kernel_id) each, default all.
 * An event pair costs the stream a few microseconds; a timed region that wants one kernel's
 * durations only (bench.py: the dominant one) selects that family. */
int rn_prof_select(rn_ctx *ctx, uint32_t kernel_mask);
int rn_prof_end(rn_ctx *ctx, int32_t *count, int32_t *kernel_ids_host, int32_t *n_rays_host,
                float *ms_host);
/* after rn_prof_end: start of every recorded launch, in ms after the first one's start
 * (the gaps between launches = what the host side costs; tools/timeline.py) */
int rn_prof_offsets(rn_ctx *ctx, float *start_ms_host);

/* Self-test of the exact arithmetic shortcut of the index maps (raynet_kernels.h:
 * round_half_away), for tests/: out is [2][n] -- roundf(a), round_half_away(a). */
int rn_selftest_arith(rn_ctx *ctx, int32_t n, const float *a, float *out, void *stream);
/* The same for round_quotient_fast, which stands in for the two divisions of a projection
 * (feature_similarities.cu:24-25) wherever it is sure of the rounded result: out is [3][n] --
 * round_half_away(x / d), the shortcut's value, 1.0 where it is sure (elsewhere the kernels
 * take the division). */
int rn_selftest_quotient(rn_ctx *ctx, int32_t n, const float *x, const float *d, float *out,
                         void *stream);

/* The plane sweep's index arithmetic alone (feature_similarities.cu:10-61, 84-98): for every
 * ray, view and depth plane the feature vector the sweep would gather, as fy * (W + padding + 1)
 * + fx -- out [n][N][D][2]: by the generic sweep's expressions (the reference's, operation for
 * operation) and by the cooperative sweep's (rounded quotients through the reciprocal where
 * provably the same).  tests/ compares both with the reference's own NumPy `project`. */
int rn_selftest_feature_offsets(rn_ctx *ctx, int32_t n, const float *P, const float *ray_start,
                                const float *ray_end, int32_t *out, void *stream);

/* The same for the two shortcuts of the planes -> voxels mapping of the resident path
 * (raynet_kernels.h: markstein_div, plane_index_from_table), which stand in for the division
 * by |ray|^2 and for the plane walk of planes_voxels_mapping.cu:48-67.  out is [5][n]:
 * a / b (IEEE), Markstein's quotient from RN(1 / b), 1.0 where the kernels use it (b within
 * 2^-60 .. 2^60 and |a| < 2^60; the IEEE division elsewhere), the walk's plane index for clamp(t, 1e-4,
 * 1 - 1e-4) with the context's D, the table look-up's. */
int rn_selftest_mapping(rn_ctx *ctx, int32_t n, const float *a, const float *b, const float *t,
                        float *out, void *stream);

/* ---- differentiable MRF block (training) --------------
 * The reference builds this block from TensorFlow ops and lets autodiff
 * differentiate it (raynet/tf_implementations/forward_backward_pass.py:194-246,
 * raynet/mrf/mrf_tf.py:1-236); these entry points are the forward on a column that the
 * framework has already clipped + renormalised (so that step stays differentiable there)
 * and the analytic reverse-mode derivative of one BP sweep / of the depth distribution.
 * rvi is the [n][M][3] traversal output (K5 layout). */

/* left plane index and interpolation weights (c1, c2), [n][M] each, of the
 * planes->voxels mapping (planes_voxels_mapping.cu:48-84): S_voxel[i] is
 * normalise_i(c1[i] S[left[i]] + c2[i] S[left[i]+1]) */
int rn_plane_weights(rn_ctx *ctx, int32_t n, const int32_t *rvi, const int32_t *rvc,
                     const float *ray_start, const float *ray_end, int32_t *left, float *c1,
                     float *c2, void *stream);

/* rn_bp_sweep / rn_depth_estimation without their internal clip_and_renorm */
int rn_train_bp_sweep(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                      const int32_t *rvc, const float *acc_in, const float *msgs_in,
                      float *acc_out, float *msgs_out, void *stream);
int rn_train_depth(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                   const int32_t *rvc, const float *acc, const float *msgs, float *S_new,
                   void *stream);

/* Backward of rn_train_bp_sweep.  g_msgs_out [n][M]: gradient w.r.t. the new messages
 * from their direct use by the next sweep; g_acc_out [G] (may be NULL): gradient w.r.t.
 * acc_out.  Results: g_Sr [n][M] += , g_acc_in [G] += (atomic; caller zeroes),
 * g_msgs_in [n][M] = . */
int rn_train_bp_sweep_bwd(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                          const int32_t *rvc, const float *acc_in, const float *msgs_in,
                          const float *g_msgs_out, const float *g_acc_out, float *g_Sr,
                          float *g_acc_in, float *g_msgs_in, void *stream);

/* Backward of rn_train_depth for g_S_new [n][M]; same output conventions. */
int rn_train_depth_bwd(rn_ctx *ctx, int32_t n, const float *Sr, const int32_t *rvi,
                       const int32_t *rvc, const float *acc, const float *msgs,
                       const float *g_S_new, float *g_Sr, float *g_acc, float *g_msgs,
                       void *stream);

/* ---- consumers of the depth maps ----------------------
 * Pixel i = u*H + v (column-major, common/image.py:252-255), depth maps are [H][W] f32 as
 * the forward pass writes them (scripts/forward_pass.py:136-142); matrices and points are
 * float64 like the NumPy code these entry points restate. */

/* raynet/pointcloud.py:121-147 (_generate_points_per_image without the pixel selection):
 * points [3][H*W] f64 = centre + depth * normalised ray direction, for every pixel.
 * P_pinv [4][3], camera_center [4] (device). */
int rn_depthmap_points(rn_ctx *ctx, int32_t H, int32_t W, const double *P_pinv,
                       const double *camera_center, const float *depth_map, double *points,
                       void *stream);

/* One neighbour view of raynet/pointcloud.py:205-245: tau[i] = max(tau[i], |depth_map at the
 * point's projection - distance of the point to that camera|), inf where the projection
 * falls outside the view; first != 0 starts tau.  points [3][n] f64, P [3][4]. */
int rn_consistency_tau(rn_ctx *ctx, int32_t n, int32_t H, int32_t W, int32_t first,
                       const double *points, const double *P, const double *camera_center,
                       const float *depth_map, double *tau, void *stream);

/* Exact nearest neighbours (the KDTree.query of raynet/pointcloud.py:63-72, behind
 * Accuracy / Completeness, metrics.py:155-236): for every query point the distance to and
 * the index of the closest reference point (either output may be NULL).  Points are
 * [n][4] f32 (x, y, z, unused). */
int rn_nearest_neighbors(rn_ctx *ctx, int32_t n_ref, const float *ref_xyzw, int32_t n_query,
                         const float *query_xyzw, float *dist, int32_t *idx, void *stream);

/* hipEvent pair on `stream`; rn_timer_stop returns elapsed milliseconds after
 * synchronising on the stop event (bench.py's per-kernel timing). */
int rn_timer_start(rn_ctx *ctx, void *stream);
int rn_timer_stop(rn_ctx *ctx, void *stream, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* RAYNET_HIP_H */
