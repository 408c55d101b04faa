// raynet_prepare.inl -- the per-ray prefix of the path: ray sampling (K8), voxel traversal
// (K5) and the plane sweep + planes->voxels mapping (K6 / K7 / K9-K12, the K1 prefix).
// Included by raynet_hip.hip inside its anonymous namespace.

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

// occupancy_to_ray(prior, 0) by the device's own arithmetic (see first_sweep_messages)
__global__ void k_first_occupancy(float prior, float *out) { out[0] = occupancy_to_ray(prior, 0.0f); }

// ------------------------------------------------------------ K5 traversal
// One thread per ray (the DDA is a chain of sequential fp32 additions, bit-exactness
// forbids re-associating it).  Source of the segment: explicit starts/ends, or the
// camera (sample_in_bbox), as in the fused kernels.
// A thread writing its own row step by step produces one partial-line write per voxel
// (4x write amplification measured).  Each 64-thread block therefore collects
// [64 rays][TRAV_TILE steps] in LDS and writes finished tiles as coalesced row segments
// (16 steps: 0.42 ms per scene against 0.46 at 32 and 0.70 at 64 -- the tile's LDS decides how
// many of these serial, latency-bound threads a CU holds).
#define RN_TRAV_TILE 16
constexpr int TRAV_TILE = RN_TRAV_TILE;     // steps collected per flush (the packed list's slab boxes are per 16-step tile: k_traverse<true> needs 16; 32 / 64 compile for the reference layout only)
template <bool PACKED>
__global__ __launch_bounds__(WAVE) void k_traverse(Params p, int n,
                                                   const int32_t *__restrict__ ray_idxs,
                                                   const float *__restrict__ P_inv,
                                                   const float *__restrict__ cc,
                                                   const float *__restrict__ starts,
                                                   const float *__restrict__ ends, int32_t *vox,
                                                   int32_t *rvc, int cam_stride,
                                                   int64_t rows_per_image, float *seg_out,
                                                   int2 *slab_boxes = nullptr) {
    static_assert(TRAV_TILE == SLAB_BOX_STEPS || !PACKED, "slab boxes are per flushed tile");
    __shared__ int32_t tile[WAVE * (TRAV_TILE + 1)];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * WAVE;
    if (rows_per_image > 0) {       // blockIdx.y = reference image of a scene-wide launch
        const int g = blockIdx.y;
        P_inv += (size_t)g * cam_stride;
        cc += (size_t)g * cam_stride;
        if (vox) vox += (size_t)g * rows_per_image * p.M * (PACKED ? 1 : 3);
        rvc += (size_t)g * rows_per_image;
        if (seg_out) seg_out += (size_t)g * rows_per_image * 8;
        if (slab_boxes) slab_boxes += (size_t)g * (rows_per_image / WAVE) * slab_box_count(p.M);
    }
    if (slab_boxes) slab_boxes += (size_t)blockIdx.x * slab_box_count(p.M);
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
    // The walk on the PACKED voxel word: a step adds +-1 to one 10-bit field (a field that
    // leaves [0, g) may borrow from / carry into its neighbour -- the ray is switched off in
    // that very step and the word never emitted), the end test is one comparison of words,
    // and leaving the grid is the sign of a per-axis count of the steps that are left
    // (decremented by the step's own mask: subtract-with-borrow, no select).  Same choices,
    // same additions on tx / ty / tz, same list: 29 -> 20 vector instructions per step.
    int pk = pack_voxel(cx, cy, cz);
    const bool last_in = (unsigned)last[0] < (unsigned)g[0] && (unsigned)last[1] < (unsigned)g[1] &&
                         (unsigned)last[2] < (unsigned)g[2];
    const int pk_last = last_in ? pack_voxel(last[0], last[1], last[2]) : -1;   // -1: no voxel's word
    const int ux = step[0] * (1 << 20), uy = step[1] * (1 << 10), uz = step[2];
    int room_x = step[0] > 0 ? g[0] - 1 - cx : cx, room_y = step[1] > 0 ? g[1] - 1 - cy : cy,
        room_z = step[2] > 0 ? g[2] - 1 - cz : cz;
    // `active` = this ray still has a voxel (cx,cy,cz) to emit at index `count`
    // one step of an active ray: emit, then advance (ray_tracing.pyx:166-197).  Everything is
    // computed unconditionally and the ray switched off at the end -- a ray that has just
    // emitted its last voxel moves once more into a word nobody reads.  `count >= M` is the
    // loops' own bound (every active ray has emitted base + k + 1 voxels); the count itself
    // is read off the counters once per tile.
#define RN_DDA_STEP(k)                                                             \
    if (active) {                                                                  \
        if (vox) tile[lane * (TRAV_TILE + 1) + (k)] = pk;                          \
        const bool at_last = pk == pk_last;                                        \
        const bool a_ = tx < ty, b_ = tx < tz, c_ = ty < tz;                       \
        const bool mx = a_ & b_, my = !a_ & c_, mxy = mx | my; /* mz = !mxy */     \
        const int uyz = my ? uy : uz;                                              \
        pk += mx ? ux : uyz;                                                       \
        tx += mx ? td0 : 0.0f;                                                     \
        ty += my ? td1 : 0.0f;                                                     \
        tz += mxy ? 0.0f : td2;                                                    \
        room_x -= (int)mx;                                                         \
        room_y -= (int)my;                                                         \
        room_z = room_z - 1 + (int)mxy;                                            \
        active = !at_last && (room_x | room_y | room_z) >= 0;                      \
    }
    const float td0 = td[0], td1 = td[1], td2 = td[2];
    // every emitted voxel is followed by one move (also the last one, and the move that leaves
    // the grid): voxels emitted = moves made = what the three counters have lost
    const int room_sum0 = room_x + room_y + room_z;
    const bool whole_tiles = p.M % TRAV_TILE == 0;
    for (int base = 0; base < p.M; base += TRAV_TILE) {
        if (__ballot(active) == 0) break;
        if (whole_tiles) {
#pragma unroll
            for (int k = 0; k < TRAV_TILE; k++) { RN_DDA_STEP(k) }
        } else {
            for (int k = 0; k < TRAV_TILE && base + k < p.M; k++) { RN_DDA_STEP(k) }
        }
        count = room_sum0 - (room_x + room_y + room_z);
        if (!vox) continue;          // count-only launch (rn_scene_count_voxels): nothing to flush
        wave_sync();
        // Bounding box of the voxels these 64 rays emitted in this slab of steps, for the
        // accumulator scatter (k_scatter_box), which otherwise rebuilds it from the lists in
        // every BP iteration.  A DDA moves monotonically along every axis, so a ray's extreme
        // coordinates within the slab are those of its first and its last voxel there.
        if (PACKED && slab_boxes) {
            const int emitted = min(max(count - base, 0), TRAV_TILE);
            int lo0 = 1 << 30, lo1 = 1 << 30, lo2 = 1 << 30, hi0 = -1, hi1 = -1, hi2 = -1;
            if (emitted > 0) {
                const int a = tile[lane * (TRAV_TILE + 1)];
                const int b = tile[lane * (TRAV_TILE + 1) + emitted - 1];
                const int ax = a >> 20, ay = (a >> 10) & 1023, az = a & 1023;
                const int bx = b >> 20, by = (b >> 10) & 1023, bz = b & 1023;
                lo0 = min(ax, bx); hi0 = max(ax, bx);
                lo1 = min(ay, by); hi1 = max(ay, by);
                lo2 = min(az, bz); hi2 = max(az, bz);
            }
            lo0 = wave_min_i(lo0); lo1 = wave_min_i(lo1); lo2 = wave_min_i(lo2);
            hi0 = wave_max_i(hi0); hi1 = wave_max_i(hi1); hi2 = wave_max_i(hi2);
            if (lane == 0)      // an empty slab (hi < lo) never reaches the scatter's merge
                slab_boxes[base / TRAV_TILE] = make_int2(hi0 < 0 ? 0x7fffffff : pack_voxel(lo0, lo1, lo2),
                                                         hi0 < 0 ? -1 : pack_voxel(hi0, hi1, hi2));
        }
        // flush: TRAV_TILE consecutive steps of one ray are one contiguous segment
        if (PACKED && TRAV_TILE == 16 && (p.M & 3) == 0) {
            // four steps per lane and store (16-byte aligned: M and the tile base are multiples
            // of 4): 4 store rounds per tile instead of 16; a row's last, partial group of four
            // is written entry by entry -- nothing beyond a ray's count is touched
#pragma unroll
            for (int j = 0; j < WAVE; j += WAVE / 4) {
                const int row = j + (lane >> 2), q4 = (lane & 3) * 4;
                const int nv = min(__shfl(count, row) - base - q4, 4);
                if (r0 + row < n && nv > 0) {
                    const int32_t *t = tile + row * (TRAV_TILE + 1) + q4;
                    int32_t *dst = vox + (size_t)(r0 + row) * p.M + base + q4;
                    if (nv == 4) {
                        *reinterpret_cast<int4 *>(dst) = make_int4(t[0], t[1], t[2], t[3]);
                    } else {
                        dst[0] = t[0];
                        if (nv > 1) dst[1] = t[1];
                        if (nv > 2) dst[2] = t[2];
                    }
                }
            }
            wave_sync();
            continue;
        }
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
//          3: as 2, then BP iteration 0 of the ray (first_sweep_messages) -> msgs_out
// Dynamic LDS: [axes gx+gy+gz][plane positions][per wave: D plane column, M values (3 M for 3)]
// waves per SIMD the register allocation has to leave room for.  Round 1: 7 (<= 72 VGPRs; the
// kernel needs 74 unconstrained) -1.8 % on the 5-view sweep, 8 (64 VGPRs) -0.5 %.  Round 2
// (profiles/r02_exp_knobs.txt, after the ray index moved to an SGPR and rays without voxels
// stopped sweeping): 6 is best -- 2.80 ms against 2.91 at 7, 2.81 at 5, 2.84 at 4, 3.02 at 8.
// The wide sweeps (two load rounds of 7+ views do not fit), the reference-layout variants
// and the 4-view sweep (which would spill) are left alone
// cache policy of the list's LDS-DMA loads: 2 = non-temporal (read once here; k_sweep_map 2.957 -> 2.904 ms
// with it: the feature gathers keep the L2), 0 = default
#define RN_SWEEP_LIST_CPOL 2
#define RN_SWEEP_MIN_WAVES 6
// BP iteration 0 of one ray straight from its clipped + renormalised column, which the plane
// sweep still holds in LDS when it stores it (mrf_bp.cu:88-177 with the prior in every voxel and
// no messages yet: ONE occupancy for the whole ray, nothing to gather, nothing to read) -- the
// first k_bp launch of a pass, which read the column and the voxel list back from HBM to do
// exactly this, goes away.  Operation for operation bp_ray's const-occupancy path (same scans,
// same carries, same order): the messages are the same bits, whichever kernel writes them.
// col[i] = clipped value (renormalised by `inv_sum` here, stored to Sr_row as k_bp would read
// it); ts_row / cex_row: two more LDS rows of M floats of this wavefront.
__device__ __forceinline__ void first_sweep_messages(int count, int lane, float *col, float inv_sum,
                                                     float *ts_row, float *cex_row, float o_const,
                                                     float *Sr_row, float *msg_row) {
    // o_const = occupancy_to_ray(prior, 0): the same for every ray of every pass with this
    // prior -- evaluated ONCE on the device (k_first_occupancy, the bits k_bp's own evaluation
    // gives) and handed in, instead of ~30 instructions per wavefront
    float carryT = 1.0f, carryC = 0.0f;
    // The transmittance scan of a chunk multiplies the SAME constant 1 - o in every valid lane,
    // and a lane's inclusive product depends on the lanes below it only: for every valid lane
    // it is the scan of a full chunk, whichever chunk -- scanned once per wavefront, the very
    // values (and the very carry, lane 63 of a full chunk) the per-chunk scans produce.
    const float incl_full = wave_scan_mul(1.0f - o_const);
    const float t_shift = wave_shift1(incl_full, 1.0f), t_carry = lane63(incl_full);
    for (int base = 0; base < count; base += WAVE) {
        const int i = base + lane;
        const bool valid = i < count;
        float sv = 0.0f;
        if (valid) {
            sv = col[i] * inv_sum;
            // streamed out, read again only by later kernels: keep it out of the L2 the feature
            // gathers live in
            __builtin_nontemporal_store(sv, Sr_row + i);
        }
        const float o = valid ? o_const : 0.0f;
        const float T = carryT * t_shift;
        carryT = carryT * t_carry;
        const float ts = T * sv;
        const float w = valid ? o * ts : 0.0f;
        const float inclC = wave_scan_add(w);
        const float cex = carryC + wave_shift1(inclC, 0.0f);
        carryC = carryC + lane63(inclC);
        if (valid) {
            ts_row[i] = ts;
            cex_row[i] = cex;
            col[i] = w;
        }
    }
    float carryS = 0.0f;
    for (int base = ((count - 1) / WAVE) * WAVE; base >= 0; base -= WAVE) {
        const int i = base + lane;
        float tot;
        const float suf = carryS + wave_suffix_excl(i < count ? col[i] : 0.0f, lane, tot);
        carryS = carryS + tot;
        if (i < count) col[i] = suf;
    }
    for (int i = lane; i < count; i += WAVE) {
        const float cex = cex_row[i];
        const float pos = cex + ts_row[i];
        const float neg = cex + bp_div(col[i], 1.0f - o_const);
        __builtin_nontemporal_store(bp_log_ratio(pos, neg), msg_row + i);    // (as k_bp's rows)
    }
}

// MAPMODE 0's tail for one ray: the plane column out, and K10's points / first arg-max plane /
// distance to the camera (similarities.py:199-227)
__device__ __forceinline__ void planes_out(const Params &p, int r, const float s[3], const float e[3],
                                           const float *Sl, int lane, const float *__restrict__ cc,
                                           float *S_planes, float *depth_from_planes, float *points) {
    for (int k = lane; k < p.D; k += WAVE) S_planes[(size_t)r * p.D + k] = Sl[k];
    if (!depth_from_planes) return;
    float best = -INFINITY;
    int best_k = 0;
    for (int k = lane; k < p.D; k += WAVE) {
        float pt[3];
        plane_point(s, e, k, p.D, pt);
        reinterpret_cast<float4 *>(points)[(size_t)r * p.D + k] = make_float4(pt[0], pt[1], pt[2], 1.0f);
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

// ... and for the RPW rays of a wavefront of k_sweep_map_packed at once: lane l = plane l % DPAD of
// ray l / DPAD (its row in `r`, its segment in s / e), the arg-max folded within the ray's lanes.
// The comparisons are planes_out's (larger value, then smaller index): the same plane wins.
template <int RPW>
__device__ __forceinline__ void planes_out_packed(const Params &p, int r, const float s[3],
                                                  const float e[3], const float *Sl, int lane,
                                                  int nrays, const float *__restrict__ cc,
                                                  float *S_planes, float *depth_from_planes,
                                                  float *points) {
    constexpr int DPAD = WAVE / RPW;
    const int q = lane / DPAD, k = lane % DPAD;
    const bool mine = k < p.D && q < nrays;
    float best = -INFINITY;
    int best_k = 0;
    if (mine) {
        const float v = Sl[q * p.D + k];
        S_planes[(size_t)r * p.D + k] = v;
        if (depth_from_planes) {
            float pt[3];
            plane_point(s, e, k, p.D, pt);
            reinterpret_cast<float4 *>(points)[(size_t)r * p.D + k] = make_float4(pt[0], pt[1], pt[2], 1.0f);
            if (v > best) {
                best = v;
                best_k = k;
            }
        }
    }
    if (!depth_from_planes) return;
#pragma unroll
    for (int o = DPAD / 2; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int ok = __shfl_xor(best_k, o);
        if (ob > best || (ob == best && ok < best_k)) {
            best = ob;
            best_k = ok;
        }
    }
    if (k == 0 && q < nrays) {
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

// MAPMODE 1 / 2 / 3's tail for one ray: planes -> voxels from the softmaxed column Sl, then the
// normalised column (1), clip + renormalise (2), and BP iteration 0's messages (3)
template <int MAPMODE, bool PACKED>
__device__ __forceinline__ void voxels_out(const Params &p, int r, int count, int n_staged,
                                           const float s[3], const float e[3], const float *Sl,
                                           float *vals, const float *axes, const float *pos,
                                           const int32_t *__restrict__ vrow, float *S_voxel,
                                           float *msgs_out, float o_first, int lane) {
    constexpr bool RESIDENT = MAPMODE >= 2;
    float *out = S_voxel + (size_t)r * p.M;
    if (PACKED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of the voxel row
    // MAPMODE 2 = the resident path: value-only divisions through the hardware reciprocal
    // (the plane index walk inside stays IEEE); MAPMODE 1 = K6 / K11, reference arithmetic
    constexpr bool MAP_TABLE = RESIDENT;
    const float srsum =
        map_planes_to_voxels<PACKED, RESIDENT, PACKED, MAP_TABLE>(p, axes, vrow, count, s, e, Sl,
                                                                  vals, lane, n_staged, pos);
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
        if (MAPMODE == 3) {
            first_sweep_messages(count, lane, vals, sum, vals + p.M, vals + 2 * p.M, o_first, out,
                                 msgs_out + (size_t)r * p.M);
        } else {
            // streamed out, read again only by later kernels: keep it out of the L2 the feature
            // gathers live in
            for (int i = lane; i < count; i += WAVE)
                __builtin_nontemporal_store(vals[i] * sum, out + i);
        }
    }
}

// the ray's packed voxel ids straight from global memory into an LDS row (LDS-DMA: no staging
// registers; LDS address = wave-uniform base + 4 * lane, which is the row's own layout)
__device__ __forceinline__ int stage_voxel_row(const Params &p, const int32_t *__restrict__ vrow,
                                               int count, float *vals, int lane) {
    typedef const __attribute__((address_space(1))) void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    for (int c = 0; c < count; c += WAVE)
        if (c + lane < p.M)
            __builtin_amdgcn_global_load_lds((gptr)(vrow + c + lane), (lptr)(vals + c), 4, 0, RN_SWEEP_LIST_CPOL);
    return (count + WAVE - 1) & ~(WAVE - 1);
}

template <int SIM, int NV, int LPS, int MAPMODE, bool PACKED>
__global__ __launch_bounds__(SWEEP_BLOCK, (SIM == 2 && MAPMODE >= 2 && NV >= 5 && NV <= SWEEP_UNROLL2_MAX_VIEWS ? RN_SWEEP_MIN_WAVES : 1))
void k_sweep_map(
    Params p, int n, const int32_t *__restrict__ ray_idxs, FeatureViews fv,
    const float *__restrict__ P, const float *__restrict__ P_inv, const float *__restrict__ cc,
    const float *__restrict__ starts, const float *__restrict__ ends,
    const float *__restrict__ S_in, const float *__restrict__ axes_g,
    const int32_t *__restrict__ vox, const int32_t *__restrict__ rvc, float *S_planes,
    float *S_voxel, float *depth_from_planes, float *points,
    const int32_t *__restrict__ order, const float *const *__restrict__ fv_table, int cam_stride,
    int64_t rows_per_image, const float *__restrict__ seg, float *msgs_out, float o_first,
    float4 *zero_buf, int zero_count4, int xcd_chunk) {
    constexpr bool RESIDENT = MAPMODE >= 2;      // value-only divisions through the reciprocal
    // MAPMODE 3 stands in for the first k_bp launch of a pass, including what that launch clears
    // on the side: the partial accumulator the first scatter adds into (see k_bp)
    if (MAPMODE == 3 && zero_buf && blockIdx.y == 0) {
        const int nw = (n + SWEEP_WAVES - 1) / SWEEP_WAVES * SWEEP_WAVES;
        const int w = blockIdx.x * SWEEP_WAVES + (int)(threadIdx.x >> 6);
        for (int i = w * WAVE + (int)(threadIdx.x & (WAVE - 1)); i < zero_count4; i += nw * WAVE)
            zero_buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
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
        if (MAPMODE == 3) msgs_out += (size_t)g * rows_per_image * p.M;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int naxes = p.gx + p.gy + p.gz;
    float *axes = smem;
    const int wid = threadIdx.x >> 6;
    // [axes][plane positions 0 .. D (resident path)][per wave: D plane column, M values]
    float *pos = smem + ((naxes + 3) & ~3);
    float *Sl = pos + ((p.D + 4) & ~3) + wid * (p.D + (MAPMODE == 3 ? 3 : 1) * p.M);
    float *vals = Sl + p.D;
    if (MAPMODE != 0) {
        for (int i = threadIdx.x; i < naxes; i += SWEEP_BLOCK) axes[i] = axes_g[i];
        // the very expression the walk evaluates (planes_voxels_mapping.cu:60-67), once per plane
        if (RESIDENT)
            for (int i = threadIdx.x; i <= p.D; i += SWEEP_BLOCK) pos[i] = 0.0f + i * p.plane_step;
        __syncthreads();
    }
    int lane;
    int r = ray_of_wave<SWEEP_BLOCK, RN_XCD_CHUNK_SWEEP>(n, lane, xcd_chunk);
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

    // The ray's voxel row is needed only after the sweep -- three dependent ~1.5k-cycle loads
    // later if fetched there.  Its count is fetched with the segment and the packed ids go
    // straight into this wave's vals[] row (unused until the mapping): the sweep covers the
    // latency.
    int count = 0, n_staged = 0;
    const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
    if (MAPMODE != 0) {
        count = min(uniform(rvc[r]), p.M);
        // resident path: the column of a ray with <= 1 voxels is never read (such rays send no
        // message, mrf_np.py:300, and their depth is that of voxel 0): no sweep for the rays
        // that miss the box (5 % of config 2's)
        if (RESIDENT && count <= 1) return;
        if (PACKED) n_staged = stage_voxel_row(p, vrow, count, vals, lane);
    }
    if (SIM == 0) {
        for (int k = lane; k < p.D; k += WAVE) Sl[k] = S_in[(size_t)r * p.D + k];
    } else {
        if (SIM == 1)
            sweep_generic(p, fv, fv_table, P, s, e, lane, Sl);
        else
            sweep_coop<NV, LPS, RESIDENT>(p, fv, fv_table, P, s, e, lane, Sl);
        wave_sync();
        softmax_column<RESIDENT>(p.D, lane, Sl);
    }
    wave_sync();

    if (MAPMODE == 0)
        planes_out(p, r, s, e, Sl, lane, cc, S_planes, depth_from_planes, points);
    else
        voxels_out<MAPMODE, PACKED>(p, r, count, n_staged, s, e, Sl, vals, axes, pos, vrow, S_voxel,
                                    msgs_out, o_first, lane);
}

// The cooperative sweep for D <= 32 (the reference's own default, scripts/arguments.py:154) and
// D <= 16: RPW = 2 / 4 rays share a wavefront through projection, gathers, pair sums and softmax
// -- lane l = plane l % DPAD of ray l / DPAD, every one of the 8 load rounds full of live samples
// -- where one ray per wavefront would project 32 / 48 dead lanes and run 4 / 6 load rounds of
// duplicates; the per-ray tail (planes -> voxels, clip, BP iteration 0: lanes over VOXELS) then
// runs for the wavefront's rays one after the other.  Per (ray, plane) the arithmetic is the
// instruction sequence of k_sweep_map: same bits (tests/test_packed_sweep_gpu.py).
// Wavefront w takes rows RPW w .. RPW w + RPW - 1 of the launch.
// Dynamic LDS: [axes][plane positions][per wave: RPW x D plane columns, M values (3 M for 3)]
template <int NV, int LPS, int MAPMODE, bool PACKED, int RPW>
__global__ __launch_bounds__(SWEEP_BLOCK)
void k_sweep_map_packed(
    Params p, int n, const int32_t *__restrict__ ray_idxs, FeatureViews fv,
    const float *__restrict__ P, const float *__restrict__ P_inv, const float *__restrict__ cc,
    const float *__restrict__ starts, const float *__restrict__ ends,
    const float *__restrict__ axes_g, const int32_t *__restrict__ vox,
    const int32_t *__restrict__ rvc, float *S_planes, float *S_voxel, float *depth_from_planes,
    float *points, const int32_t *__restrict__ order, const float *const *__restrict__ fv_table,
    int cam_stride, int64_t rows_per_image, const float *__restrict__ seg, float *msgs_out,
    float o_first, float4 *zero_buf, int zero_count4, int xcd_chunk) {
    static_assert(RPW == 2 || RPW == 4, "two or four rays per wavefront");
    constexpr int DPAD = WAVE / RPW;
    constexpr bool RESIDENT = MAPMODE >= 2;
    const int nwaves = (n + RPW - 1) / RPW;      // wavefronts with a ray
    if (MAPMODE == 3 && zero_buf && blockIdx.y == 0) {     // (as k_sweep_map)
        const int nw = (nwaves + SWEEP_WAVES - 1) / SWEEP_WAVES * SWEEP_WAVES;
        const int w = blockIdx.x * SWEEP_WAVES + (int)(threadIdx.x >> 6);
        for (int i = w * WAVE + (int)(threadIdx.x & (WAVE - 1)); i < zero_count4; i += nw * WAVE)
            zero_buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
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
        if (MAPMODE == 3) msgs_out += (size_t)g * rows_per_image * p.M;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int naxes = p.gx + p.gy + p.gz;
    float *axes = smem;
    const int wid = threadIdx.x >> 6;
    float *pos = smem + ((naxes + 3) & ~3);
    float *Sl = pos + ((p.D + 4) & ~3) + wid * (RPW * p.D + (MAPMODE == 3 ? 3 : 1) * p.M);
    float *vals = Sl + RPW * p.D;
    if (MAPMODE != 0) {
        for (int i = threadIdx.x; i < naxes; i += SWEEP_BLOCK) axes[i] = axes_g[i];
        if (RESIDENT)
            for (int i = threadIdx.x; i <= p.D; i += SWEEP_BLOCK) pos[i] = 0.0f + i * p.plane_step;
        __syncthreads();
    }
    int lane;
    const int w = ray_of_wave<SWEEP_BLOCK, RN_XCD_CHUNK_SWEEP>(nwaves, lane, xcd_chunk);
    if (w < 0) return;
    const int r0 = w * RPW;
    const int nrays = min(RPW, n - r0);
    // this lane's ray (the lanes of a ray that does not exist shadow the wavefront's last one)
    int r = r0 + min(lane / DPAD, nrays - 1);
    if (order) r = order[r];               // schedule only
    float s[3], e[3];
    if (seg) {
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
    int count = 0, n_staged = 0;
    if (MAPMODE != 0) {
        count = min(rvc[r], p.M);
        // (as k_sweep_map: rays with <= 1 voxels are never read on the resident path)
        if (RESIDENT && __all(count <= 1)) return;
        // the first ray's voxel row on its way under the sweep; the others' when their turn comes
        const int c0 = __builtin_amdgcn_readfirstlane(count);
        const int rr = __builtin_amdgcn_readfirstlane(r);
        if (PACKED && !(RESIDENT && c0 <= 1))
            n_staged = stage_voxel_row(p, vox + (size_t)rr * p.M, c0, vals, lane);
    }
    sweep_coop<NV, LPS, RESIDENT, RPW>(p, fv, fv_table, P, s, e, lane, Sl, nrays);
    wave_sync();
    softmax_columns<RESIDENT, RPW>(p.D, lane, Sl);
    wave_sync();
    if (MAPMODE == 0) {          // K7 / K9 / K10: no per-voxel tail, the rays stay side by side
        planes_out_packed<RPW>(p, r, s, e, Sl, lane, nrays, cc, S_planes, depth_from_planes, points);
        return;
    }
    for (int q = 0; q < nrays; q++) {
        // ray q's row, segment and count from its first lane into SGPRs: the tail is k_sweep_map's
        const int rq = __builtin_amdgcn_readlane(r, q * DPAD);
        float su[3], eu[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            su[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s[i]), q * DPAD));
            eu[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e[i]), q * DPAD));
        }
        const int cq = __builtin_amdgcn_readlane(count, q * DPAD);
        if (RESIDENT && cq <= 1) continue;
        const int32_t *vrow = vox + (size_t)rq * p.M * (PACKED ? 1 : 3);
        if (PACKED && q > 0) {
            wave_sync();                   // the previous ray's tail is done with the rows
            n_staged = stage_voxel_row(p, vrow, cq, vals, lane);
        }
        voxels_out<MAPMODE, PACKED>(p, rq, cq, n_staged, su, eu, Sl + q * p.D, vals, axes, pos, vrow,
                                    S_voxel, msgs_out, o_first, lane);
    }
}
