// raynet_mrf.inl -- the MRF kernels: one BP sweep per ray (K3), the accumulator scatters
// (slab-ordered for ray-index rows, LDS box for patch-ordered rows) and the depth estimate
// (K4 / K2 tail).  Included by raynet_hip.hip inside its anonymous namespace.

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
// a * b + c on 24-bit operands (b wave-uniform): the full-rate v_mad_u32_u24.  Written out
// because the compiler, not knowing the range of a kernel argument, takes `a * b + c` (and
// __umul24) to the quarter-rate v_mad_u64_u32.
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b_uniform, unsigned c) {
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
}
// entry of a voxel in a workgroup's LDS box (d1, d2 wave-uniform extents, all < 1024)
__device__ __forceinline__ unsigned box_index(int dx, int dy, int dz, int d1, int d2) {
    return mad_u24(mad_u24((unsigned)dx, (unsigned)d1, (unsigned)dy), (unsigned)d2, (unsigned)dz);
}
template <bool BRICK>
__device__ __forceinline__ int lin_xyz(const Params &p, int x, int y, int z) {
    // 24-bit multiplies: coordinates < 1024, 256 bricks per axis at most
    if (BRICK)
        return (int)((mad_u24(mad_u24((unsigned)x >> 2, (unsigned)p.nby, (unsigned)y >> 2),
                              (unsigned)p.nbz, (unsigned)z >> 2) << 6) |
                     ((x & 3) << 4) | ((y & 3) << 2) | (z & 3));
    return (x * p.gy + y) * p.gz + z;
}
template <bool BRICK>
__device__ __forceinline__ int lin_of(const Params &p, int v) {
    if (BRICK) {
        // straight from the packed word: 8-bit brick coordinates (provably 24-bit operands for
        // the multiplies) and the three 2-bit offsets inside the brick
        const unsigned u = (unsigned)v;
        const unsigned b = mad_u24(mad_u24((u >> 22) & 255u, (unsigned)p.nby, (u >> 12) & 255u),
                                   (unsigned)p.nbz, (u >> 2) & 255u);
        return (int)((b << 6) | ((u >> 16) & 48u) | ((u >> 8) & 12u) | (u & 3u));
    }
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
// flight; requesting the first 64 entries of the rows before the ray's count is known; all
// kernel arguments in one scalar fetch up front -- no gain, the limit was never the dependent
// round trips (profiles/r02_exp_wave_startup.txt).  What did help is not reading gridDim /
// blockDim at all: ray_of_wave.)
#define RN_BP_NT true
#define RN_DEPTH_NT true
// bodies of up to this many chunks issue all their accumulator gathers back to back (bp_ray)
#define RN_GATHER_BATCH_MAX 6
template <int NCH>
struct RayRows {
    float sv[NCH], mv[NCH];
    int pk[NCH];
};
// A ray's rows start at a wave-uniform address; element i of a row is addressed as that
// uniform base + a 32-bit byte offset per lane (global_load_dword v, v_off, s[base:base+1]):
// no 64-bit address arithmetic on the VALU, which is what these kernels are short of.
typedef const __attribute__((address_space(1))) char *gbytes;
// NT: streamed once by this kernel -- kept out of the L2 ways the accumulator gathers live in
template <bool NT = false, typename T>
__device__ __forceinline__ T row_load(const T *row_uniform, unsigned i) {
    typedef const __attribute__((address_space(1))) T *gT;
    if (NT) return __builtin_nontemporal_load((gT)((gbytes)row_uniform + i * (unsigned)sizeof(T)));
    return *(gT)((gbytes)row_uniform + i * (unsigned)sizeof(T));
}
template <bool NT = false, typename T>
__device__ __forceinline__ void row_store(T *row_uniform, unsigned i, T v) {
    typedef __attribute__((address_space(1))) T *gT;
    typedef __attribute__((address_space(1))) char *gb;
    if (NT) __builtin_nontemporal_store(v, (gT)((gb)row_uniform + i * (unsigned)sizeof(T)));
    else *(gT)((gb)row_uniform + i * (unsigned)sizeof(T)) = v;
}
// the first `count` entries of the ray's column, voxel list and (optionally) messages
template <int NCH, bool PACKED, bool NT = false, bool ALL_ROWS = false>
__device__ __forceinline__ void load_rows(const Params &p, RayRows<NCH> &R,
                                          const float *__restrict__ S,
                                          const int32_t *__restrict__ vox, const float *msgs, int r,
                                          int count, int lane, bool need_vox = true) {
    const float *Srow = S + (size_t)r * p.M;
    const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
    const float *mrow = msgs ? msgs + (size_t)r * p.M : nullptr;
    // the voxel words first, for all chunks: the accumulator gathers depend on them and on
    // nothing else, and loads return in the order they were issued -- the columns and messages
    // are still in flight while the gathers go out
    constexpr bool VOX_FIRST = NCH <= RN_GATHER_BATCH_MAX;
#pragma unroll
    for (int ch = 0; ch < (VOX_FIRST ? NCH : 0); ch++) {
        const int i = ch * WAVE + lane;
        R.pk[ch] = 0;
        // (a sweep over ONE constant accumulator value gathers nothing: no voxel row)
        if ((ALL_ROWS || need_vox) && ch * WAVE < count && i < count) {
            if (PACKED) R.pk[ch] = row_load<NT>(vrow, (unsigned)i);
            else R.pk[ch] = load_packed<PACKED>(vrow, i);
        }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const int i = ch * WAVE + lane;
        R.sv[ch] = 0.0f; R.mv[ch] = 0.0f;
        if (!VOX_FIRST) R.pk[ch] = 0;
        if (ch * WAVE < count && i < count) {
            R.sv[ch] = row_load<NT>(Srow, (unsigned)i);
            if (!VOX_FIRST && (ALL_ROWS || need_vox)) {
                if (PACKED) R.pk[ch] = row_load<NT>(vrow, (unsigned)i);
                else R.pk[ch] = load_packed<PACKED>(vrow, i);
            }
            if (ALL_ROWS || mrow) R.mv[ch] = row_load<NT>(mrow, (unsigned)i);
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
        // (the caller's raw column: the literal min / max, which also takes a signalling NaN to eps)
        if (i < count) v = fminf(fmaxf(sv[ch], (float)1e-5), (float)(1 - 1e-5));
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
// STEADY: an iteration after the first of the plan path -- messages exist, every voxel's
// accumulator entry is gathered, the buffer holds sums and the prior is added at the gather.
// Known at compile time, none of it is tested per chunk (as run-time flags each cost the load
// section of a body two vector instructions and a branch per chunk and per row).
template <int NB, bool PACKED, bool CLIP_IN, bool STEADY = false>
__device__ __forceinline__ void bp_ray(const Params &p, int r, int count, int lane,
                                       const float *__restrict__ S,
                                       const int32_t *__restrict__ vox,
                                       const float *__restrict__ acc_in, const float *msgs_in,
                                       float *msgs_out, bool uniform_acc, float acc_bias,
                                       bool biased) {
    if (STEADY) {
        uniform_acc = false;
        biased = true;
    }
    RayRows<NB> cur;
    load_rows<NB, PACKED, RN_BP_NT, STEADY>(p, cur, S, vox, msgs_in, r, count, lane, !uniform_acc);
    // accumulator gather (depends on the voxel rows).  uniform_acc: every voxel holds
    // acc_in[0] (the first iteration starts from the prior everywhere) -- nothing to gather,
    // and with zero messages on top the occupancy is one constant for the whole sweep.
    float av[NB];
    const float a0 = uniform_acc ? (biased ? acc_bias : acc_in[0]) : 0.0f;
#pragma unroll
    for (int ch = 0; ch < NB; ch++) av[ch] = a0;
    if (!uniform_acc) {
        // All chunks' gathers are issued back to back and waited for ONCE: entries beyond the
        // count hold voxel word 0 (load_rows) and gather accumulator entry 0 -- valid memory,
        // the lanes are masked where the value would be used.  (Guarded per chunk, every
        // gather sat in its own block with the addition that uses it and the wavefront paid
        // NB dependent round trips in a row.)
        // (Bodies of more than RN_GATHER_BATCH_MAX chunks -- config 4's M = 768 -- keep the
        // guarded form: twelve gathers at once into a 67 MB accumulator were measured slower,
        // k_bp 9.42 -> 9.80 ms per step there against 1.571 -> 1.515 at config 2.)
        if (NB <= RN_GATHER_BATCH_MAX) {
#pragma unroll
            for (int ch = 0; ch < NB; ch++) av[ch] = gather_acc(acc_in, lin_of<PACKED>(p, cur.pk[ch]));
            // acc_in holds the messages' SUM only and the prior is added here (the same
            // `prior + sum` rn_acc_combine stores, without that kernel and its 3 G floats)
            if (biased) {
#pragma unroll
                for (int ch = 0; ch < NB; ch++) av[ch] = acc_bias + av[ch];
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < NB; ch++) {
                const int i = ch * WAVE + lane;
                if (ch * WAVE < count && i < count) {
                    av[ch] = gather_acc(acc_in, lin_of<PACKED>(p, cur.pk[ch]));
                    if (biased) av[ch] = acc_bias + av[ch];
                }
            }
        }
    }
    // const_o (iteration 0 of a pass: the prior everywhere, no messages yet): one occupancy
    // for the whole sweep.  Real (uniform) branches on it: as selects, every ray pays for the
    // constant's exponential AND every chunk for the per-voxel ones, whichever is used.
    const bool const_o = !STEADY && uniform_acc && msgs_in == nullptr;
    float o_const = 0.0f;
    if (const_o) {
        asm volatile("" ::: "memory");      // (keeps the branch a branch)
        o_const = occupancy_to_ray(a0, 0.0f);
    }
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
            if (!const_o) {
                asm volatile("" ::: "memory");
                o = occupancy_to_ray(av[ch], cur.mv[ch]);
            }
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
                const float m = bp_log_ratio(pos, neg);
                row_store<RN_BP_NT>(mout_row, (unsigned)i, m);
            }
        }
    }
}

template <int NCH, bool PACKED, bool CLIP_IN, bool STEADY = false>
__global__ __launch_bounds__(RAY_BLOCK) void k_bp(Params p, int n, const float *__restrict__ S,
                                              const int32_t *__restrict__ vox,
                                              const int32_t *__restrict__ rvc,
                                              const float *__restrict__ acc_in,
                                              const float *msgs_in, float *msgs_out,
                                              int uniform_acc, float acc_bias, int biased,
                                              float4 *zero_buf, int zero_count4) {
    int lane;
    // zero_buf: the partial accumulator the NEXT scatter adds into (nobody reads it during this
    // launch) is cleared here, a float4 per lane of the first wavefronts, instead of by a
    // kernel of its own between the sweeps
    if (zero_buf) {
        constexpr int WPB = RAY_BLOCK / WAVE;
        const int nw = (n + WPB - 1) / WPB * WPB;
        const int w = blockIdx.x * WPB + (int)(threadIdx.x >> 6);
        for (int i = w * WAVE + (int)(threadIdx.x & (WAVE - 1)); i < zero_count4; i += nw * WAVE)
            zero_buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int r = ray_of_wave<RAY_BLOCK, RN_XCD_CHUNK_BP>(n, lane);
    if (r < 0) return;
    const int count = min(uniform(rvc[r]), p.M);
    if (count <= 1) return;   // mrf_np.py:300: such rays send nothing
    const int nch = (count + WAVE - 1) / WAVE;
#define RN_BP_BODY(NB) \
    bp_ray<NB, PACKED, CLIP_IN, STEADY>(p, r, count, lane, S, vox, acc_in, msgs_in, msgs_out, uniform_acc != 0, \
                                acc_bias, biased != 0)
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
#define RN_SLAB_STEPS 32
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
                if (emit) {
                    if (tail)
                        __hip_atomic_fetch_add(acc_out + lin, val, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
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
// round trip (whole cache lines) and is the default with 4096 voxels (32 KB + 720 B of tables and
// 104 VGPRs: 4 workgroups per CU; 5 or 6 with smaller boxes were measured the same, DESIGN.md
// section 5); scenes whose bundles do not fit (fine grids, oblique views) first get 6144 voxels, then
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
                                                       unsigned *overflow_stats, int BOX_CAP,
                                                       const int2 *__restrict__ slab_boxes,
                                                       const int32_t *__restrict__ items) {
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
    // Work list (rn_scene_bind_scatter_items): one workgroup = `cn` consecutive chunks of one tile,
    // the list sorted longest first -- a tile of long rays (12 chunks at M = 384) is several
    // items, a border tile one, and no workgroup is started for chunks no ray of its tile reaches.
    // Without a list: grid.x = tiles, grid.y workgroups share a tile taking every grid.y-th chunk.
    int r0, chunk0 = (int)blockIdx.y, chunk_step = (int)gridDim.y, chunk_end = 1 << 20;
    if (items) {
        const int item = items[xcd_block<RN_XCD_CHUNK_SCATTER>(blockIdx.x, gridDim.x)];
        r0 = (item >> 12) * BOX_RAYS;
        chunk0 = (item >> 6) & 63;
        chunk_end = chunk0 + (item & 63);
        chunk_step = 1;
    } else {
        r0 = xcd_block<RN_XCD_CHUNK_SCATTER>(blockIdx.x, gridDim.x) * BOX_RAYS;
    }
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
    // longest sending ray of every 64-row half (wave w loaded rows 64 w ... 64 w + 63 above:
    // BOX_RAYS <= BLOCK): the slabs the traversal certainly wrote for that half
    int half_true[(BOX_RAYS + WAVE - 1) / WAVE];
#pragma unroll
    for (int hb = 0; hb < (BOX_RAYS + WAVE - 1) / WAVE; hb++) half_true[hb] = uniform(red_cnt[hb]);
#pragma unroll
    for (int k = 0; k < WAVES_PER_BLOCK; k++) maxc = max(maxc, red_cnt[k]);
    maxc = uniform(maxc);

    // the LDS box is zero whenever a chunk starts: zeroed once here, and every flush clears
    // what it read (no zero pass and one barrier less per chunk)
    for (int i = tid; i < BOX_CAP; i += BLOCK) box[i] = 0;
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
            const typename Sum::box_t bv = box[i];
            box[i] = 0;                 // the box is all zero again when the next chunk starts
            Sum::flush(acc_out, lin_xyz<PACKED>(p, lo0 + i0, lo1 + i1, lo2 + i2), bv);
            i2 += sz;
            if (i2 >= d2) { i2 -= d2; i1++; }
            i1 += sy;
            if (i1 >= d1) { i1 -= d1; i0++; }
            i0 += sx;
        }
    };
    // gridDim.y workgroups share a tile, taking every gridDim.y-th chunk: small launches (a
    // rank of a multi-GPU run) still fill the chip
    // this chunk's (message, voxel) pairs into registers
    auto load_pairs = [&](int s0, float (&m)[BOX_NB], int (&v)[BOX_NB], unsigned &okmask) {
        const int st = s0 + col;
        okmask = 0;
#pragma unroll
        for (int k = 0; k < BOX_NB; k++) {
            const bool ok = st < cnts[j0 + k * STRIDE];
            okmask |= (unsigned)ok << k;
            // rows of padding / short rays are read at the tile's first row: valid memory
            const int rr = ok ? r0 + j0 + k * STRIDE : r0, ss = ok ? st : 0;
            m[k] = msgs[(size_t)rr * p.M + ss];
            v[k] = load_packed<PACKED>(vox + (size_t)rr * p.M * (PACKED ? 1 : 3), ss);
        }
    };
    auto process = [&](int s0, const float (&m)[BOX_NB], const int (&v)[BOX_NB], unsigned okmask) {
        const int st = s0 + col;
        int lo0 = 1 << 30, lo1 = 1 << 30, lo2 = 1 << 30, hi0 = -1, hi1 = -1, hi2 = -1;
        if (slab_boxes) {
            // the traversal left the box of every (64 rows, 16 steps) slab: merge the tile's
            // (wave-uniform loads; a 64-row half is read only up to ITS longest ray -- slabs
            // beyond that were never written)
            constexpr int HB = (BOX_RAYS + WAVE - 1) / WAVE, SL = BOX_STEPS / SLAB_BOX_STEPS;
            const int nsl = slab_box_count(p.M);
            const int2 *tb = slab_boxes + (size_t)(r0 / WAVE) * nsl + s0 / SLAB_BOX_STEPS;
#pragma unroll
            for (int hb = 0; hb < HB; hb++) {
#pragma unroll
                for (int sl = 0; sl < SL; sl++) {
                    if (s0 + sl * SLAB_BOX_STEPS < half_true[hb]) {
                        const int2 bx = tb[(size_t)hb * nsl + sl];
                        lo0 = min(lo0, bx.x >> 20); lo1 = min(lo1, (bx.x >> 10) & 1023);
                        lo2 = min(lo2, bx.x & 1023);
                        if (bx.y >= 0) {
                            hi0 = max(hi0, bx.y >> 20); hi1 = max(hi1, (bx.y >> 10) & 1023);
                            hi2 = max(hi2, bx.y & 1023);
                        }
                    }
                }
            }
            lo0 = uniform(lo0); lo1 = uniform(lo1); lo2 = uniform(lo2);
            hi0 = uniform(hi0); hi1 = uniform(hi1); hi2 = uniform(hi2);
            __syncthreads();          // every thread has finished the previous flush
        } else {
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
        }
        if (hi0 < 0) return;          // (cannot happen below maxc; uniform anyway)
        const int d0 = hi0 - lo0 + 1, d1 = hi1 - lo1 + 1, d2 = hi2 - lo2 + 1;
        const int V = d0 * d1 * d2;
        if (V <= BOX_CAP) {
#pragma unroll
            for (int k = 0; k < BOX_NB; k++)
                if (okmask >> k & 1) {
                    const int x = v[k] >> 20, y = (v[k] >> 10) & 1023, z = v[k] & 1023;
                    __hip_atomic_fetch_add(box + box_index(x - lo0, y - lo1, z - lo2, d1, d2),
                                           Sum::from_msg(m[k]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            __syncthreads();
            flush_box(lo0, lo1, lo2, d0, d1, d2);
            return;
        }
        if (tid == 0 && overflow_stats) atomicAdd(overflow_stats + 1, 1u);
        // ---- too big for LDS.  With the traversal's slab boxes at hand the chunk is done again
        // slab by slab (64 rows x 16 steps each): every piece's box is already known -- no scan,
        // no workgroup reduction -- and the pairs are still in registers.  (Measured on the
        // shard of an 8-rank run whose 17 overflowing chunks of 7472, boxes of 4.1 - 5.2 k
        // voxels, sat in a few tiles: the re-scanning path below made their workgroups the
        // launch's tail, 0.39 ms against 0.21 without overflows.)
        if (slab_boxes) {
            constexpr int HB = (BOX_RAYS + WAVE - 1) / WAVE, SL = BOX_STEPS / SLAB_BOX_STEPS;
            constexpr int KPH = WAVE / STRIDE;          // of a thread's rows, those per 64-row half
            const int nsl = slab_box_count(p.M);
            const int2 *tb = slab_boxes + (size_t)(r0 / WAVE) * nsl + s0 / SLAB_BOX_STEPS;
#pragma unroll
            for (int hb = 0; hb < HB; hb++) {
#pragma unroll
                for (int sl = 0; sl < SL; sl++) {
                    if (!(s0 + sl * SLAB_BOX_STEPS < half_true[hb])) continue;      // (uniform)
                    const int2 bx = tb[(size_t)hb * nsl + sl];
                    if (uniform(bx.y) < 0) continue;
                    const int a0 = uniform(bx.x >> 20), a1 = uniform((bx.x >> 10) & 1023),
                              a2 = uniform(bx.x & 1023);
                    const int e0 = uniform(bx.y >> 20) - a0 + 1,
                              e1 = uniform((bx.y >> 10) & 1023) - a1 + 1,
                              e2 = uniform(bx.y & 1023) - a2 + 1;
                    const bool fits = e0 * e1 * e2 <= BOX_CAP;
                    const bool mine = col / SLAB_BOX_STEPS == sl;
                    __syncthreads();                    // the previous piece's flush is over
#pragma unroll
                    for (int k = hb * KPH; k < (hb + 1) * KPH && k < BOX_NB; k++)
                        if (mine && (okmask >> k & 1)) {
                            const int x = v[k] >> 20, y = (v[k] >> 10) & 1023, z = v[k] & 1023;
                            if (fits)
                                __hip_atomic_fetch_add(box + box_index(x - a0, y - a1, z - a2, e1, e2),
                                                       Sum::from_msg(m[k]), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            else
                                Sum::direct(acc_out, lin_of<PACKED>(p, v[k]), m[k]);
                        }
                    if (fits) {
                        __syncthreads();
                        flush_box(a0, a1, a2, e0, e1, e2);
                    }
                }
            }
            return;
        }
        // ---- without slab boxes (rows that are not patch-ordered, lists from elsewhere): the
        // chunk again in quarters, pairs re-read (L2-hot); a quarter that still does not fit
        // takes the direct atomics
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
#pragma unroll 1
            for (int k = 0; k < BOX_NB; k++) {
                const int j = j0 + k * STRIDE;
                if (mine && st < cnts[j]) {
                    const size_t row = (size_t)(r0 + j) * p.M;
                    const float mm = msgs[row + st];
                    const int pv = load_packed<PACKED>(vox + row * (PACKED ? 1 : 3), st);
                    const int x = pv >> 20, y = (pv >> 10) & 1023, z = pv & 1023;
                    if (fits)
                        __hip_atomic_fetch_add(box + box_index(x - lo0, y - lo1, z - lo2, e1, e2),
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
    };
    // (Measured, profiles/r02_exp_scatter_prefetch.txt: keeping the NEXT chunk's pairs in flight
    // while working on the current one -- the LDS box, not the registers, limits this kernel to
    // four waves per SIMD -- changes nothing: 1.46 against 1.43 ms per step.)
    const int s_end = min(maxc, chunk_end * BOX_STEPS);
    for (int s0 = chunk0 * BOX_STEPS; s0 < s_end; s0 += chunk_step * BOX_STEPS) {
        float m[BOX_NB];
        int v[BOX_NB];
        unsigned okmask;
        load_pairs(s0, m, v, okmask);
        process(s0, m, v, okmask);
    }
    if (tid == 0 && overflow_stats) {
        const int live = max(0, (s_end + BOX_STEPS - 1) / BOX_STEPS - chunk0);      // chunks from chunk0 on
        atomicAdd(overflow_stats, (unsigned)((live + chunk_step - 1) / chunk_step));
    }
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
// (STEADY: the plan path's depth sweep -- messages exist, sums + prior, no distribution written)
template <int NB, bool PACKED, bool CLIP_IN, bool STEADY = false>
__device__ __forceinline__ void depth_ray(const Params &p, int r, int count, int lane,
                                          const float *__restrict__ S,
                                          const int32_t *__restrict__ vox,
                                          const float *__restrict__ acc,
                                          const float *__restrict__ msgs, float *S_new, float &best,
                                          int &best_i, int &best_pk, float acc_bias, bool biased) {
    if (STEADY) {
        biased = true;
        S_new = nullptr;
    }
    RayRows<NB> cur;
    // (rows streamed once, like k_bp's: non-temporal, out of the L2 ways the gathers live in --
    // k_depth 0.742 -> 0.715 ms per step; -DRN_DEPTH_NT=false: plain loads)
    load_rows<NB, PACKED, RN_DEPTH_NT, STEADY>(p, cur, S, vox, msgs, r, count, lane);
    float av[NB];
    if (NB <= RN_GATHER_BATCH_MAX) {
        // (all gathers back to back, entries beyond the count gather entry 0: see bp_ray)
#pragma unroll
        for (int ch = 0; ch < NB; ch++) av[ch] = gather_acc(acc, lin_of<PACKED>(p, cur.pk[ch]));
        if (biased) {
#pragma unroll
            for (int ch = 0; ch < NB; ch++) av[ch] = acc_bias + av[ch];
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < NB; ch++) {
            const int i = ch * WAVE + lane;
            av[ch] = 0.0f;
            if (ch * WAVE < count && i < count) {
                av[ch] = gather_acc(acc, lin_of<PACKED>(p, cur.pk[ch]));
                if (biased) av[ch] = acc_bias + av[ch];      // see bp_ray
            }
        }
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
            if (S_new) row_store(S_new + (size_t)r * p.M, (unsigned)i, d);
            if (d > best) {   // ascending i per lane: keeps the first maximum
                best = d;
                best_i = i;
                best_pk = cur.pk[ch];     // (its voxel word: no second trip to the list)
            }
        }
    }
}

// where a ray's depth goes: depth_map[row] (pixel_of_row == nullptr) or, per group of
// rays_per_center rows (one reference image), depth_map[group * image_stride + pixel_of_row[row
// within the group]] for the group's first `rows` rows (the rest are padding)
struct DepthDest {
    const int32_t *pixel_of_row = nullptr;
    int rows = 0;
    int64_t image_stride = 0;
};
template <int NCH, bool PACKED, bool CLIP_IN, bool STEADY = false>
__global__ __launch_bounds__(RAY_BLOCK) void k_depth(Params p, int n, const float *S,
                                                 const int32_t *__restrict__ vox,
                                                 const int32_t *__restrict__ rvc,
                                                 const float *__restrict__ acc,
                                                 const float *__restrict__ msgs,
                                                 const float *__restrict__ axes,
                                                 const float *__restrict__ cc, float *S_new,
                                                 float *depth_map, int rays_per_center,
                                                 float acc_bias, int biased, int cc_stride,
                                                 DepthDest dest) {
    int lane;
    const int r = ray_of_wave<RAY_BLOCK, RN_XCD_CHUNK_DEPTH>(n, lane);
    if (r < 0) return;
    const int group = rays_per_center > 0 ? r / rays_per_center : 0;
    if (rays_per_center > 0 && cc) cc += (size_t)cc_stride * group;
    // where the depth goes: fetched HERE (a scalar load under the row loads), not behind the
    // arg-max -- a dependent round trip at the end of every wavefront's life costs the launch 3 %
    int64_t out_at = r;
    if (dest.pixel_of_row) {
        const int lr = r - group * rays_per_center;
        out_at = lr < dest.rows
                     ? (int64_t)group * dest.image_stride + uniform(dest.pixel_of_row[lr]) : -1;
    }
    const int count = min(uniform(rvc[r]), p.M);
    float best = -INFINITY;
    int best_i = 0, best_pk = 0;
    if (count > 1) {
        const int nch = (count + WAVE - 1) / WAVE;
#define RN_DE_BODY(NB) \
    depth_ray<NB, PACKED, CLIP_IN, STEADY>(p, r, count, lane, S, vox, acc, msgs, S_new, best, best_i, best_pk, \
                                   acc_bias, biased != 0)
        RN_DISPATCH_CHUNKS(NCH, nch, RN_DE_BODY);
#undef RN_DE_BODY
    } else if (S_new) {
        // mrf_np.py:370-377: skipped rays keep an all-zero row (first `count` entries)
        for (int i = lane; i < count; i += WAVE) S_new[(size_t)r * p.M + i] = 0.0f;
    }
    if (!depth_map) return;
    // raynet_fp.py:193-226.  Entries beyond count are zero in the reference's zero-filled
    // buffer and every d_i > 0, so the arg-max lies in [0, count); for count <= 1 the row is
    // all zeros and index 0 wins.  First maximum of the wave: the largest value (DPP
    // reduction), then the smallest index among the lanes that hold it -- two reductions on
    // the VALU instead of a six-step shuffle butterfly through LDS.
    // The winning voxel's word is in the registers of the lane that holds the maximum: it is
    // passed on from there instead of being read from the list again (one dependent round trip
    // less at the end of every wavefront's life).
    int won_pk = -1;
    {
        const float top = wave_max(best);
        const int mine = best_i;
        best_i = wave_min_i(best == top ? best_i : 0x7fffffff);
        const unsigned long long who = __ballot(best == top && mine == best_i);
        if (who) won_pk = __builtin_amdgcn_readlane(best_pk, (int)__builtin_ctzll(who));
        if (best_i == 0x7fffffff) best_i = 0;       // (a NaN column: nobody equals the maximum)
    }
    if (lane == 0) {
        const int32_t *vrow = vox + (size_t)r * p.M * (PACKED ? 1 : 3);
        int x = 0, y = 0, z = 0;
        if (count > 1 && won_pk >= 0) {
            x = won_pk >> 20; y = (won_pk >> 10) & 1023; z = won_pk & 1023;
        } else if (count > 0) {
            load_voxel<PACKED>(vrow, count > 1 ? best_i : 0, x, y, z);
        }
        const float pt[3] = {axes[x], axes[p.gx + y], axes[p.gx + p.gy + z]};
        float sum = 0.0f;
        for (int i = 0; i < 3; i++) {
            const float d = pt[i] - cc[i];
            sum += d * d;
        }
        // row order, or the map in PIXEL order (forward_pass.py:744 hands out `.reshape(W, H).T`
        // of the ray-index-ordered vector): no reordering pass behind the sweep
        if (out_at >= 0) depth_map[out_at] = sqrtf(sum);
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

