// valu_issue_bench.hip -- how fast does ONE SIMD of gfx950 issue wave64 VALU instructions?
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_bench.hip -o /tmp/valu_issue_bench && /tmp/valu_issue_bench
// Streams of instructions without dependences between neighbours (16 accumulators in turn):
// v_fma_f32, v_pk_fma_f32, v_add_f32_dpp (the scans' instruction) and v_rcp_f32 (transcendental
// rate), at 1 / 2 / 4 / 8 wavefronts per SIMD, every CU of the chip busy.  Reported per variant:
// wave-instructions per second and SIMD (hipEvent wall time), and cycles per instruction by the
// wave's own s_memtime and by the nominal 2.4 GHz -- the figure bench.py's VALU-issue model
// (valu_issue_ms) and DESIGN.md section 5 use.  (MI355X_MICROARCH.md: "4 SIMD-32 units ... issues
// each VALU instruction over 2 cycles"; rounds 1-3 of this repo assumed 4.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 16;      // independent accumulators
constexpr int INNER = 64;       // instructions per accumulator and outer iteration -> 1024 per iteration

typedef float f2 __attribute__((ext_vector_type(2)));

// one instruction of the stream on accumulator a (f2: the packed ones use both halves)
#define RN_KINDS(X)                                                                                  \
    X(0, "v_fma_f32", asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(m), "v"(c)))        \
    X(1, "v_mul_f32", asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a.x) : "v"(m)))                    \
    X(2, "v_add_f32", asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.x) : "v"(c)))                    \
    X(3, "v_max_f32", asm volatile("v_max_f32 %0, %0, %1" : "+v"(a.x) : "v"(c)))                    \
    X(4, "v_med3_f32", asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(c), "v"(m)))      \
    X(5, "v_add_u32", asm volatile("v_add_u32 %0, %0, %1" : "+v"(a.x) : "v"(7)))                    \
    X(6, "v_and_b32", asm volatile("v_and_b32 %0, %0, %1" : "+v"(a.x) : "v"(0x7fffffff)))           \
    X(7, "v_lshlrev_b32", asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a.x)))                      \
    X(8, "v_mov_b32", asm volatile("v_mov_b32 %0, %1" : "=v"(a.x) : "v"(a.y)))                      \
    X(9, "v_cndmask_b32", asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a.x) : "v"(c), "s"(0x5555555555555555ull))) \
    X(10, "v_cmp_lt_f32", asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a.x), "v"(c) : "vcc"))    \
    X(11, "v_cvt_i32_f32", asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a.x)))                        \
    X(12, "v_rndne_f32", asm volatile("v_rndne_f32 %0, %0" : "+v"(a.x)))                            \
    X(13, "v_pk_fma_f32", asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(mm), "v"(cc))) \
    X(14, "v_pk_mul_f32", asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(mm)))              \
    X(15, "v_pk_add_f32", asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(cc)))              \
    X(16, "v_add_f32_dpp", asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a.x))) \
    X(17, "v_mov_b32_dpp", asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a.x) : "v"(a.y))) \
    X(18, "v_mad_u32_u24", asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a.x) : "v"(3), "v"(7))) \
    X(19, "v_mul_lo_u32", asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a.x) : "v"(3)))             \
    X(20, "v_bcnt_u32_b32", asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a.x) : "v"(0)))         \
    X(21, "v_mbcnt_lo_u32_b32", asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a.x) : "s"(0x55555555))) \
    X(22, "v_rcp_f32", asm volatile("v_rcp_f32 %0, %0" : "+v"(a.x)))                                \
    X(23, "v_exp_f32", asm volatile("v_exp_f32 %0, %0" : "+v"(a.x)))                                \
    X(24, "v_log_f32", asm volatile("v_log_f32 %0, %0" : "+v"(a.x)))                                \
    X(25, "v_sqrt_f32", asm volatile("v_sqrt_f32 %0, %0" : "+v"(a.x)))                              \
    X(26, "v_fma_f64", asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(dm)))                \
    X(27, "v_readlane_b32", asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(sg) : "v"(a.x)))        \
    X(28, "v_mfma_f32_32x32x2f32", asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16) : "v"(a.x), "v"(a.y))) \
    X(29, "v_cndmask (vcc)", asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a.x) : "v"(c)))  \
    X(30, "v_cndmask (2 sgpr masks)", if (i & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a.x) : "v"(c), "s"(mask_a)); \
                                      else asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a.x) : "v"(c), "s"(mask_b))) \
    X(31, "v_mfma_f32_4x4x1_16b", asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(q) : "v"(a.x), "v"(a.y))) \
    X(32, "v_mfma_f32_16x16x4_f32", asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(q) : "v"(a.x), "v"(a.y))) \
    X(33, "v_mfma_f32_16x16x1_4b", asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc16) : "v"(a.x), "v"(a.y)))

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void k_stream(int outer, float seed, float *out, unsigned long long *cycles) {
    f2 acc[UNROLL];
    double dd[UNROLL];
    f16v acc16 = {0.f};
    f4v acc4[UNROLL];
    int sg = 0;
    // (two lane masks in SGPR pairs the compiler cannot fold: the select's cost without a chain
    // through ONE mask register -- round 4's row read 23.6 cycles, the benchmark's artefact)
    unsigned long long mask_a, mask_b;
    asm volatile("s_mov_b64 %0, 0x55555555\n\ts_mov_b64 %1, 0x33333333\n\ts_mov_b64 vcc, 0x0f0f0f0f" : "=s"(mask_a), "=s"(mask_b) : : "vcc");
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { acc[i] = f2{seed + i + threadIdx.x, -(seed + i)}; dd[i] = seed + i; acc4[i] = f4v{0.f, 0.f, 0.f, 0.f}; }
    const float m = 0.999f, c = 1e-3f;
    const f2 mm = {m, m}, cc = {c, c};
    const double dm = 0.999;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int o = 0; o < outer; o++) {
#pragma unroll
        for (int k = 0; k < INNER; k++) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) {
                f2 &a = acc[i];
                double &d = dd[i];
                f4v &q = acc4[i];
#define X(ID, NAME, STMT) if (KIND == ID) { STMT; }
                RN_KINDS(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)sg + acc16[0] + acc16[7];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s += acc[i].x + acc[i].y + (float)dd[i] + acc4[i][0] + acc4[i][3];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
void run(const char *name, int cus) {
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 4)); CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-22s", name);
    for (int wps : {1, 2, 4, 8}) {
        // wps wavefronts per SIMD: workgroups of 256 * min(wps, 4) threads, wps / 4 (>= 1) of them per CU
        const int threads = 256 * (wps < 4 ? wps : 4);
        const int blocks = cus * (wps <= 4 ? 1 : wps / 4);
        const int outer = 400;
        hipLaunchKernelGGL((k_stream<KIND>), dim3(blocks), dim3(threads), 0, 0, 10, 1.0f, out, cyc);   // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_stream<KIND>), dim3(blocks), dim3(threads), 0, 0, outer, 1.0f, out, cyc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_simd = (double)outer * INNER * UNROLL * wps;   // wave-instructions issued by one SIMD
        const double rate = per_simd / (ms * 1e-3);                    // ... per second
        printf("  %dw: %6.3f G/s = %5.2f cyc", wps, rate * 1e-9, 2.4e9 / rate);
    }
    printf("\n");
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, clock %d MHz (reported).  Per instruction and wavefronts per SIMD (1w .. 8w): "
           "wave64 instructions per second and SIMD, and cycles per instruction at the nominal 2.4 GHz\n",
           p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
#define X(ID, NAME, STMT) run<ID>(NAME, p.multiProcessorCount);
    RN_KINDS(X)
#undef X
    return 0;
}
