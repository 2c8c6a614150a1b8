// pre3_cholp.hip -- update.m:32-33 (S = L L', W = L^-1 [HP | nu]) as ONE persistent launch (fp32 contexts, r <= 13 panels of 64).
//
// The launch-per-panel form (k_chol_step, pre3_update.hip) carries all W strips through every panel launch in lock-step with the
// r x r factorisation and pays a launch boundary, a cold reload and a ramp per panel.  Here the r x r factorisation is a task graph
// whose critical path never leaves ONE workgroup, and W is a blocked triangular solve on the matrix cores that trails it:
//
//   crit   (block 0, 12 waves)   for J = 0 .. nrb-1:  chain on the diagonal block D_J (chol_chain, pre3_chain.h) with X := I, so that
//          M_J = inv(L_JJ) falls out of the same lock-step solve;  then, on the bf16 matrix cores from planes in LDS,
//          L(J+1,J) = A(J+1,J) M_J' and D_{J+1} = A(J+1,J+1) - L(J+1,J) L(J+1,J)'.  Nothing on this path waits for another workgroup:
//          the two tiles of row J+1 it needs (updated through panel J-1) are fetched by its two side waves WHILE the chain of panel J
//          runs, and M_J / L(J+1,J) leave as bf16 planes through the same side waves.  (Round 5: the first product is computed transposed and
//          its accumulators become the result's plane granules by v_permlane32_swap, with no trip through LDS; the second product runs as the
//          next chain's first act on the waves that are that chain's D workers, and D_{J+1} never leaves their accumulators.)
//   row i  (blocks 1.., i >= 2, 4 waves)  for J = 0 .. i-2:  L(i,J) = A(i,J) M_J' once M_J is published, then A(i,k) -= L(i,J) L(k,J)'
//          for k = J+1 .. i.  After panel i-2 the row's two leading tiles go to crit (A(i,i-1) as planes, A(i,i) as f32).
//   strip s (32 columns of [HP | nu], 4 waves)  for J = 0 .. nrb-1:  W_J = M_J (HP_J - sum_{K<J} L(J,K) W_K), the W_K as bf16 planes in
//          LDS; epilogue: W in f32 and as the bf16 planes k_downdate_b3 reads (sc1 stores, then the strip's flag for panel J).  Behind the last
//          panel the strips finish the state (x-update) and -- inside pre3_step -- project every landmark at x_k_k for the rescue stage
//          (strip_proj_body), in the shadow of the consumers' epilogue.
//   down-date consumer g (the CUs the factorisation leaves idle; round 4)  up to twelve 64 x 64 tiles of P's upper triangle, one per wave,
//          accumulators resident for the whole launch: for J = 0 .. nrb-1, as soon as the strips that own the group's column blocks have
//          published W_J, the panel's planes come in by LDS-DMA and acc += W_J' W_J (update.m:37 is a sum over panels); P is read, down-dated
//          and written (with its mirror image) once, behind the last panel.  Same products in the same order as k_downdate_b3: bit-identical.
//          The tiles of block row 0 also leave rows 3..6 -- before update.m:42-46 -- in a side buffer for the chi2 gate that rides with that pass
//          (jn_q; GateRide, pre3_geom.hip).
//
// Every hand-off is: payload by 16-byte sc1 (write-through) stores, each storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier,
// ONE lane stores the flag (agent scope); consumer: one lane polls the flag (sc1), workgroup barrier, then every load of the payload
// is an sc1 buffer load (crit, rows, and since round 5 the strips' fragments of L) or a plain load behind ONE agent-scope acquire (consumers:
// they share the planes through their XCD's L2).  Flags are monotonic words tagged with a per-launch epoch; every wait is bounded
// (guard word stats[7], as in pre3_geomdev.h) and no access depends on a value that a give-up would leave undefined.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "pre3_internal.h"
#include "pre3_geomdev.h"
#include "pre3_chain.h"
#include "pre3_chain_async.h"
#include "pre3_cholp.h"

namespace pre3 {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef int frag_t __attribute__((ext_vector_type(4)));      // 8 bf16
typedef float f4v_t __attribute__((ext_vector_type(4)));

#ifdef PRE3_PROBE
// wall-clock stamps (s_memrealtime, 100 MHz, chip-wide) of the last launch: [role 0 crit main, 1 crit side, 2..15 rows, 16 strip 0, 17 last strip, 20 consumer 0, 21 last consumer][panel 16][slot 8]
static __device__ unsigned long long g_cp[24 * 16 * 8];
#define CP_CLK(role, J, slot) do { if ((threadIdx.x & 63) == 0 && (J) < 16) { unsigned long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_cp[((role) * 16 + (J)) * 8 + (slot)] = t_; } } while (0)
#define CP_STAMP(role, J, slot) do { if ((threadIdx.x & 63) == 0 && (J) < 16) { unsigned long long t_; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_cp[((role) * 16 + (J)) * 8 + (slot)] = t_; } } while (0)
#else
#define CP_STAMP(role, J, slot)
#define CP_CLK(role, J, slot)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cp_rsrc(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
// 16-byte sc1 load / store at a byte offset (aux 16 = sc1: bypasses this CU's L1 / writes through)
__device__ __forceinline__ u32x4_t ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ __forceinline__ void st16_sc1(u32x4_t v, __amdgpu_buffer_rsrc_t r, unsigned off) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
// four floats at a dword-aligned byte offset (the whole vector is re-typed: __builtin_bit_cast of ONE element of an ext-vector reads element 0)
__device__ __forceinline__ f4v_t ld4f(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f4v_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
__device__ __forceinline__ f4v_t ld4f_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f4v_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16)); }
__device__ __forceinline__ int ld_i32_sc1(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32_sc1(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned cf_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cf_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool cf_reached(unsigned v, unsigned target) { return (int)(v - target) >= 0; }
// one lane: bounded poll (the same budget as bounded_wait)
__device__ __forceinline__ void cf_wait(const unsigned *p, unsigned target, int32_t *guard)
{
    for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
        if (cf_reached(cf_load(p), target)) return;
        // another wait of this launch has already given up: what it was waiting for will not come either, let the launch drain
        if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
    atomicExch(guard, 1);
}
// the whole workgroup waits for a flag: lane 0 polls, everybody meets at the barrier
__device__ __forceinline__ void wg_wait(const unsigned *p, unsigned target, int32_t *guard)
{
    if (threadIdx.x == 0) cf_wait(p, target, guard);
    __syncthreads();
}

// x[0..7] -> the three bf16 planes' granules (as b3_split_store, values returned)
__device__ __forceinline__ void b3_split3(const float (&x)[8], u32x4_t &a, u32x4_t &b, u32x4_t &c)
{
    bf16x8_t pa, pb, pc;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 ha = (__bf16)x[j];
        const float r1 = x[j] - (float)ha;
        const __bf16 hb = (__bf16)r1;
        const float r2 = r1 - (float)hb;
        pa[j] = ha; pb[j] = hb; pc[j] = (__bf16)r2;
    }
    a = __builtin_bit_cast(u32x4_t, pa); b = __builtin_bit_cast(u32x4_t, pb); c = __builtin_bit_cast(u32x4_t, pc);
}

// 64 x 64 x 64 on the bf16 matrix cores, six products per f32 product: acc(row, col) += sum_k A(row, k) B(col, k) for this wave's
// 32 x 32 tile; fA[q][pl] / fB[q][pl]: the wave's fragments of k-step q, plane pl
__device__ __forceinline__ void mma6(const frag_t (&fA)[4][3], const frag_t (&fB)[4][3], f32x16_t &acc, int q0 = 0, int q1 = 4)
{
#define CP_MMA(px, py) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fA[q][px]), __builtin_bit_cast(bf16x8_t, fB[q][py]), acc, 0, 0, 0)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q >= q0 && q < q1) { CP_MMA(0, 0); CP_MMA(0, 1); CP_MMA(1, 0); CP_MMA(1, 1); CP_MMA(0, 2); CP_MMA(2, 0); }
#undef CP_MMA
}
// the same with the A fragments read from an LDS plane block k-step by k-step (12 registers instead of 48)
__device__ __forceinline__ void mma6_alds(const frag_t *Ablk, int half, int lane, const frag_t (&fB)[4][3], f32x16_t &acc)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const frag_t a0 = Ablk[q * 384 + half * 64 + lane], a1 = Ablk[q * 384 + 128 + half * 64 + lane], a2 = Ablk[q * 384 + 256 + half * 64 + lane];
#define CP_MMA(ax, py) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, fB[q][py]), acc, 0, 0, 0)
        CP_MMA(a0, 0); CP_MMA(a0, 1); CP_MMA(a1, 0); CP_MMA(a1, 1); CP_MMA(a0, 2); CP_MMA(a2, 0);
#undef CP_MMA
    }
}
// fragments of one 64 x 64 plane block ([q 4][plane 3][half 2][lane 64] granules) from LDS / from global memory (sc1)
__device__ __forceinline__ void frags_lds(const frag_t *blk, int half, int lane, frag_t (&f)[4][3])
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[q][pl] = blk[q * 384 + pl * 128 + half * 64 + lane];
}
// one granule of a plane block in global memory: the lane's part of the address in a VGPR, the block / k-step / plane part in an SGPR
__device__ __forceinline__ frag_t ld_granule(__amdgpu_buffer_rsrc_t r, unsigned lane_off /* (half * 64 + lane) * 16 */, unsigned granule /* uniform */, int aux)
{
    return __builtin_bit_cast(frag_t, aux ? __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, granule * 16u, 16) : __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, granule * 16u, 0));
}
__device__ __forceinline__ void frags_sc1(__amdgpu_buffer_rsrc_t r, unsigned blk_granule, int half, int lane, frag_t (&f)[4][3])
{
    const unsigned lo = (unsigned)(half * 64 + lane) * 16u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[q][pl] = ld_granule(r, lo, blk_granule + q * 384 + pl * 128, 1);
}
__device__ __forceinline__ void frags_plain(const frag_t *blk, int half, int lane, frag_t (&f)[4][3])
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[q][pl] = blk[q * 384 + pl * 128 + half * 64 + lane];
}

// one f32 of a 32 x 32 accumulator tile in a row-major matrix: the lane's part of the address in a VGPR (voff), the register's part
// (its row) and the tile's origin in an SGPR (soff) -- sixteen 64-bit addresses per tile would cost 32 registers
__device__ __forceinline__ unsigned acc_voff(int lane, int ldm) { return (unsigned)((4 * (lane >> 5)) * ldm + (lane & 31)) * 4u; }
__device__ __forceinline__ unsigned acc_soff(int e, int ldm) { return (unsigned)(((e & 3) + 8 * (e >> 2)) * ldm) * 4u; }
__device__ __forceinline__ float ld_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ void st_f32_sc1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 16); }
__device__ __forceinline__ float ld_f32_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 16)); }
__device__ __forceinline__ void st_f32(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0); }

constexpr int CP_YS = NB + 1;                       // row stride of the f32 transposition buffers
// accumulator layout of a 32 x 32 tile: register e of lane l holds (row (e&3) + 8 (e>>2) + 4 (l>>5), column l & 31)
__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// Y (f32 [64][CP_YS], rows x k) -> plane block: granule job g in [0, 512): (q, half, lane) -> Y[32 half + r][16 q + 8 h + 0..7]
__device__ __forceinline__ void y_granule(const float *Y, int g, u32x4_t &a, u32x4_t &b, u32x4_t &c, int &gi)
{
    const int q = g >> 7, half = (g >> 6) & 1, l = g & 63, r = l & 31, h = l >> 5;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = Y[(32 * half + r) * CP_YS + 16 * q + 8 * h + j];
    b3_split3(x, a, b, c);
    gi = q * 384 + half * 64 + l;                 // plane 0; planes are 128 granules apart
}

// ------------------------------------------------------------------------------------------------------------------------------
// crit
// ------------------------------------------------------------------------------------------------------------------------------
struct CpArgs {
    float *S; float *W; int ldw; int ld;
    void *Sp; int sp_stride;             // plane blocks of S: block (rb, J) at (rb * sp_stride + J) * 1536 granules
    void *Tp;                            // per row: the planes of A(i, i-1) handed to crit, 1536 granules each
    void *Wp; int nst_total;             // k_downdate_b3's planes of W
    unsigned *cf; unsigned base;         // flags, epoch
    int32_t *status;                     // stats[6] (not positive definite), stats[7] (a wait gave up)
    const int32_t *n_dev; int nrb; int nrb_max; int n_strips;
    int win;                             // strips: blocks of W kept in LDS (a ring: block K in slot K % win); older blocks are re-read from Wp
    int stride;                          // crit and the rows are blocks 0, stride, 2 stride, ..
    int n; const double *x_prior; double *x_out; double *params; int xu;      // xu: the strips finish with x_out = x_prior + W'(L^-1 nu) (update.m:36,42,48)
    float *P; const int32_t *dd; int n_dd; int rows; int dd_mode; int poll_budget, poll_from; float *jn_q;      // (jn_q: rows 3..6 of the down-dated P, before update.m:42-46, for the gate that rides with that pass)      // down-date consumers: group table (DG_WORDS each), groups in this launch, rows if host-known
    int row_late;                        // rows: in a row's last panel L(i, J)'s flag goes up with the tiles for crit (round 6)
    int crit_early;                      // crit: the last panel's chain stops behind its last real sub-panel (round 6)
    int strip_rl;                        // strips: the right-looking, flag-driven panel loop (strip_rl_loop; round 6)
    int proj;                            // the strips end with the rescue stage's projection of every landmark at x_k_k (strip_proj_body; tables in CpTail's LDS slot)
    int tail;                            // the rescue stage and the HI update follow inside this launch (CpTail): the HI rows are panel `nrb` of the same factorisation
    float *Wt; int kcap;                 // tail: W once more, column-major (column j at Wt + j * kcap, k contiguous): what the gate's y = H J W' reads
    const void *Wp_pend; int pend_ns;    // PRE3_OPT_PEND_HI: planes of the HI update left pending by the step before (Wp's layout and nst_total) and their k-stages (0: none):
                                         // the consumers take them as the panels in front of panel 0 -- they idle until W_0 is out anyway -- and P is written once for both
};
// mono_slam.m:184-187 inside the launch (round 5).  Once the strips have x_k_k:  (B) every landmark is projected and linearised at x_k_k
// (rescue_hi_inliers.m:32-33); for the candidates (individually compatible, not a low-innovation inlier) y_i = H_i J W' (2 x r), the gate
// nu' inv(H_i J P+ J' H_i') nu < chi2 with H_i J P+ J' H_i' = (H_i J) P (H_i J)' - y_i y_i' (P+ = P - W'W is never formed; J = the normalisation
// Jacobian of update.m:42-46), y_i as bf16 planes (Yp) and the row H_i J (Hb).  (C) crit collects the rescued landmarks (rescue_hi_inliers.m:44-46)
// and publishes them (hib).  (D) their rows are panel nrb of the block factorisation the launch is running anyway: L(hi, K) = Y_hi(:, block K),
// D_hi = (H J) P (H J)' + I - Y_hi Y_hi' on crit, chain -> M_hi; strips: W~ = M_hi ((H J) P - Y_hi W) -- i.e. L_hi^-1 H J P+ --, x += J W~' (L_hi^-1 nu);
// consumers: one more panel, acc += W~' W~, and P - acc is written ONCE.  What is left of update.m:42-46 for BOTH updates -- rows / columns 3..6 <- (J2 J) . --
// is one pending pass (params[96..]: the next prediction's launch applies it, or k_jnorm_P).
struct CpTail {
    int on, N, m, seq, ykcap, pad0;
    double chi2;
    const int32_t *lm_type, *lm_off, *lm_ic, *lm_li, *meas;
    int32_t *lm_hi, *has_h, *hi_meas, *sel_rows, *stats, *mail;
    double *h, *Hc, *Hl; const double *z;
    CamD cam;
    void *Yp;                            // [N + 1][2 rows][3 planes][ykcap] bf16: y_i = H_i J W' of candidate landmark i (slot N stays zero: padding rows)
    float *Hb;                           // [N][2][16]: the row H_i J in ELL order (7 pose + 6 landmark entries), [13] = nu = z - h
#ifdef PRE3_TAIL_DEBUG
    double *dbgS;
#endif
    int32_t *hib;                        // what crit publishes: [0] count, [1] rows in this launch (0: none or more than HF_TAIL_MAXL), [4..35] landmark per entry, [64..] rows [64][16]
};
constexpr int HT_MAXL = 32;              // rescued landmarks the launch itself updates with (one panel); more: the host's general path
constexpr int HT_HIB_WORDS = 64 + 64 * 16;
// A call passes its arguments in VGPRs: the callee makes the (wave-uniform) launch arguments scalar again, word by word
__device__ __forceinline__ CpArgs cp_uniform(const CpArgs &v)
{
    CpArgs a;
    const unsigned *src = reinterpret_cast<const unsigned *>(&v);
    unsigned *dst = reinterpret_cast<unsigned *>(&a);
#pragma unroll
    for (unsigned w = 0; w < sizeof(CpArgs) / 4; ++w) dst[w] = (unsigned)__builtin_amdgcn_readfirstlane((int)src[w]);
    return a;
}
__device__ __forceinline__ CpTail ct_uniform(const CpTail &v)
{
    CpTail a;
    const unsigned *src = reinterpret_cast<const unsigned *>(&v);
    unsigned *dst = reinterpret_cast<unsigned *>(&a);
#pragma unroll
    for (unsigned w = 0; w < sizeof(CpTail) / 4; ++w) dst[w] = (unsigned)__builtin_amdgcn_readfirstlane((int)src[w]);
    return a;
}
// The tail's code reads the launch arguments from LDS (k_cholp puts them there): as by-value arguments of the out-of-line roles they would be a
// hundred more registers live through the chain and the panel loop (measured: 300 scratch instructions inside the chain).
constexpr size_t CP_T_OFF = 160 * 1024 - 1024, CP_TA_OFF = CP_T_OFF + 512;
static_assert(sizeof(CpTail) <= 512 && sizeof(CpArgs) <= 512, "LDS slots of the launch arguments");
constexpr size_t CP_KA_TAIL = (sizeof(CpArgs) + alignof(CpTail) - 1) / alignof(CpTail) * alignof(CpTail);      // k_cholp(CpArgs, CpTail): the second argument's offset in the kernel-argument segment
static_assert(sizeof(CpArgs) % 4 == 0 && sizeof(CpTail) % 4 == 0, "launch arguments are copied word by word");
template <typename S> __device__ __forceinline__ S args_from_lds(size_t off)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    S a;
    const unsigned *src = reinterpret_cast<const unsigned *>(cp_smem + off);
    unsigned *dst = reinterpret_cast<unsigned *>(&a);
#pragma unroll
    for (unsigned w = 0; w < sizeof(S) / 4; ++w) dst[w] = (unsigned)__builtin_amdgcn_readfirstlane((int)src[w]);
    return a;
}
// The strips and the consumers (241 of the launch's 256 workgroups) get the launch arguments as a POINTER to the kernel-argument segment and read them
// by scalar loads.  As a by-value CpArgs they travel through scratch: 240 B stored per thread at the call, 47 MB of dirty lines per launch that the
// consumers' epilogue then pushes out of the L2s -- 39 of the LI launch's 103 MB of HBM writes (round 6: tools/pmc_env_write.sh, tools/pmc_ab_write.sh).
//  * __builtin_amdgcn_kernarg_segment_ptr() inside an out-of-line role returns null with this compiler: the kernel passes the pointer.
//  * not_tail_called: a call that passes no pointer into the caller's frame gets the optimiser's `tail` mark (in tail position or not), and a local
//    function with such a caller loses the "no callee-saved registers" treatment: ~75 registers stored per thread at the role's entry.
//  * the rows (14 workgroups) keep the by-value form: with no by-value call left in k_cholp the register allocation of crit's chain (inlined in the
//    kernel) falls apart -- 405 spilled registers against 51, the X workers' accumulators reloaded in every step (measured: LI launch 97 -> 182 us).
typedef __attribute__((address_space(4))) const unsigned char *cp_ka_t;
template <typename S> __device__ __forceinline__ S args_from_kernarg(cp_ka_t ka_v, size_t off)
{
    typedef __attribute__((address_space(4))) const unsigned char kb_t;
    typedef __attribute__((address_space(4))) const unsigned kw_t;
    const unsigned long long kav = (unsigned long long)ka_v;
    const unsigned long long kau = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(kav >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)kav);
    kw_t *src = (kw_t *)((kb_t *)kau + off);
    S a;
    unsigned *dst = reinterpret_cast<unsigned *>(&a);
#pragma unroll
    for (unsigned w = 0; w < sizeof(S) / 4; ++w) dst[w] = src[w];
    return a;
}
#define CP_ROLE __attribute__((noinline, not_tail_called))
constexpr int CF_MP = 0, CF_ROWL = 32, CF_ROWA = 32 * 65, CF_STRIP = 32 * 130;     // one 128-byte line per flag; the strips' flags follow
constexpr int CF_HI = CF_ROWL;           // (row 0 has no flag of its own) crit's word for the tail: the HI list is out; word [1] = its rows in this launch
// values of a strip's flag behind the panels (base + J + 1, J < 16): x_k_k of its states is out / its landmarks are gated / its block of W~ is out
constexpr unsigned CFV_X = 20, CFV_GATE = 21, CFV_HIL = 22, CFV_HIW = 23;
__device__ __forceinline__ unsigned *cf_rowL(unsigned *cf, int i) { return cf + CF_ROWL + 32 * i; }
__device__ __forceinline__ unsigned *cf_rowA(unsigned *cf, int i) { return cf + CF_ROWA + 32 * i; }
__device__ __forceinline__ unsigned *cf_strip(unsigned *cf, int s) { return cf + CF_STRIP + 32 * s; }

struct CritSmem {
    ChSmem<float> ch;                                            // Ls, Xs, As | pipe, Bs
    __attribute__((aligned(16))) frag_t MPl[B3_SGRAN];           // planes of M_J (built while the chain runs)
    __attribute__((aligned(16))) frag_t T1p[B3_SGRAN];           // planes of A(J+1, J); after B1: the planes of L(J+1, J)
    __attribute__((aligned(16))) unsigned fpoll[64];             // wave 11's view of row J+1's flag: filled by LDS-DMA, read without a memory wait
};

// planes of L(J+1, J), written by the first product straight from its accumulators (round 5), behind CritSmem: the region is the rescue stage's
// between the LI update's last panel and the HI panel (TailSmem | parts), when no product is running
constexpr size_t CP_LP_OFF = (sizeof(CritSmem) + 127) / 128 * 128;
static_assert(CP_LP_OFF + sizeof(frag_t) * B3_SGRAN <= 160 * 1024 - 1024, "crit's LDS with the planes of L(J+1, J)");
__device__ __forceinline__ frag_t *crit_lp(CritSmem &sm) { return reinterpret_cast<frag_t *>(reinterpret_cast<unsigned char *>(&sm) + CP_LP_OFF); }

// panel 0: the raw blocks straight from S (written by the launch in front); all twelve waves
__device__ __forceinline__ void crit_prologue(const CpArgs &a, int nrb, CritSmem &sm)
{
    auto &Ls = sm.ch.Ls; auto &Xs = sm.ch.Xs;
    float *T2 = &sm.ch.Bs[0][0];
    const int tid = threadIdx.x, lds = nrb * NB;
    for (int idx = tid; idx < NB * NB; idx += blockDim.x) {
        const int i = idx >> 6, c = idx & 63;
        Ls[i][c] = a.S[(size_t)i * lds + c];
        Xs[i][c] = i == c ? 1.f : 0.f;
        if (nrb > 1) T2[i * NB + c] = a.S[(size_t)(NB + i) * lds + NB + c];
    }
    if (nrb > 1 && tid < 512) {                                   // planes of A(1, 0): 8 consecutive k of one row per job
        const int q = tid >> 7, half = (tid >> 6) & 1, l = tid & 63, r = l & 31, h = l >> 5;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a.S[(size_t)(NB + 32 * half + r) * lds + 16 * q + 8 * h + j];
        u32x4_t p0, p1, p2;
        b3_split3(x, p0, p1, p2);
        frag_t *d = sm.T1p + q * 384 + half * 64 + l;
        d[0] = __builtin_bit_cast(frag_t, p0); d[128] = __builtin_bit_cast(frag_t, p1); d[256] = __builtin_bit_cast(frag_t, p2);
    }
    __syncthreads();
}

// waves 0-9: the chain and the two products.  Barriers per panel: the chain's ten, then b0, b2 (crit_side keeps the same count).
__device__ __attribute__((noinline)) int crit_tail(int nrb_v);

__device__ __forceinline__ void crit_main(const CpArgs &a, int nrb, CritSmem &sm)
{
    const float *T2 = &sm.ch.Bs[0][0];                           // A(J+1, J+1), f32 [64][64]
    const int tid0 = threadIdx.x;
    if ((tid0 >> 6) == 8) __builtin_amdgcn_s_setprio(3);
    else if ((tid0 >> 6) == 9) __builtin_amdgcn_s_setprio(2);
    bool bad = false;
    // (two passes over ONE instance of the panel loop: the LI update's panels [0, nrb), then -- if the tail brings rescued landmarks -- their rows as
    //  panel nrb.  The rescue stage's call sits BETWEEN the passes: inside the panel loop it cost the chain 4 % of its cycles and the products 0.5 us per
    //  panel, although nothing of it was live there.)
    for (int pass = 0; pass < 2; ++pass) {
    const int j0 = pass == 0 ? 0 : nrb, j1 = pass == 0 ? nrb : nrb + 1;
    // The workers' tiles: a pass's first panel loads them from LDS (Ls = the raw diagonal block, Xs = I); from then on the D workers -- the waves
    // that used to compute D_{J+1} into Ls, with the same tiles and the same register layout -- compute it as the chain's first act, straight into
    // the accumulators, and the X workers write their piece of the identity into registers  (round 5: Ls / Xs were written, a barrier passed and
    // both read back)
    bool loaded = false;
    for (int J = j0; J < j1; ++J) {
        if (tid0 == 0) { CP_STAMP(0, J, 0); CP_CLK(19, J, 0); }
        {
            typename ChW<float>::acc_t acc[ChW<float>::NBLK][ChW<float>::NBLK];
            chol_chain<float, true>(sm.ch, acc, loaded, true, bad, [](int) {}, CH_NSP, ChNoTail{},
                [&](typename ChW<float>::acc_t (&ac)[ChW<float>::NBLK][ChW<float>::NBLK], const bool xside, const int) {
                    // (everything here hangs off a fenced copy of the thread index: shared with the chain's own address arithmetic it would stretch
                    //  live ranges through the chain -- sixteen registers of scratch traffic per step in the z wave's branch)
                    int tq = threadIdx.x;
                    asm volatile("" : "+v"(tq));
                    const int lane = tq & 63, wv = (tq >> 6) & 3, fa = (wv >> 1) & 1, fb = wv & 1;
                    if (xside) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) ac[0][0][e] = (fa == fb && acc_row(e, lane) == (lane & 31)) ? 1.f : 0.f;
                    } else if (wv != 1) {
                        // B2: D_{J+1} = A(J+1, J+1) - L(J+1, J) L(J+1, J)'   (the tile above the diagonal is never read)
                        const frag_t *LP = crit_lp(sm);
                        frag_t fB[4][3];
                        frags_lds(LP, fb, lane, fB);
                        float t2[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) t2[e] = T2[(32 * fa + acc_row(e, lane)) * NB + 32 * fb + (lane & 31)];
                        f32x16_t c2;
#pragma unroll
                        for (int e = 0; e < 16; ++e) c2[e] = 0.f;
                        mma6_alds(LP, fa, lane, fB, c2);
#pragma unroll
                        for (int e = 0; e < 16; ++e) ac[0][0][e] = t2[e] - c2[e];
                        if (threadIdx.x == 0) CP_STAMP(0, J - 1, 6);
                    }
                });
        }
        if (tid0 == 0) { CP_STAMP(0, J, 1); CP_CLK(19, J, 1); }
        // Ls = L_JJ, Xs = M_J
        // (the products' address arithmetic hangs off a value defined HERE: hoisted out of the panel loop it would live through
        //  the chain, whose code already takes every register, and come back as scratch traffic inside the chain)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if (J + 1 >= j1) break;
        __syncthreads();                                                            // b0: MPl complete, T1p / T2 landed
        if (tid0 == 0) CP_STAMP(0, J, 2);
        const int wave = tid >> 6, lane = tid & 63, fa = (wave >> 1) & 1, fb = wave & 1;
        frag_t *LP = crit_lp(sm);
        if (wave < 4) {
            // B1, transposed: L(J+1, J)'(a, i) = sum_c M_J(a, c) A(J+1, J)(i, c) -- rows a: the fb half of MPl (read k-step by k-step), columns i: the
            // fa half of T1p
            frag_t fB[4][3];
            frags_lds(sm.T1p, fa, lane, fB);
            f32x16_t c1;
#pragma unroll
            for (int e = 0; e < 16; ++e) c1[e] = 0.f;
            mma6_alds(sm.MPl, fb, lane, fB, c1);
            // The accumulator holds column i = lane & 31 at rows a = (e & 3) + 8 (e >> 2) + 4 (lane >> 5): v_permlane32_swap of register groups (0, 1) and
            // (2, 3) leaves every lane with eight consecutive a of its i -- exactly its granule of L(J+1, J)'s planes (k-steps 2 fb, 2 fb + 1; lanes >= 32:
            // the upper eight k), with no trip through LDS (round 5: Y -> barrier -> 512 granule jobs -> barrier used to stand here)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // (__float_as_uint of a copy: __builtin_bit_cast of ONE element of an ext-vector reads element 0)
                    const float lo = c1[8 * g2 + j], hi = c1[8 * g2 + 4 + j];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    x[j] = __uint_as_float(s0); x[4 + j] = __uint_as_float(s1);
                }
                u32x4_t p0, p1, p2;
                b3_split3(x, p0, p1, p2);
                const int gi = (2 * fb + g2) * 384 + fa * 64 + lane;
                LP[gi] = __builtin_bit_cast(frag_t, p0); LP[gi + 128] = __builtin_bit_cast(frag_t, p1); LP[gi + 256] = __builtin_bit_cast(frag_t, p2);
                // the f32 image is part of the final factor (k_gain; not read again in this launch): 32 bytes of row i
                float *sr = a.S + (size_t)((J + 1) * NB + 32 * fa + (lane & 31)) * (nrb * NB) + J * NB + 32 * fb + 16 * g2 + 8 * (lane >> 5);
                *reinterpret_cast<f4v_t *>(sr) = f4v_t{ x[0], x[1], x[2], x[3] };
                *reinterpret_cast<f4v_t *>(sr + 4) = f4v_t{ x[4], x[5], x[6], x[7] };
            }
            if (tid0 == 0) CP_STAMP(0, J, 7);
        }
        __syncthreads();                                                            // b2: LP = planes of L(J+1, J)
        if (tid0 == 0) CP_STAMP(0, J, 5);
        loaded = true;
        // (the second product is the next chain's first act -- worker_init above; no barrier in between: the chain's first step ends in one, and until
        //  then nothing another wave reads is written: crit_side's fetch starts behind that barrier)
        if (tid0 == 0) CP_STAMP(0, J, 3);
    }
    // the last panel of the LI update is factored: rescue stage (crit_tail: all twelve waves, crit_side calls it at the same barrier).  With rescued
    // landmarks it leaves Ls = D_hi, Xs = I and the second pass runs the chain of panel nrb
    if (pass == 1 || !a.tail || crit_tail(nrb) == 0) break;
    }
    if (bad && tid0 == 512) atomicExch(a.status, 1);
}

// waves 10, 11: M_J and L(J+1, J) leave through wave 10; both fetch the tiles of row J+1 while the chain of panel J runs
__device__ __forceinline__ void crit_side(const CpArgs &a, int nrb, CritSmem &sm)
{
    auto &Ls = sm.ch.Ls; auto &Xs = sm.ch.Xs;
    float *T2 = &sm.ch.Bs[0][0];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lds = nrb * NB;
    const __amdgpu_buffer_rsrc_t rSp = cp_rsrc(a.Sp);
    int32_t *guard = a.status + 1;
    // Fetch of the two tiles of row J+1 (published by its row workgroup: A(J+1, J+1) in f32, then A(J+1, J) as planes; flag = base + 2 once
    // both are out) by LDS-DMA: no registers are held while the bytes travel.  State: 0 poll -> 1 test, issue the DMAs -> 2 wait for them
    // -> 5 done.  Whatever is issued in one pipeline step of the chain is consumed `cool` steps later (an sc1 access takes 1-1.4 us, a step
    // 0.6), and the loop's barriers are raw s_barriers: the chain's barriers are never held for memory.  (Each of these lines is read once
    // per launch by this CU, so there is no older copy for its L1 to hold; the loads carry sc1 all the same.)
    int fst = 5, cool = 0;
    unsigned pv = 0;
    const long long poll_budget = a.poll_budget;               // shader clocks per chain step spent polling row J+1's flag (0: one poll per step)
    for (int pass = 0; pass < 2; ++pass) {                         // (as crit_main)
    const int j0 = pass == 0 ? 0 : nrb, j1 = pass == 0 ? nrb : nrb + 1;
    for (int J = j0; J < j1; ++J) {
        const bool more = J + 1 < j1;
        const int fr = J + 1;                                     // the row whose tiles this panel's products need
        const unsigned *fflag = cf_rowA(a.cf, fr < 64 ? fr : 63);
        fst = (more && J > 0) ? 0 : 5;                            // (panel 0's tiles came with the prologue)
        cool = 0;
        // The flag is polled at every step of the chain without ever waiting on memory: a 4-byte LDS-DMA of the flag word is sent off, and
        // what the earlier ones have brought is read from LDS (a word that has not landed yet only reads as "not yet").  No poll of the
        // previous panel is still in flight here: its tile fetch ended in vmcnt(0).
        if (wave == 11 && fst == 0) sm.fpoll[lane] = a.base;
        auto fetch_step = [&](bool blocking) {
            if (!blocking && cool > 0) { --cool; return; }
            if (fst == 0) {
                pv = *reinterpret_cast<volatile unsigned *>(&sm.fpoll[lane]);
                if (!cf_reached(pv, a.base + 2)) {
                    __builtin_amdgcn_global_load_lds(fflag, (__attribute__((address_space(3))) void *)sm.fpoll, 4, 0, 16);
                    if (blocking) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    const frag_t *src1 = static_cast<const frag_t *>(a.Tp) + (size_t)fr * B3_SGRAN + lane;
#pragma unroll
                    for (int t = 0; t < 24; ++t)
                        __builtin_amdgcn_global_load_lds(src1 + t * 64, (__attribute__((address_space(3))) void *)(sm.T1p + t * 64), 16, 0, 16);
                    // A(fr, fr): 1024 granules of 4 floats; granule g = t * 64 + lane -> row g >> 4, columns 4 (g & 15); LDS image [64][64]
                    const float *src2 = a.S + (size_t)(fr * NB + (lane >> 4)) * lds + fr * NB + (lane & 15) * 4;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        __builtin_amdgcn_global_load_lds(src2 + (size_t)(4 * t) * lds, (__attribute__((address_space(3))) void *)(T2 + (t * 64) * 4), 16, 0, 16);
                    fst = 2; cool = 2;
                    CP_STAMP(1, J, 0);
                }
            } else if (fst == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                fst = 5;
                CP_STAMP(1, J, 3);
            }
        };
        // rows 8 sp .. 8 sp + 7 of M_J (final once the z wave has passed them) -> planes in LDS and in Sp(J, J)
        u32x4_t mp0, mp1, mp2; unsigned mgb = 0;                                    // the last rows' planes on their way out (publish_m_rows(7, false))
        auto publish_m_rows = [&](int sp, bool store = true) {
            const int arow = 8 * sp + (lane >> 3), cg = lane & 7;                   // 8 consecutive c of one row per lane
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = Xs[arow][8 * cg + j];
            u32x4_t p0, p1, p2;
            b3_split3(x, p0, p1, p2);
            const int q = cg >> 1, h = cg & 1, half = arow >> 5, r = arow & 31;
            const int gi = q * 384 + half * 64 + 32 * h + r;
            sm.MPl[gi] = __builtin_bit_cast(frag_t, p0); sm.MPl[gi + 128] = __builtin_bit_cast(frag_t, p1); sm.MPl[gi + 256] = __builtin_bit_cast(frag_t, p2);
            const unsigned gb = ((unsigned)(J * a.sp_stride + J) * B3_SGRAN + gi) * 16u;
            if (store) { st16_sc1(p0, rSp, gb); st16_sc1(p1, rSp, gb + 128 * 16); st16_sc1(p2, rSp, gb + 256 * 16); }
            else { mp0 = p0; mp1 = p1; mp2 = p2; mgb = gb; }
        };
        // columns 8 sp .. 8 sp + 7 of L_JJ (final once the factor wave has passed them; zero above the diagonal) to S -- part of the final
        // factor (k_gain; not read again in this launch): wave 11, lane = row
        auto publish_l_cols = [&](int sp) {
            const int i = lane, C = 8 * sp;
            if (J >= nrb) return;                                                   // (the HI panel's factor is not part of S)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c4 = C + 4 * u;
                f4v_t w = *reinterpret_cast<const f4v_t *>(&Ls[i][c4]);
                w.x = c4 <= i ? w.x : 0.f; w.y = c4 + 1 <= i ? w.y : 0.f; w.z = c4 + 2 <= i ? w.z : 0.f; w.w = c4 + 3 <= i ? w.w : 0.f;
                *reinterpret_cast<f4v_t *>(a.S + (size_t)(J * NB + i) * lds + J * NB + c4) = w;
            }
        };
#pragma unroll 1
        for (int k = -1; k <= CH_NSP; ++k) {                     // one barrier per pipeline step of the chain
            // the fetch first: it is the only thing here that waits on memory (the compiler's wait covers every older operation of the
            // wave, so this step's stores must come after it -- a store's write-through acknowledge takes about as long as a step)
            // (round 5: while the flag is awaited the step's time is spent polling -- a poll every ~0.1 us instead of one per step: the word is seen
            //  0.5 us earlier on average, and the tiles' DMAs leave mid-step.  The budget keeps the wave in front of the step's barrier.)
            if (wave == 11 && k >= 0) {                            // (k = -1: the second product of the panel before may still be reading T2)
                if (fst == 0 && poll_budget > 0 && k >= a.poll_from) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                    do {
                        fetch_step(false);
                        if (fst != 0) break;
                        __builtin_amdgcn_s_sleep(3);
                    } while ((long long)(__builtin_amdgcn_s_memtime() - t0) < poll_budget);
                } else fetch_step(false);
            }
            if (wave == 10) {
                if (k == -1 && J > 0 && J < nrb) {      // L(J, J-1)'s planes (stored during the last products) have drained: publish
                    drain_stores();
                    if (lane == 0) cf_store(cf_rowL(a.cf, J), a.base + (unsigned)J);
                    CP_STAMP(1, J, 5);
                }
                if (k >= 1) publish_l_cols(k - 1);
                if (k >= 2) publish_m_rows(k - 2);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this step's LDS writes have landed
            __builtin_amdgcn_s_barrier();                        // (raw: __syncthreads() would also wait for every LDS-DMA in flight)
        }
        // chain done: Ls = L_JJ, Xs = M_J (rows 56..63 not yet in planes)
        // (the products only need the last rows' planes in LDS: their copies for the other workgroups leave behind b0, in the products' shadow)
        if (wave == 10) {
            publish_m_rows(7, !more);
            if (!more) { drain_stores(); if (lane == 0) cf_store(a.cf + CF_MP, a.base + (unsigned)J + 1); CP_STAMP(1, J, 4); }
        }
        if (!more) break;
        {   // the fetch must be complete before the products (normally it is: the tiles arrive mid-chain)
            int spin = 0;
            while (wave == 11 && fst < 5 && spin < SPIN_LIMIT) {
                const int before = fst;
                fetch_step(true);
                if (fst == before) {
                    __builtin_amdgcn_s_sleep(1); ++spin;
                    if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                }
            }
            if (wave == 11 && fst < 5 && lane == 0) atomicExch(guard, 1);
        }
        CP_STAMP(wave == 10 ? 1 : 18, J, wave == 10 ? 7 : 2);                         // arrival at b0
        __syncthreads();                                                            // b0
        // M_J is out once wave 10's stores have drained -- in the shadow of the first product, not in front of it (the tiles of row J+1
        // arrive before the chain ends, so b0 waits for nothing else)
        if (wave == 10) {
            st16_sc1(mp0, rSp, mgb); st16_sc1(mp1, rSp, mgb + 128 * 16); st16_sc1(mp2, rSp, mgb + 256 * 16);
            drain_stores(); if (lane == 0) cf_store(a.cf + CF_MP, a.base + (unsigned)J + 1); CP_STAMP(1, J, 4);
        }
        __syncthreads();                                                            // b2: LP = planes of L(J+1, J), Y = its f32 image
        if (wave == 10) {
            // L(J+1, J) leaves as planes: 24 granules per lane in three batches (the LDS reads of a batch first); the flag follows at the first
            // step of the next chain, once these stores have drained (the rows need it only after their own L(i, J))
#pragma unroll
            for (int b8 = 0; b8 < 3; ++b8) {
                frag_t g8[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) g8[t] = crit_lp(sm)[(b8 * 8 + t) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    st16_sc1(__builtin_bit_cast(u32x4_t, g8[t]), rSp, ((unsigned)((J + 1) * a.sp_stride + J) * B3_SGRAN + (b8 * 8 + t) * 64 + lane) * 16u);
            }
            CP_STAMP(1, J, 6);
        }
    }
    if (pass == 1 || !a.tail || crit_tail(nrb) == 0) break;
    }
}

// The D workers' 16 x 16 blocks of the flag-driven chain in crit (pre3_chain_async.h, DInit).  First panel of a pass: the raw block is in Ls.  Otherwise
// B2, D_{J+1} = A(J+1, J+1) - L(J+1, J) L(J+1, J)', block by block on v_mfma_f32_16x16x32_bf16 from the planes the first product left in LDS (LP: [k-step of
// 16][plane][32-row half][lane] granules, lane = row & 31 + 32 (upper eight k)) and the f32 tile the fetcher brought (T2): six plane products per f32
// product in k_downdate_b3's order, accumulated from zero and subtracted once.  The result lands in the layout the chain's D workers keep (row 4 (lane >> 4)
// + reg, column lane & 15): no trip through LDS, and the block column that holds sub-panels 0 and 1 is done -- and published -- first.
struct CritDInit {
    static constexpr bool from_ls = false;
    CritSmem *sm; bool loaded;
    typedef float f4acc_t __attribute__((ext_vector_type(4)));
    // the two 16 x 16 blocks (rows r0 .. r0+15 and r0+16 .. r0+31, columns c0 .. c0+15) of one block column; a block above the diagonal (live = false) is zero
    __device__ __forceinline__ void operator()(f4acc_t &acc0, f4acc_t &acc1, const bool live0, const bool live1, const int r0, const int c0, const int lane_in) const
    {
        // (the address arithmetic hangs off a fenced copy of the lane id: hoisted out of the panel loop it lives through the chain, whose code takes every
        //  register, and comes back as scratch reloads -- with a vmcnt(0) each -- on the D workers' way to their first publication)
        int lane = lane_in;
        asm volatile("" : "+v"(lane));
        const int g = lane >> 4, cl = lane & 15;
        if (!loaded) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc0[e] = live0 ? sm->ch.Ls[r0 + 4 * g + e][c0 + cl] : 0.f; acc1[e] = live1 ? sm->ch.Ls[r0 + 16 + 4 * g + e][c0 + cl] : 0.f; }
            return;
        }
        const frag_t *LP = crit_lp(*sm);
        const float *T2 = &sm->ch.Bs[0][0];
        const int ra0 = r0 + cl, ra1 = r0 + 16 + cl, rb = c0 + cl;
        float t20[4], t21[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { t20[e] = T2[(r0 + 4 * g + e) * NB + c0 + cl]; t21[e] = T2[(r0 + 16 + 4 * g + e) * NB + c0 + cl]; }
        f4acc_t c0v = { 0.f, 0.f, 0.f, 0.f }, c1v = c0v;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int qq = 2 * kk + (g >> 1), kh = g & 1;
            const frag_t *pa0 = LP + qq * 384 + (ra0 >> 5) * 64 + (ra0 & 31) + 32 * kh, *pa1 = LP + qq * 384 + (ra1 >> 5) * 64 + (ra1 & 31) + 32 * kh;
            const frag_t *pb = LP + qq * 384 + (rb >> 5) * 64 + (rb & 31) + 32 * kh;
            const frag_t b0 = pb[0], b1 = pb[128], b2 = pb[256];
            const frag_t x0 = pa0[0], x1 = pa0[128], x2 = pa0[256], y0 = pa1[0], y1 = pa1[128], y2 = pa1[256];
            // (two independent accumulate chains, interleaved: a dependent 16 x 16 x 32 product waits for its predecessor's result)
#define CD_MMA(c, x, y) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, x), __builtin_bit_cast(bf16x8_t, y), c, 0, 0, 0)
            if (live0) { CD_MMA(c0v, x0, b0); } CD_MMA(c1v, y0, b0);
            if (live0) { CD_MMA(c0v, x0, b1); } CD_MMA(c1v, y0, b1);
            if (live0) { CD_MMA(c0v, x1, b0); } CD_MMA(c1v, y1, b0);
            if (live0) { CD_MMA(c0v, x1, b1); } CD_MMA(c1v, y1, b1);
            if (live0) { CD_MMA(c0v, x0, b2); } CD_MMA(c1v, y0, b2);
            if (live0) { CD_MMA(c0v, x2, b0); } CD_MMA(c1v, y2, b0);
#undef CD_MMA
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc0[e] = live0 ? t20[e] - c0v[e] : 0.f; acc1[e] = t21[e] - c1v[e]; }
    }
};

#ifndef PRE3_CRIT_ASYNC
#define PRE3_CRIT_ASYNC 1            // crit runs the flag-driven chain (pre3_chain_async.h); 0: the lock-step chain of rounds 2-5 (crit_main / crit_side)
#endif
// Round 6: crit on the flag-driven chain.  All twelve waves run ONE panel loop; per panel: [products of the previous panel: D_{J+1} into Ls, X := I]
// -> barrier -> chain (waves 0-9) beside the publisher (wave 10: columns of L_JJ to S and rows of M_J to planes as the factor / z wave finish
// them) and the fetcher (wave 11: row J+1's two tiles by LDS-DMA; it may now block on memory -- no barrier waits for it until the chain is over)
// -> barrier (b0) -> first product L(J+1, J) = A M_J' (waves 0-3), M_J's flag (wave 10) -> barrier (b2) -> L(J+1, J)'s planes leave (wave 10).
__device__ __forceinline__ void crit_loop_async(const CpArgs &a, int nrb, int rows, CritSmem &sm)
{
    auto &Ls = sm.ch.Ls; auto &Xs = sm.ch.Xs;
    float *T2 = &sm.ch.Bs[0][0];                                 // A(J+1, J+1), f32 [64][64]
    const int tid0 = threadIdx.x;
    const int lds = nrb * NB;
    const __amdgpu_buffer_rsrc_t rSp = cp_rsrc(a.Sp);
    int32_t *guard = a.status + 1;
    bool bad = false;
    for (int pass = 0; pass < 2; ++pass) {
    const int j0 = pass == 0 ? 0 : nrb, j1 = pass == 0 ? nrb : nrb + 1;
    bool loaded = false;                                         // false: the pass's first panel, Ls / Xs hold the block and the identity
    for (int J = j0; J < j1; ++J) {
        const bool more = J + 1 < j1;
        // the update's last panel: the sub-panels behind its last real row are padding (identity columns of S against zero rows of [HP | nu]); the
        // flag-driven chain skips them at no cost to the others (the lock-step chain's step-skipping form cost every panel 6 %: rounds 3-5 ran all ten
        // steps here), and the publishers send the identity in their place
        const int nsp_eff = (pass == 0 && J == nrb - 1 && a.crit_early) ? __builtin_amdgcn_readfirstlane((rows - J * NB + CH_MB - 1) / CH_MB) : CH_NSP;
        if (tid0 == 0) { CP_STAMP(0, J, 0); CP_CLK(19, J, 0); }
        // rows 8 sp .. 8 sp + 7 of M_J (final once the z wave has passed them) -> planes in LDS (the first product's operand) and in Sp(J, J) (the rows'
        // and strips'): one wave, 8 consecutive c of one row per lane
        auto publish_m = [&](unsigned *fl, const int sp) {
            int tq = threadIdx.x;
            asm volatile("" : "+v"(tq));
            const int lane = tq & 63;
            if (!cha_wait(fl + CHF_Z, (unsigned)(sp + 1))) bad = true;
            const int arow = 8 * sp + (lane >> 3), cg = lane & 7;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = sp < nsp_eff ? Xs[arow][8 * cg + j] : (arow == 8 * cg + j ? 1.f : 0.f);
            u32x4_t p0, p1, p2;
            b3_split3(x, p0, p1, p2);
            const int q = cg >> 1, h = cg & 1, half = arow >> 5, r = arow & 31;
            const int gi = q * 384 + half * 64 + 32 * h + r;
            sm.MPl[gi] = __builtin_bit_cast(frag_t, p0); sm.MPl[gi + 128] = __builtin_bit_cast(frag_t, p1); sm.MPl[gi + 256] = __builtin_bit_cast(frag_t, p2);
            const unsigned gb = ((unsigned)(J * a.sp_stride + J) * B3_SGRAN + gi) * 16u;
            st16_sc1(p0, rSp, gb); st16_sc1(p1, rSp, gb + 128 * 16); st16_sc1(p2, rSp, gb + 256 * 16);
        };
        {
            typename ChW<float>::acc_t acc[ChW<float>::NBLK][ChW<float>::NBLK];
            chol_chain_async<float, true, true>(sm.ch, acc, loaded, true, bad,
                [&](unsigned *fl) {
                    int tq = threadIdx.x;
                    asm volatile("" : "+v"(tq));
                    const int wave = __builtin_amdgcn_readfirstlane(tq >> 6), lane = tq & 63;
                    if (wave == 1) {
                        // columns 8 sp .. 8 sp + 7 of L_JJ (zero above the diagonal) to S as the factor wave finishes them: part of the final factor (k_gain;
                        // not read again in this launch).  (The HI panel's factor is not part of S.)
                        if (J < nrb) {
                            __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
                            for (int sp = 0; sp < CH_NSP; ++sp) {
                                if (!cha_wait(fl + CHF_F, (unsigned)(sp + 1))) bad = true;
                                const int i = lane, C = 8 * sp;
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const int c4 = C + 4 * u;
                                    f4v_t w = *reinterpret_cast<const f4v_t *>(&Ls[i][c4]);
                                    if (sp >= nsp_eff) w = f4v_t{ c4 == i ? 1.f : 0.f, c4 + 1 == i ? 1.f : 0.f, c4 + 2 == i ? 1.f : 0.f, c4 + 3 == i ? 1.f : 0.f };
                                    w.x = c4 <= i ? w.x : 0.f; w.y = c4 + 1 <= i ? w.y : 0.f; w.z = c4 + 2 <= i ? w.z : 0.f; w.w = c4 + 3 <= i ? w.w : 0.f;
                                    *reinterpret_cast<f4v_t *>(a.S + (size_t)(J * NB + i) * lds + J * NB + c4) = w;
                                }
                                if ((sp & 1) && sp < CH_NSP - 1) publish_m(fl, sp);     // (rows of M_J: sub-panels 1, 3, 5 here, the others on wave 10)
                            }
                            __builtin_amdgcn_s_setprio(0);
                        } else {
                            for (int sp = 1; sp < CH_NSP - 1; sp += 2) publish_m(fl, sp);
                        }
                        // this wave's planes of M_J must be out before wave 10 raises M_J's flag (behind the chain's last barrier): they left at least
                        // two sub-panels ago, the wait is normally over when it starts
                        drain_stores();
                    } else if (wave == 10) {
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
                        for (int sp = 0; sp < CH_NSP; sp += 2) publish_m(fl, sp);
                        publish_m(fl, CH_NSP - 1);
                        __builtin_amdgcn_s_setprio(0);
                        // M_J is out once these stores have drained (the flag goes up behind the chain's last barrier, in the shadow of the first product)
                    } else if (wave == 11) {
                        // L(J, J-1) leaves as planes (the first product of the panel before left them in LP: 24 granules per lane in three batches, the LDS
                        // reads of a batch first) -- here, behind the chain's first barrier, which then does not wait for this wave's store instructions --;
                        // once they have drained: row J's flag (the rows need it only after their own L(i, J-1))
                        if (J > 0 && J < nrb) {
                            const frag_t *LPs = crit_lp(sm);
#pragma unroll
                            for (int b8 = 0; b8 < 3; ++b8) {
                                frag_t g8[8];
#pragma unroll
                                for (int t = 0; t < 8; ++t) g8[t] = LPs[(b8 * 8 + t) * 64 + lane];
#pragma unroll
                                for (int t = 0; t < 8; ++t)
                                    st16_sc1(__builtin_bit_cast(u32x4_t, g8[t]), rSp, ((unsigned)(J * a.sp_stride + J - 1) * B3_SGRAN + (b8 * 8 + t) * 64 + lane) * 16u);
                            }
                            CP_STAMP(1, J - 1, 6);
                            drain_stores();
                            if (lane == 0) cf_store(cf_rowL(a.cf, J), a.base + (unsigned)J);
                            CP_STAMP(1, J, 5);
                        }
                        if (more && J > 0) {
                        // row J+1's two tiles (published by its row workgroup: flag = base + 2 once both are out).  (panel 0's came with the prologue)
                        const int fr = J + 1;
                        const unsigned *fflag = cf_rowA(a.cf, fr < 64 ? fr : 63);
                        bool gave_up = true;
                        for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
                            if (cf_reached(cf_load(fflag), a.base + 2)) { gave_up = false; break; }
                            if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gave_up = false; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (gave_up && lane == 0) atomicExch(guard, 1);
                        // (the D workers multiply D_{J+1} out of T2 behind the chain's first barrier: every one of them has its tile before T2 is overwritten)
                        if (!cha_wait(fl + CHF_DI0 + 0, 1u) || !cha_wait(fl + CHF_DI0 + 2, 1u) || !cha_wait(fl + CHF_DI0 + 3, 1u)) bad = true;
                        CP_STAMP(1, J, 0);
                        const frag_t *src1 = static_cast<const frag_t *>(a.Tp) + (size_t)fr * B3_SGRAN + lane;
#pragma unroll
                        for (int t = 0; t < 24; ++t)
                            __builtin_amdgcn_global_load_lds(src1 + t * 64, (__attribute__((address_space(3))) void *)(sm.T1p + t * 64), 16, 0, 16);
                        // A(fr, fr): 1024 granules of 4 floats; granule g = t * 64 + lane -> row g >> 4, columns 4 (g & 15); LDS image [64][64]
                        const float *src2 = a.S + (size_t)(fr * NB + (lane >> 4)) * lds + fr * NB + (lane & 15) * 4;
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            __builtin_amdgcn_global_load_lds(src2 + (size_t)(4 * t) * lds, (__attribute__((address_space(3))) void *)(T2 + (t * 64) * 4), 16, 0, 16);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        CP_STAMP(1, J, 3);
                        }
                    }
                },
                [&](typename ChW<float>::acc_t (&ac)[ChW<float>::NBLK][ChW<float>::NBLK], const bool xside, const int) {
                    int tq = threadIdx.x;
                    asm volatile("" : "+v"(tq));
                    const int lane = tq & 63, wv = (tq >> 6) & 3, fa = (wv >> 1) & 1, fb = wv & 1;
                    if (xside) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) ac[0][0][e] = (fa == fb && acc_row(e, lane) == (lane & 31)) ? 1.f : 0.f;
                    }
                },
                CritDInit{ &sm, loaded }, nsp_eff);
        }
        // (the chain's last barrier = b0: Ls = L_JJ, Xs = M_J, MPl complete, T1p / T2 landed)
        if (tid0 == 0) { CP_STAMP(0, J, 1); CP_CLK(19, J, 1); CP_STAMP(0, J, 2); }
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int wave = tid >> 6, lane = tid & 63, fa = (wave >> 1) & 1, fb = wave & 1;
        if (wave == 10) {
            // M_J's last planes have left with the publisher's loop: drained -> the flag, in the shadow of the first product
            drain_stores();
            if (lane == 0) cf_store(a.cf + CF_MP, a.base + (unsigned)J + 1);
            CP_STAMP(1, J, 4);
        }
        if (!more) break;
        frag_t *LP = crit_lp(sm);
        if (wave < 4) {
            // B1, transposed (as crit_main): L(J+1, J)'(a, i) = sum_c M_J(a, c) A(J+1, J)(i, c)
            frag_t fB[4][3];
            frags_lds(sm.T1p, fa, lane, fB);
            f32x16_t c1;
#pragma unroll
            for (int e = 0; e < 16; ++e) c1[e] = 0.f;
            mma6_alds(sm.MPl, fb, lane, fB, c1);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = c1[8 * g2 + j], hi = c1[8 * g2 + 4 + j];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    x[j] = __uint_as_float(s0); x[4 + j] = __uint_as_float(s1);
                }
                u32x4_t p0, p1, p2;
                b3_split3(x, p0, p1, p2);
                const int gi = (2 * fb + g2) * 384 + fa * 64 + lane;
                LP[gi] = __builtin_bit_cast(frag_t, p0); LP[gi + 128] = __builtin_bit_cast(frag_t, p1); LP[gi + 256] = __builtin_bit_cast(frag_t, p2);
                float *sr = a.S + (size_t)((J + 1) * NB + 32 * fa + (lane & 31)) * (nrb * NB) + J * NB + 32 * fb + 16 * g2 + 8 * (lane >> 5);
                *reinterpret_cast<f4v_t *>(sr) = f4v_t{ x[0], x[1], x[2], x[3] };
                *reinterpret_cast<f4v_t *>(sr + 4) = f4v_t{ x[4], x[5], x[6], x[7] };
            }
            if (tid0 == 0) CP_STAMP(0, J, 7);
        }
        // (no barrier of its own behind the first product: the next chain's first barrier orders LP for the D workers and the publisher)
        if (tid0 == 0) { CP_STAMP(0, J, 5); CP_STAMP(0, J, 3); }
        loaded = true;
    }
    if (pass == 1 || !a.tail || crit_tail(nrb) == 0) break;
    }
    if (bad) atomicExch(a.status, 1);
}

// what crit keeps behind CritSmem (and a strip behind its plane slots) of the rescue stage's outcome
struct TailSmem {
    int cnt, r_hi, pad[2];
    int lm[HT_MAXL], off[HT_MAXL], d[HT_MAXL];
    __attribute__((aligned(16))) float rv[NB][16];               // row a of (H J) for the listed landmarks: 7 pose + 6 landmark entries, [13] = nu
    __attribute__((aligned(16))) float tpose[NB][8];             // crit: (H J P)(a, 0..6)
};
constexpr size_t CP_TAIL_OFF = (sizeof(CritSmem) + 127) / 128 * 128;
constexpr size_t CP_PART_OFF = (CP_TAIL_OFF + sizeof(TailSmem) + 127) / 128 * 128;      // six [32][33] f32 tiles: shares of Y_hi Y_hi' (crit_tail)
static_assert(CP_PART_OFF + 6 * 32 * 33 * sizeof(float) <= 160 * 1024 - 1024, "crit's LDS with the tail");
static_assert(offsetof(CritSmem, T1p) == offsetof(CritSmem, MPl) + sizeof(frag_t) * B3_SGRAN, "crit_tail keeps (H J P) at the landmark columns in MPl | T1p");

__device__ __forceinline__ void crit_body(const CpArgs &a, int nrb, int rows, unsigned char *smem_raw)
{
    CritSmem &sm = *reinterpret_cast<CritSmem *>(smem_raw);
    crit_prologue(a, nrb, sm);
#if PRE3_CRIT_ASYNC
    crit_loop_async(a, nrb, rows, sm);
#else
    if (threadIdx.x < 640) crit_main(a, nrb, sm);
    else crit_side(a, nrb, sm);
#endif
    if (a.tail) {
        const CpTail t = args_from_lds<CpTail>(CP_T_OFF);
        // the rescue flags in measurement order (rescue_hi_inliers.m:44-46 as pre3_get_flags reports it): off the rescue stage's path
        for (int j = threadIdx.x; j < t.m; j += CP_NTH) {
            const int i = t.meas[j];
            t.hi_meas[j] = (t.lm_ic[i] == 1 && t.lm_li[i] == 0) ? ld_i32_sc1(t.lm_hi + i) : 0;
        }
        // the rescue stage's count reaches the host with the device's error words, as k_collect_hi / k_hi_fused publish them -- behind the HI panel's
        // chain, so that a factorisation that failed HERE fails the call that reads this count
        __syncthreads();
        if (threadIdx.x == 0) {
            const TailSmem &ts = *reinterpret_cast<const TailSmem *>(smem_raw + CP_TAIL_OFF);
            t.mail[5] = ts.cnt;
            t.mail[6] = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t.mail[7] = __hip_atomic_load(a.status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(&t.mail[9], t.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// crit between the LI update's last panel and the HI panel (all twelve waves; same barrier count on every path)
// ------------------------------------------------------------------------------------------------------------------------------
// one 16-byte granule of y's planes: the lane's row (landmark, image row, k half) in the VGPR part, plane / block / k-step in the SGPR part
__device__ __forceinline__ unsigned yp_voff(int lm, int c, int h, int ykcap) { return (unsigned)(((lm * 2 + c) * 3) * ykcap + 8 * h) * 2u; }
__device__ __forceinline__ unsigned yp_soff(int pl, int K, int q, int ykcap) { return (unsigned)(pl * ykcap + 64 * K + 16 * q) * 2u; }

// rescue_hi_inliers.m:44-46 from the strips' gate flags; D_hi = (H J) P (H J)' + I - Y_hi Y_hi' -> Ls, I -> Xs.  Returns the HI update's rows in
// this launch (0: nothing rescued, or more than HT_MAXL landmarks: the host's general path takes that update).
__device__ __attribute__((noinline)) int crit_tail(int nrb_v)
{
    const CpArgs a = args_from_lds<CpArgs>(CP_TA_OFF);
    const CpTail t = args_from_lds<CpTail>(CP_T_OFF);
    const int nrb = __builtin_amdgcn_readfirstlane(nrb_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    CritSmem &sm = *reinterpret_cast<CritSmem *>(cp_smem);
    TailSmem &ts = *reinterpret_cast<TailSmem *>(cp_smem + CP_TAIL_OFF);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int32_t *guard = a.status + 1;
    if (tid == 0) CP_STAMP(22, 0, 0);
    // (C1) every strip has projected and gated its landmarks
    if (wave < 2) {
        bool gave_up = true;
        for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
            bool ok = true;
            for (int s0 = wave * 64; s0 < a.n_strips; s0 += 128) {
                const int sx = s0 + lane;
                if (sx < a.n_strips) ok = ok && cf_reached(cf_load(cf_strip(a.cf, sx)), a.base + CFV_GATE);
            }
            if (__all(ok)) { gave_up = false; break; }
            if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gave_up = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (gave_up && lane == 0) atomicExch(guard, 1);
    }
    __syncthreads();
    if (tid == 0) CP_STAMP(22, 0, 1);
    // (C2) the HI list in measurement order (rescue_hi_inliers.m:44-46) from the candidates' flags: the selection stage listed the candidates in that
    //      order (sel_rows[m ..]: measurement << 16 | landmark), so one load level separates the list from the flags
    const int n_all = t.m - *a.n_dev;
    unsigned char *flg = reinterpret_cast<unsigned char *>(&sm.ch.Bs[0][0]);       // [n_all] (m <= 16 K: the host checks)
    for (int r = tid; r < n_all; r += CP_NTH) flg[r] = (unsigned char)(ld_i32_sc1(t.lm_hi + (t.sel_rows[t.m + r] & 0xffff)) != 0);
    __syncthreads();
    if (tid < 64) {
        int cnt = 0;
        for (int b0 = 0; b0 < n_all; b0 += 64) {
            const int r = b0 + tid;
            const int in = r < n_all ? flg[r] : 0;
            const unsigned long long mask = __ballot(in);
            const int pos = cnt + __popcll(mask & ((1ull << tid) - 1ull));
            if (in) {
                const int v = t.sel_rows[t.m + r];
                if (pos < HT_MAXL) ts.lm[pos] = v & 0xffff;
                t.sel_rows[pos] = v >> 16;                                          // (pos <= r < m: the candidates' own entries lie behind the first m)
            }
            cnt += __popcll(mask);
        }
        if (tid == 0) { ts.cnt = cnt; ts.r_hi = (cnt >= 1 && cnt <= HT_MAXL) ? 2 * cnt : 0; }
    }
    __syncthreads();
    const int cnt = ts.cnt, r_hi = ts.r_hi;
    if (tid == 0) { t.stats[5] = cnt; t.stats[8] = 0; CP_STAMP(22, 0, 2); }
    // (C3) the listed landmarks' rows (H J, nu: written by the gate) -> LDS and, with the list, to the strips
    const __amdgpu_buffer_rsrc_t rHb = cp_rsrc(t.Hb), rHib = cp_rsrc(t.hib);
    if (tid < NB * 4) {
        const int row = tid >> 2, g4 = tid & 3;
        u32x4_t v = { 0u, 0u, 0u, 0u };
        if (row < r_hi) v = ld16_sc1(rHb, (unsigned)((ts.lm[row >> 1] * 2 + (row & 1)) * 16 + 4 * g4) * 4u);
        *reinterpret_cast<u32x4_t *>(&ts.rv[row][4 * g4]) = v;
        st16_sc1(v, rHib, (unsigned)(64 + row * 16 + 4 * g4) * 4u);
    } else if (tid < NB * 4 + HT_MAXL) {
        const int l = tid - NB * 4;
        int lm = 0, off = 0, d = 0;
        if (2 * l < r_hi) { lm = ts.lm[l]; off = t.lm_off[lm]; d = t.lm_type[lm] == PRE3_INVDEPTH ? 6 : 3; }
        ts.off[l] = off; ts.d[l] = d;
        st_i32_sc1(t.hib + 4 + l, lm);
    } else if (tid == NB * 4 + HT_MAXL) {
        st_i32_sc1(t.hib + 0, cnt); st_i32_sc1(t.hib + 1, r_hi);
        st_i32_sc1(reinterpret_cast<int32_t *>(a.cf + CF_HI + 1), r_hi);          // (the consumers read it from the flag's own line)
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) { cf_store(a.cf + CF_HI, a.base + CFV_HIL); CP_STAMP(22, 0, 3); }
    if (r_hi == 0) return 0;
    // (D0) Y_hi Y_hi' on the matrix cores: wave -> (quadrant of the lower triangle, every n-th block of k); with at most 32 rows only the first
    //      quadrant holds anything and all twelve waves share its blocks.  The first block's planes are requested before anything else: they
    //      travel while (D1) gathers.  (One CU takes in ~70 GB/s of handed-off bytes: the planes, 72 KB per block over three quadrants, are what this costs.)
    const bool one_quad = r_hi <= 32;
    const int quad = one_quad ? 0 : wave % 3, kg = one_quad ? wave : wave / 3, kstep = one_quad ? 12 : 4, fa = quad >= 1 ? 1 : 0, fb = quad == 2 ? 1 : 0;
    const __amdgpu_buffer_rsrc_t rY = cp_rsrc(t.Yp);
    unsigned va, vb;
    {
        const int ra = 32 * fa + (lane & 31), rb2 = 32 * fb + (lane & 31), h = lane >> 5;
        va = yp_voff(ra < r_hi ? ts.lm[ra >> 1] : t.N, ra & 1, h, t.ykcap); vb = yp_voff(rb2 < r_hi ? ts.lm[rb2 >> 1] : t.N, rb2 & 1, h, t.ykcap);
    }
    frag_t fA[4][3], fB[4][3];
    auto load_y = [&](int K) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                fA[q][pl] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rY, va, yp_soff(pl, K, q, t.ykcap), 16));
                fB[q][pl] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rY, vb, yp_soff(pl, K, q, t.ykcap), 16));
            }
    };
    if (kg < nrb) load_y(kg);
    // (D1) T = (H J P) at the columns the listed rows touch: pose columns -> ts.tpose, landmark l's six -> tlm[a][6 l ..].  P is symmetric to the bit,
    //      so T[a][k] = sum_s rv[a][s] P[ucol k][col(a, s)]: one row of P per job, its 13 entries in four loads, both image rows of a landmark per job
    float *tlm = reinterpret_cast<float *>(sm.MPl);                                // [64][192] = MPl | T1p (free until the chain publishes M)
    {
        const __amdgpu_buffer_rsrc_t rP = cp_rsrc(a.P);
        const int nU = 7 + 6 * cnt, njobs = cnt * nU;
        for (int job = tid; job < njobs; job += CP_NTH) {
            const int la = job / nU, k = job - la * nU;
            const int lk = k < 7 ? 0 : (k - 7) / 6, tk = k < 7 ? k : (k - 7) - 6 * lk;
            const int ucol = k < 7 ? k : (tk < ts.d[lk] ? ts.off[lk] + tk : 0);
            const unsigned rb = (unsigned)ucol * (unsigned)a.ld * 4u, ob = (unsigned)ts.off[la] * 4u;
            const f4v_t p0 = ld4f(rP, rb), p1 = ld4f(rP, rb + 16u), p2 = ld4f(rP, rb + ob), p3 = ld4f(rP, rb + ob + 16u);
            const float pv[13] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, p2.w, p3.x, p3.y };
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float *rv = ts.rv[2 * la + c];
                float sacc = 0.f;
#pragma unroll
                for (int u = 0; u < 13; ++u) sacc = fmaf(rv[u], pv[u], sacc);
                if (k < 7) ts.tpose[2 * la + c][k] = sacc;
                else tlm[(2 * la + c) * 192 + 6 * lk + tk] = sacc;
            }
        }
    }
    if (tid == 0) CP_STAMP(22, 0, 4);
    // every wave's 32 x 32 share of Y_hi Y_hi' goes to an LDS tile of its own ([32][33]; three in As, three in Bs, six behind the tail's block):
    // ONE barrier, then (D3) adds the shares of an entry in a fixed order
    auto tile = [&](int w) -> float * {
        return w < 3 ? &sm.ch.As[0][0] + w * (32 * 33) : w < 6 ? &sm.ch.Bs[0][0] + (w - 3) * (32 * 33) : reinterpret_cast<float *>(cp_smem + CP_PART_OFF) + (w - 6) * (32 * 33);
    };
    {
        f32x16_t c2;
#pragma unroll
        for (int e = 0; e < 16; ++e) c2[e] = 0.f;
        for (int K = kg; K < nrb; K += kstep) {
            mma6(fA, fB, c2);
            if (K + kstep < nrb) load_y(K + kstep);
        }
        float *mine = tile(wave);
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[acc_row(e, lane) * 33 + (lane & 31)] = c2[e];
        if (tid == 0) CP_STAMP(22, 0, 5);
        __syncthreads();
        // (D3) D_hi on and below the diagonal (k_hi_fused's S, minus Y Y'), identity padding; X side of the chain: the identity
        for (int idx = tid; idx < NB * NB; idx += CP_NTH) {
            const int r1 = idx >> 6, r2 = idx & 63;
            float v = 0.f;
            if (r2 <= r1) {
                if (r1 < r_hi) {
                    const float *rv = ts.rv[r2];
                    const float *tl = tlm + r1 * 192 + 6 * (r2 >> 1);
                    float g = 0.f;
#pragma unroll
                    for (int u = 0; u < 13; ++u) g = fmaf(rv[u], u < 7 ? ts.tpose[r1][u] : tl[u - 7], g);
                    const int o = (r1 & 31) * 33 + (r2 & 31);
                    float yy = 0.f;
                    if (one_quad) { for (int w = 0; w < 12; ++w) yy += tile(w)[o]; }
                    else { const int q = (r1 >> 5) + (r2 >> 5); for (int w = q; w < 12; w += 3) yy += tile(w)[o]; }
                    v = (g - yy) + (r1 == r2 ? 1.f : 0.f);
                } else v = r1 == r2 ? 1.f : 0.f;
            }
            sm.ch.Ls[r1][r2] = v;
            sm.ch.Xs[r1][r2] = r1 == r2 ? 1.f : 0.f;
        }
    }
    __syncthreads();
    if (tid == 0) CP_STAMP(22, 0, 6);
    return r_hi;
}


// ------------------------------------------------------------------------------------------------------------------------------
// row i >= 2
// ------------------------------------------------------------------------------------------------------------------------------
struct RowSmem {
    __attribute__((aligned(16))) frag_t OL[B3_SGRAN];            // planes of the row's own L(i, J)
    __attribute__((aligned(16))) frag_t NA[B3_SGRAN];            // planes of A(i, J): the operand of the next L(i, J)
    float patch[12][32 * 33];                                    // wave-private transposition patches
    unsigned cnt[4];                                             // arrivals of the four waves behind a hand-over to crit; [2]: bulk release
    unsigned tick[64];                                           // per panel: the next bulk item
};

// all lanes of a wave poll one flag (one request); bounded
__device__ __forceinline__ void wave_wait(const unsigned *p, unsigned target, int32_t *guard)
{
    for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
        if (cf_reached(cf_load(p), target)) return;
        if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
    if ((threadIdx.x & 63) == 0) atomicExch(guard, 1);
}
// lane l < n polls the flag of row k0 + l: ONE wait for every operand of a panel's bulk tiles
__device__ __forceinline__ void wave_wait_rows(unsigned *cf, int k0, int n, unsigned target, int lane, int32_t *guard)
{
    const unsigned *p = cf_rowL(cf, k0 + (lane < n ? lane : 0));
    for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
        const bool ok = lane >= n || cf_reached(cf_load(p), target);
        if (__all(ok)) return;
        if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) atomicExch(guard, 1);
}
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }

// a wave's 32 x 32 tile (accumulator layout, quadrant (fa, fb) of a 64 x 64 block, rows x k) -> its 128 granules of the block's planes:
// through the wave's private patch; fn(gi, p0, p1, p2) receives plane 0's granule index (planes are 128 granules apart)
template <typename F>
__device__ __forceinline__ void wave_tile_granules(float *patch, const float (&v)[16], int fa, int fb, int lane, F &&fn)
{
#pragma unroll
    for (int e = 0; e < 16; ++e) patch[acc_row(e, lane) * 33 + (lane & 31)] = v[e];
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int g = lane + 64 * u, r = g & 31, cg = g >> 5;            // row r, columns 8 cg .. 8 cg + 7 of the quadrant
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = patch[r * 33 + 8 * cg + j];
        u32x4_t p0, p1, p2;
        b3_split3(x, p0, p1, p2);
        fn((2 * fb + (cg >> 1)) * 384 + fa * 64 + 32 * (cg & 1) + r, p0, p1, p2);
    }
    wave_lds_sync();
}

// Row i: twelve waves.  Waves 0-3 are the row's critical path -- L(i, J) as soon as M_J is out, then the tile the next panel (or crit)
// needs; waves 4-11 work through the other tiles of the panel, one quadrant per item, with no workgroup barrier between items.
__device__ __attribute__((noinline)) void row_body(CpArgs a_v, int nrb_v, int i_v)
{
    const CpArgs a = cp_uniform(a_v);
    const int nrb = __builtin_amdgcn_readfirstlane(nrb_v), i = __builtin_amdgcn_readfirstlane(i_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    RowSmem &sm = *reinterpret_cast<RowSmem *>(cp_smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane_in = tid & 63;
    const int lds = nrb * NB;
    const __amdgpu_buffer_rsrc_t rS = cp_rsrc(a.S), rSp = cp_rsrc(a.Sp), rTp = cp_rsrc(a.Tp);
    int32_t *guard = a.status + 1;
    float *patch = sm.patch[wave];
    auto tile_soff = [&](int k) { return (unsigned)((i * NB) * lds + k * NB) * 4u; };
    // NA <- planes of the raw A(i, 0)
    if (tid < 512) {
        const int q = tid >> 7, half = (tid >> 6) & 1, l = tid & 63, r = l & 31, h = l >> 5;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a.S[(size_t)(i * NB + 32 * half + r) * lds + 16 * q + 8 * h + j];
        u32x4_t p0, p1, p2;
        b3_split3(x, p0, p1, p2);
        frag_t *d = sm.NA + q * 384 + half * 64 + l;
        d[0] = __builtin_bit_cast(frag_t, p0); d[128] = __builtin_bit_cast(frag_t, p1); d[256] = __builtin_bit_cast(frag_t, p2);
    }
    if (tid < 4) sm.cnt[tid] = 0;
    if (tid < 64) sm.tick[tid] = 0;
    __syncthreads();
    const int lane0 = lane_in;
    for (int J = 0; J + 2 <= i; ++J) {
        const bool last = J + 2 == i;                           // after this panel the row's leading tiles go to crit
        int lane = lane0;
        asm volatile("" : "+v"(lane));                          // (address arithmetic is redone per panel instead of living in registers across panels)
        auto quad_voff = [&](int fa, int fb) { return acc_voff(lane, lds) + (unsigned)((32 * fa) * lds + 32 * fb) * 4u; };
        if (wave < 4) {
            // ---- L(i, J) = A(i, J) M_J'
            const int fa = (wave >> 1) & 1, fb = wave & 1;
            wave_wait(a.cf + CF_MP, a.base + (unsigned)J + 1, guard);
            if (lane == 0 && wave == 0 && i < 16) CP_STAMP(i, J, 0);
            frag_t fB[4][3];
            frags_sc1(rSp, (unsigned)(J * a.sp_stride + J) * B3_SGRAN, fb, lane, fB);
            f32x16_t c1;
#pragma unroll
            for (int e = 0; e < 16; ++e) c1[e] = 0.f;
            mma6_alds(sm.NA, fa, lane, fB, c1);
            float v[16];
            const unsigned tv = quad_voff(fa, fb);
#pragma unroll
            for (int e = 0; e < 16; ++e) { v[e] = c1[e]; st_f32(c1[e], rS, tv, tile_soff(J) + acc_soff(e, lds)); }      // the final factor (not read again here)
            wave_tile_granules(patch, v, fa, fb, lane, [&](int gi, u32x4_t p0, u32x4_t p1, u32x4_t p2) {
                sm.OL[gi] = __builtin_bit_cast(frag_t, p0); sm.OL[gi + 128] = __builtin_bit_cast(frag_t, p1); sm.OL[gi + 256] = __builtin_bit_cast(frag_t, p2);
                const unsigned gb = ((unsigned)(i * a.sp_stride + J) * B3_SGRAN + gi) * 16u;
                st16_sc1(p0, rSp, gb); st16_sc1(p1, rSp, gb + 128 * 16); st16_sc1(p2, rSp, gb + 256 * 16);
            });
            // (round 6: in the row's LAST panel these waves go straight on to the tile crit waits for; L(i, J)'s publication -- wanted by the rows below and
            //  the strips, not by crit -- rides with that tile's drain and flag, ~2.5 us later, instead of standing 1 us in front of it: rl_last)
            if (!(last && a.row_late)) drain_stores();
        }
        __syncthreads();                                        // OL complete and drained; every tile store of the previous panel is ordered
        if (tid == 0 && !(last && a.row_late)) { cf_store(cf_rowL(a.cf, i), a.base + (unsigned)J + 1); if (i < 16) CP_STAMP(i, J, 1); }
        if (last && wave >= 4 && wave < 8) {
            // the diagonal tile A(i, i) -= L(i, J) L(i, J)' needs nothing from outside: waves 4-7 (idle in the row's last panel) send it to crit (in f32)
            // while waves 0-3 wait for L(J+1, J) and build the other tile  (round 5: the two tiles used to follow each other on waves 0-3, and crit had
            // them 1.1 us later -- at the very end of its chain)
            const int fa = (wave >> 1) & 1, fb = wave & 1;
            const unsigned tv = quad_voff(fa, fb);
            frag_t fB[4][3];
            float v[16];
            f32x16_t c2;
            frags_lds(sm.OL, fb, lane, fB);
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = ld_f32(rS, tv, tile_soff(i) + acc_soff(e, lds));
#pragma unroll
            for (int e = 0; e < 16; ++e) c2[e] = 0.f;
            mma6_alds(sm.OL, fa, lane, fB, c2);
#pragma unroll
            for (int e = 0; e < 16; ++e) patch[acc_row(e, lane) * 33 + (lane & 31)] = v[e] - c2[e];
            wave_lds_sync();
#pragma unroll
            for (int u = 0; u < 4; ++u) {                    // 32 rows x 8 pieces of 16 bytes
                const int g = lane + 64 * u, r = g >> 3, c4 = (g & 7) * 4;
                const float *y = patch + r * 33 + c4;
                st16_sc1(__builtin_bit_cast(u32x4_t, f4v_t{ y[0], y[1], y[2], y[3] }), rS, (unsigned)(((size_t)(i * NB + 32 * fa + r) * lds + i * NB + 32 * fb + c4) * 4));
            }
            wave_lds_sync();
            drain_stores();
            if (lane == 0 && wave == 4 && i < 16) CP_STAMP(i, J, 3);
            if (lane == 0 && atomicAdd(&sm.cnt[0], 1u) == 7u) { if (a.row_late) cf_store(cf_rowL(a.cf, i), a.base + (unsigned)J + 1); cf_store(cf_rowA(a.cf, i), a.base + 2u); if (i < 16) CP_STAMP(i, J, 4); }
        }
        if (wave < 4) {
            // ---- the tile the next panel starts from: A(i, J+1) -= L(i, J) L(J+1, J)'
            const int fa = (wave >> 1) & 1, fb = wave & 1;
            const unsigned tv = quad_voff(fa, fb);
            frag_t fB[4][3];
            float v[16];
            f32x16_t c2;
            wave_wait(cf_rowL(a.cf, J + 1), a.base + (unsigned)J + 1, guard);
            if (lane == 0 && wave == 0 && i < 16) CP_STAMP(i, J, 2);
            frags_sc1(rSp, (unsigned)((J + 1) * a.sp_stride + J) * B3_SGRAN, fb, lane, fB);
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = ld_f32(rS, tv, tile_soff(J + 1) + acc_soff(e, lds));
#pragma unroll
            for (int e = 0; e < 16; ++e) c2[e] = 0.f;
            mma6_alds(sm.OL, fa, lane, fB, c2);
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] -= c2[e];
            if (!last) {
                wave_tile_granules(patch, v, fa, fb, lane, [&](int gi, u32x4_t p0, u32x4_t p1, u32x4_t p2) {
                    sm.NA[gi] = __builtin_bit_cast(frag_t, p0); sm.NA[gi + 128] = __builtin_bit_cast(frag_t, p1); sm.NA[gi + 256] = __builtin_bit_cast(frag_t, p2);
                });
            } else {
                // to crit, as planes; the flag's second stage: both tiles are out
                wave_tile_granules(patch, v, fa, fb, lane, [&](int gi, u32x4_t p0, u32x4_t p1, u32x4_t p2) {
                    const unsigned gb = ((unsigned)i * B3_SGRAN + gi) * 16u;
                    st16_sc1(p0, rTp, gb); st16_sc1(p1, rTp, gb + 128 * 16); st16_sc1(p2, rTp, gb + 256 * 16);
                });
                drain_stores();
                if (lane == 0 && atomicAdd(&sm.cnt[0], 1u) == 7u) { if (a.row_late) cf_store(cf_rowL(a.cf, i), a.base + (unsigned)J + 1); cf_store(cf_rowA(a.cf, i), a.base + 2u); if (i < 16) CP_STAMP(i, J, 4); }
            }
        }
        if (!last) {
            // ---- the other tiles, A(i, k) -= L(i, J) L(k, J)' for k = J+2 .. i: item = (tile, quadrant).  Every wave draws items from a ticket
            //      counter in LDS -- waves 4-11 from the start of the panel, waves 0-3 once the leading tile is done -- with no workgroup barrier
            //      between items.
            const int n_items = 4 * (i - J - 1);
            // Rows J+2 .. i-1 have published L(k, J): wave 4 waits for all of them with ONE poll loop, issues ONE agent-scope acquire for the
            // workgroup and, once its invalidate has completed, releases the other waves through a word in LDS -- their operand loads are then
            // plain loads, served by the XCD's L2 (the same block is wanted by every row below it; the rows share crit's XCD, see k_cholp)
            if (wave == 4) {
                if (i - J - 2 > 0) {
                    wave_wait_rows(a.cf, J + 2, i - J - 2, a.base + (unsigned)J + 1, lane, guard);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (lane == 0) __hip_atomic_store(&sm.cnt[2], (unsigned)J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
                    if (__hip_atomic_load(&sm.cnt[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)J + 1) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            auto ticket = [&]() { return __builtin_amdgcn_readfirstlane(lane == 0 ? (int)atomicAdd(&sm.tick[J], 1u) : 0); };
            int t = ticket();
            if (t < n_items) {
                // the B fragments of the NEXT item are requested k-step by k-step into the registers the current item has just multiplied from
                frag_t fB[4][3];
                auto load_bq = [&](int tt, int q, frag_t (&d)[4][3]) {
                    const int k = J + 2 + (tt >> 2), fb = tt & 1;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        d[q][pl] = k < i ? ld_granule(rSp, (unsigned)(fb * 64 + lane) * 16u, (unsigned)(k * a.sp_stride + J) * B3_SGRAN + q * 384 + pl * 128, 0)
                                         : sm.OL[q * 384 + pl * 128 + fb * 64 + lane];
                };
#pragma unroll
                for (int q = 0; q < 4; ++q) load_bq(t, q, fB);
                while (t < n_items) {
                    const int tn = ticket();                     // the next item (its operands are requested during this one)
                    const int k = J + 2 + (t >> 2), fa = (t >> 1) & 1, fb = t & 1;
                    const unsigned tv = quad_voff(fa, fb);
                    const bool nx = tn < n_items;
                    float v[16];
                    f32x16_t c2;
#pragma unroll
                    for (int e = 0; e < 16; ++e) c2[e] = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const frag_t *ap = sm.OL + q * 384 + fa * 64 + lane;
                        const frag_t a0 = ap[0], a1 = ap[128], a2 = ap[256];
#define RW_MMA(ax, py) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, fB[q][py]), c2, 0, 0, 0)
                        RW_MMA(a0, 0); RW_MMA(a0, 1); RW_MMA(a1, 0); RW_MMA(a1, 1); RW_MMA(a0, 2); RW_MMA(a2, 0);
#undef RW_MMA
                        if (q == 0) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) v[e] = ld_f32(rS, tv, tile_soff(k) + acc_soff(e, lds));
                        }
                        if (nx) load_bq(tn, q, fB);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) st_f32(v[e] - c2[e], rS, tv, tile_soff(k) + acc_soff(e, lds));
                    t = tn;
                }
            }
        }
        __syncthreads();                                        // the panel's tiles are stored; NA complete
        if (tid == 0 && i < 16) CP_STAMP(i, J, 5);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// the rescue stage's projection and chi2 gate (rescue_hi_inliers.m:32-43), on the strips' CUs behind their x-update: strip s takes the landmarks
// i = s (mod n_strips), one wave per landmark
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rdlane_d(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ld_f64_sc1(const double *p)
{
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_f64_sc1(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int DG_SLOTS = 6, DG_KMAX = 12, DG_WORDS = 16, DG_SLOT_GRAN = 3 * 2 * 64;     // down-date consumers (dd_body)
constexpr int CP_WS = 32 + 1;                        // row stride of the strip's f32 transposition buffer
constexpr int CP_WGRAN = 4 * 3 * 64;                 // granules of one 64 x 32 block of W as B-operand planes: [q 4][plane 3][lane 64]
// LDS of the gate (the strip's CP | Yw2 region, free behind the x-update): one slot per wave of a round, y of the two candidates in hand, the
// waves' partial sums
struct GateSmem {
    struct Slot { float hjf[2][16]; double hj[2][14]; double q00, q01, q11; int lm, off, type, had; double h_old[2], z[2]; double hc[14], hl[12], zi[2]; } slot[8];      // hjf[c][0..12] = row c of H J (f32: the update's rows; hj: the same values as doubles), [13] = nu; hc / hl / zi: the projection's outputs
    double part[8][3];
};
constexpr size_t CP_GATE_OFF = 9 * 1024;            // behind the x-update's scratch (c 4 KB | sums 4 KB | quaternion), inside the strip's CP | Yw2 region
static_assert(CP_GATE_OFF + sizeof(GateSmem) <= 20 * 1024, "the gate's LDS is the strip's CP | Yw2 region");
__device__ __forceinline__ GateSmem &gate_smem(int win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    return *reinterpret_cast<GateSmem *>(cp_smem + (size_t)win * CP_WGRAN * 16 + CP_GATE_OFF);
}

// a strip's TailSmem lies behind its plane slots (crit's behind CritSmem); its tpose part, which only crit uses, holds the strip's prefetched records
__device__ __forceinline__ unsigned char *strip_tail_smem(int win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    return cp_smem + (((size_t)(win + 1) * CP_WGRAN * 16 + (size_t)NB * CP_WS * sizeof(float) + 127) / 128 * 128);
}
struct GatePre { int lm, off, type, had; double h_old[2], z[2]; };

// What the first round's projections need from the tables (candidate list -> landmark -> type / offset / the kept prediction / the measured pixel:
// three dependent load levels) is fetched when the strip starts -- it then waits for crit's first panel anyway: wave w, lane 0 -> record w
__device__ __attribute__((noinline)) void gate_prefetch(int rows_v, int s_v)
{
    const CpArgs a = args_from_lds<CpArgs>(CP_TA_OFF);
    const CpTail t = args_from_lds<CpTail>(CP_T_OFF);
    const int s = __builtin_amdgcn_readfirstlane(s_v), wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int n_all = t.m - __builtin_amdgcn_readfirstlane(rows_v) / 2, r = s + a.n_strips * wave;
    if ((threadIdx.x & 63) != 0 || r >= n_all) return;
    GatePre &sl = reinterpret_cast<GatePre *>(&reinterpret_cast<TailSmem *>(strip_tail_smem(a.win))->tpose[0][0])[wave];
    const int i = t.sel_rows[t.m + r] & 0xffff;
    sl.lm = i; sl.off = t.lm_off[i]; sl.type = t.lm_type[i];
    sl.had = t.has_h[i];
    sl.h_old[0] = t.h[2 * i]; sl.h_old[1] = t.h[2 * i + 1];
    sl.z[0] = t.z[2 * i]; sl.z[1] = t.z[2 * i + 1];
}

// projection + Jacobian of landmark i at x_k_k (lane 0 of the calling wave; predict_camera_measurements.m / calculate_derivatives.m through
// project_core), once the strips that own the pose and the landmark's entries have x out (their flag also covers every panel of their W).
// Out of line: its fp64 geometry takes most of the register file, and the gate calls it from three places.
// (inlined at its ONE call site in the gate: an out-of-line function with this much fp64 geometry saves and restores sixty-four callee-saved
//  registers per call -- about a microsecond on the rescue stage's path, and the gate sat three calls deep)
__device__ __forceinline__ void gate_project(const CpArgs &a, const CpTail &t, const int i_in, const int wave)          // i_in < 0: the landmark and its table entries are in the wave's slot
{
    GateSmem::Slot &sl = gate_smem(a.win).slot[wave];
    const int lane = threadIdx.x & 63;
    int32_t *guard = a.status + 1;
    const bool slot = i_in < 0;
    const int i = slot ? sl.lm : i_in;
    const int type = slot ? sl.type : t.lm_type[i], off = slot ? sl.off : t.lm_off[i], d = type == PRE3_INVDEPTH ? 6 : 3;
    {
        const int sw = lane == 0 ? 0 : lane == 1 ? (off >> 5) : ((off + d - 1) >> 5);
        const unsigned *fp = cf_strip(a.cf, sw);
        bool gave_up = true;
        for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
            const bool ok = lane >= 3 || cf_reached(cf_load(fp), a.base + CFV_X);
            if (__all(ok)) { gave_up = false; break; }
            if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gave_up = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (gave_up && lane == 0) atomicExch(guard, 1);
    }
    if (lane != 0) return;
    double Hc[14] = { 0 }, Hl[12] = { 0 }, zi[2] = { 0, 0 };
    double xp[7], yl[6];
#pragma unroll
    for (int u = 0; u < 7; ++u) xp[u] = ld_f64_sc1(a.x_out + u);
#pragma unroll
    for (int u = 0; u < 6; ++u) yl[u] = u < d ? ld_f64_sc1(a.x_out + off + u) : 0.0;
    const int had = slot ? sl.had : t.has_h[i];
    double h_old[2] = { 0, 0 };
    if (had) { h_old[0] = slot ? sl.h_old[0] : t.h[2 * i]; h_old[1] = slot ? sl.h_old[1] : t.h[2 * i + 1]; }
    bool fresh = false;
    const bool now = project_core(type, xp, yl, t.cam, had, h_old, zi, fresh, Hc, Hl);
    if (fresh) { t.h[2 * i] = zi[0]; t.h[2 * i + 1] = zi[1]; }
    t.has_h[i] = now ? 1 : 0;
    if (now) {
#pragma unroll
        for (int u = 0; u < 14; ++u) t.Hc[14 * i + u] = Hc[u];
#pragma unroll
        for (int u = 0; u < 12; ++u) t.Hl[12 * i + u] = Hl[u];
    } else {
        // (never predicted: the table keeps what it held -- zeros since pre3_set_map -- and the gate reads those)
#pragma unroll
        for (int u = 0; u < 14; ++u) Hc[u] = t.Hc[14 * i + u];
#pragma unroll
        for (int u = 0; u < 12; ++u) Hl[u] = t.Hl[12 * i + u];
        zi[0] = t.h[2 * i]; zi[1] = t.h[2 * i + 1];
    }
#pragma unroll
    for (int u = 0; u < 14; ++u) sl.hc[u] = Hc[u];
#pragma unroll
    for (int u = 0; u < 12; ++u) sl.hl[u] = Hl[u];
    sl.zi[0] = zi[0]; sl.zi[1] = zi[1];
}

// The strip's flag is stored INSIDE (behind its candidates): the projections of the other landmarks follow it, off the rescue stage's path.
// Candidates (individually compatible, not a low-innovation inlier: rescue_hi_inliers.m:36) were listed by the selection stage (sel_rows[m ..], in
// measurement order): the r-th goes to strip r mod n_strips, so that every strip has its share whatever the landmarks' order; of the other
// landmarks strip s takes i = s (mod n_strips).
__device__ __forceinline__ void tail_gate_body(const CpArgs &a, const CpTail &t, const int nrb, const int s, const int rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    GateSmem &gs = gate_smem(a.win);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const __amdgpu_buffer_rsrc_t rYp = cp_rsrc(t.Yp), rHb = cp_rsrc(t.Hb), rP = cp_rsrc(a.P);
    const int n_all = t.m - rows / 2;                             // candidates of the frame: the measurements the selection stage did not take
    const int n_cand = n_all > s ? (n_all - s + a.n_strips - 1) / a.n_strips : 0, n_rounds = (n_cand + 7) / 8;
    const int32_t *cand = t.sel_rows + t.m;
    auto gate_flag = [&]() {                                      // every candidate's planes, rows and flag have left: the strip's flag for crit
        drain_stores();
        __syncthreads();
        if (tid == 0) cf_store(cf_strip(a.cf, s), a.base + CFV_GATE);
        if (s == 1 && tid == 0) CP_STAMP(23, 1, 3);
    };
    if (n_rounds == 0) gate_flag();
    // Iterations 0 .. n_rounds-1 are the candidates' rounds (wave w: candidate 8 it + w; the whole workgroup meets at their barriers); behind them
    // every wave works through the other landmarks i = s (mod n_strips) on its own (rescue_hi_inliers.m:32-33 projects and linearises every
    // landmark).  ONE projection site for both.
    int j_other = wave;
    for (int it = 0; ; ++it) {
        const bool cround = it < n_rounds;
        const int c0 = 8 * it, n_round = cround ? (n_cand - c0 < 8 ? n_cand - c0 : 8) : 0;
        int item = -2;                                            // -2: nothing; -1: the wave's slot; >= 0: a landmark
        if (cround) {
            if (wave < n_round) {
                GateSmem::Slot &sl = gs.slot[wave];
                if (lane == 0) {
                    if (c0 == 0) {                                 // (the first round's table entries came with gate_prefetch)
                        const GatePre &pr = reinterpret_cast<const GatePre *>(&reinterpret_cast<const TailSmem *>(strip_tail_smem(a.win))->tpose[0][0])[wave];
                        sl.lm = pr.lm; sl.off = pr.off; sl.type = pr.type; sl.had = pr.had;
                        sl.h_old[0] = pr.h_old[0]; sl.h_old[1] = pr.h_old[1]; sl.z[0] = pr.z[0]; sl.z[1] = pr.z[1];
                    } else {
                        const int i = cand[s + a.n_strips * (c0 + wave)] & 0xffff;
                        sl.lm = i; sl.off = t.lm_off[i]; sl.type = t.lm_type[i]; sl.had = t.has_h[i];
                        sl.h_old[0] = t.h[2 * i]; sl.h_old[1] = t.h[2 * i + 1]; sl.z[0] = t.z[2 * i]; sl.z[1] = t.z[2 * i + 1];
                    }
                }
                wave_lds_sync();
                item = -1;
            }
        } else {
            for (; s + a.n_strips * j_other < t.N && item == -2; j_other += 8) {
                const int i = s + a.n_strips * j_other;
                if (!(t.lm_ic[i] == 1 && t.lm_li[i] == 0)) item = i;      // (a candidate is some strip's round's)
            }
            if (item == -2) break;
        }
        if (s == 1 && it == 0 && wave == 0 && lane == 0) CP_STAMP(23, 1, 0);
        if (item != -2) gate_project(a, t, item, wave);
        if (s == 1 && it == 0 && wave == 0 && lane == 0) CP_STAMP(23, 1, 1);
        if (!cround) continue;
        // ---- the round's slots: H J (the normalisation Jacobian of the LI update folded into the quaternion columns), nu,
        //      q = (H J) P (H J)' from the 13 x 13 block of P (innovation_body's sum)
        if (wave < n_round) {
            GateSmem::Slot &sl = gs.slot[wave];
            const int off = sl.off, d = sl.type == PRE3_INVDEPTH ? 6 : 3;
            if (lane == 0) {
                const double *Hc = sl.hc, *Hl = sl.hl, *zi = sl.zi;
                double Jn[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) Jn[u] = ld_f64_sc1(a.params + 16 + u);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) sl.hjf[c][u] = (float)Hc[7 * c + u];
                    const double *hq = Hc + 7 * c + 3;
#pragma unroll
                    for (int b = 0; b < 4; ++b) sl.hjf[c][3 + b] = (float)fma(hq[3], Jn[12 + b], fma(hq[2], Jn[8 + b], fma(hq[1], Jn[4 + b], hq[0] * Jn[b])));
#pragma unroll
                    for (int u = 0; u < 6; ++u) sl.hjf[c][7 + u] = (float)Hl[6 * c + u];
                    sl.hjf[c][13] = (float)(sl.z[c] - zi[c]);
                    sl.hjf[c][14] = 0.f; sl.hjf[c][15] = 0.f;
#pragma unroll
                    for (int u = 0; u < 13; ++u) sl.hj[c][u] = (double)sl.hjf[c][u];
                }
            }
            wave_lds_sync();
            double q00 = 0, q01 = 0, q11 = 0;
            if (lane < 7 + d) {
                const int ib = lane < 7 ? lane : off + lane - 7;
                const unsigned rb = (unsigned)ib * (unsigned)a.ld * 4u, ob = (unsigned)off * 4u;
                const f4v_t p0 = ld4f(rP, rb), p1 = ld4f(rP, rb + 16u), p2 = ld4f(rP, rb + ob), p3 = ld4f(rP, rb + ob + 16u);
                const float pv[13] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, p2.w, p3.x, p3.y };
                double hp0 = 0, hp1 = 0;
#pragma unroll
                for (int u = 0; u < 13; ++u) { hp0 = fma((double)sl.hjf[0][u], (double)pv[u], hp0); hp1 = fma((double)sl.hjf[1][u], (double)pv[u], hp1); }
                const double h0b = (double)sl.hjf[0][lane], h1b = (double)sl.hjf[1][lane];
                q00 = hp0 * h0b; q01 = hp0 * h1b; q11 = hp1 * h1b;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { q00 += __shfl_xor(q00, o); q01 += __shfl_xor(q01, o); q11 += __shfl_xor(q11, o); }
            if (lane == 0) { sl.q00 = q00; sl.q01 = q01; sl.q11 = q11; }
        }
        __syncthreads();
        // ---- y = (H J) W' of the round's candidates over the LI update's rows, from the column-major copy of W: one or two waves per candidate, a
        //      thread takes eight consecutive k of the landmark's 13 columns (26 loads in flight, all lines fully used) -- a granule of each plane of y
        //      straight from registers; then y y' and the gate nu' inv(q - y y') nu < chi2 (no R: rescue_hi_inliers.m:39)
        const int wpc = (nrb * 8 + 63) / 64, cpp = 8 / wpc;       // waves per candidate (nrb <= 16: one or two), candidates per pass
        const __amdgpu_buffer_rsrc_t rWt = cp_rsrc(a.Wt);
        for (int w0 = 0; w0 < n_round; w0 += cpp) {
            const int cnd = wave / wpc, g = (wave - cnd * wpc) * 64 + lane;
            const bool on = w0 + cnd < n_round && cnd < cpp, live = on && g < nrb * 8;
            double sm_[3] = { 0, 0, 0 };
            if (on) {
                const GateSmem::Slot &sl = gs.slot[w0 + cnd];
                if (live) {
                    const int off = sl.off, i = sl.lm;
                    f4v_t wv[13][2];
#pragma unroll
                    for (int u = 0; u < 13; ++u) {
                        const unsigned cb = (unsigned)((u < 7 ? u : off + u - 7) * a.kcap + 8 * g) * 4u;
                        wv[u][0] = ld4f_sc1(rWt, cb); wv[u][1] = ld4f_sc1(rWt, cb + 16u);
                    }
                    double a0[8], a1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a0[j] = 0; a1[j] = 0; }
#pragma unroll
                    for (int u = 0; u < 13; ++u) {
                        const double h0 = sl.hj[0][u], h1 = sl.hj[1][u];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const double w = (double)(j < 4 ? wv[u][0][j] : wv[u][1][j - 4]);
                            a0[j] = fma(h0, w, a0[j]); a1[j] = fma(h1, w, a1[j]);
                        }
                    }
                    float y0[8], y1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        y0[j] = (float)a0[j]; y1[j] = (float)a1[j];
                        sm_[0] = fma((double)y0[j], (double)y0[j], sm_[0]); sm_[1] = fma((double)y0[j], (double)y1[j], sm_[1]); sm_[2] = fma((double)y1[j], (double)y1[j], sm_[2]);
                    }
                    u32x4_t pa, pb, pc;
                    const unsigned g0 = (unsigned)(((i * 2 + 0) * 3) * t.ykcap + 8 * g) * 2u, pst = (unsigned)t.ykcap * 2u;
                    b3_split3(y0, pa, pb, pc);
                    st16_sc1(pa, rYp, g0); st16_sc1(pb, rYp, g0 + pst); st16_sc1(pc, rYp, g0 + 2u * pst);
                    b3_split3(y1, pa, pb, pc);
                    st16_sc1(pa, rYp, g0 + 3u * pst); st16_sc1(pb, rYp, g0 + 4u * pst); st16_sc1(pc, rYp, g0 + 5u * pst);
                }
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    double x = sm_[v];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
                    if (lane == 0) gs.part[wave][v] = x;
                }
            }
            __syncthreads();
            if (on && g == 0) {                                   // the gate (first lane of the candidate's first wave)
                const GateSmem::Slot &sl = gs.slot[w0 + cnd];
                double y00 = 0, y01 = 0, y11 = 0;
                for (int v = 0; v < wpc; ++v) { y00 += gs.part[wave + v][0]; y01 += gs.part[wave + v][1]; y11 += gs.part[wave + v][2]; }
                const double g00 = sl.q00 - y00, g01 = sl.q01 - y01, g11 = sl.q11 - y11;
                const double det = g00 * g11 - g01 * g01;
                const double i00 = g11 / det, i01 = -g01 / det, i11 = g00 / det;
                const double nu0 = (double)sl.hjf[0][13], nu1 = (double)sl.hjf[1][13];
                const double t0 = nu0 * i00 + nu1 * i01, t1 = nu0 * i01 + nu1 * i11;
                const double d2 = t0 * nu0 + t1 * nu1;
                st_i32_sc1(t.lm_hi + sl.lm, d2 < t.chi2 ? 1 : 0);
#ifdef PRE3_TAIL_DEBUG
                t.dbgS[4 * sl.lm + 0] = sl.q00; t.dbgS[4 * sl.lm + 1] = y00; t.dbgS[4 * sl.lm + 2] = sl.q11; t.dbgS[4 * sl.lm + 3] = y11;
#endif
            }
            if (on && g >= 8 && g < 16) {                         // the rows of H J for the update: [c][0..12], [13] = nu
                const GateSmem::Slot &sl = gs.slot[w0 + cnd];
                const int c = (g - 8) >> 2, g4 = (g - 8) & 3;
                st16_sc1(*reinterpret_cast<const u32x4_t *>(&sl.hjf[c][4 * g4]), rHb, (unsigned)((sl.lm * 2 + c) * 16 + 4 * g4) * 4u);
            }
            __syncthreads();
        }
        if (s == 1 && it == 0 && tid == 0) CP_STAMP(23, 1, 2);
        if (it == n_rounds - 1) gate_flag();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// strip s: 32 columns of [HP | nu]
// ------------------------------------------------------------------------------------------------------------------------------

// (the tail's arguments reach a strip through LDS: sixty more argument registers live through the panel loop cost it a kilobyte of scratch)

__device__ __attribute__((noinline)) void strip_tail_body(int nrb_v, int s_v, int rows_v);
__device__ __attribute__((noinline)) void gate_prefetch(int rows_v, int s_v);

// The rescue stage's projection (rescue_hi_inliers.m:31-32: every landmark at x_k_k) on the strips' CUs, idle behind their x-update while the
// consumers finish P (round 5, without the in-launch tail): strip s takes the landmarks i = s (mod n_strips), one wave each, lane 0 runs the
// fp64 geometry once the strips that own the pose and the landmark's entries have x out.  The gate that rides with the Jnorm pass behind this
// launch then starts from h / H instead of waiting 4.5 us for them.  Same project_core as project_one: the same bits.
__device__ __attribute__((noinline)) void strip_proj_body(int s_v)
{
    const CpArgs a = args_from_lds<CpArgs>(CP_TA_OFF);
    const CpTail t = args_from_lds<CpTail>(CP_T_OFF);
    const int s = __builtin_amdgcn_readfirstlane(s_v), wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    int32_t *guard = a.status + 1;
    for (int i = s + a.n_strips * wave; i < t.N; i += a.n_strips * 8) {
        const int type = t.lm_type[i], off = t.lm_off[i], d = type == PRE3_INVDEPTH ? 6 : 3;
        {
            const int sw = lane == 0 ? 0 : lane == 1 ? (off >> 5) : ((off + d - 1) >> 5);
            const unsigned *fp = cf_strip(a.cf, sw);
            bool gave_up = true;
            for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
                const bool ok = lane >= 3 || cf_reached(cf_load(fp), a.base + CFV_X);
                if (__all(ok)) { gave_up = false; break; }
                if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gave_up = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (gave_up && lane == 0) atomicExch(guard, 1);
        }
        if (lane != 0) continue;
        // (as project_one, pre3_geomdev.h: the Jacobians go straight to the tables)
        double zi[2] = { 0, 0 };
        double xp[7], yl[6];
#pragma unroll
        for (int u = 0; u < 7; ++u) xp[u] = ld_f64_sc1(a.x_out + u);
#pragma unroll
        for (int u = 0; u < 6; ++u) yl[u] = u < d ? ld_f64_sc1(a.x_out + off + u) : 0.0;
        const int had = t.has_h[i];
        double h_old[2] = { 0, 0 };
        if (had) { h_old[0] = t.h[2 * i]; h_old[1] = t.h[2 * i + 1]; }
        bool fresh = false;
        const bool now = project_core(type, xp, yl, t.cam, had, h_old, zi, fresh, t.Hc + 14 * i, t.Hl + 12 * i);
        if (fresh) { t.h[2 * i] = zi[0]; t.h[2 * i + 1] = zi[1]; }
        t.has_h[i] = now ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 6: the strip's panel loop RIGHT-LOOKING and flag-driven (PRE3_STRIP_RL, default).  The left-looking loop below sums, in front of every M_J,
// J blocks L(J+1, K) W_K whose L fragments (J x 24.6 KB per strip and panel, the same bytes for all 97 strips at the same moment) bound it at 2-2.7 us
// per panel from J = 5 on -- after round 6's crit a strip's panel (8.5-8.7 us) was longer than crit's (8.0) and the strips finished ~3 us behind it --,
// and its fifteen workgroup barriers per panel couple all eight waves to whichever is slowest.  Here every 32 x 32 tile of the right-hand side has ONE
// owner wave (tile (j, fa) of C_j = HP_j - sum_K L(j, K) W_K: wave fa + 2 (j mod 4)) that holds it as an accumulator from start to finish and adds
// L(j, J) W_J to it as soon as W_J exists -- the fragments of L(j, J) are then wanted by two waves at a time, spread over the whole panel, and only
// the term j = J+1 is urgent.  The critical path of a panel is one owner pair's:  C_J -> planes (v_permlane32_swap: no transposition buffer) -> W_J[fa] =
// M_J[fa rows] C_J (24 MFMAs, no reduction) -> planes -> the next pair's C_{J+1} += L(J+1, J) W_J (its fragments requested before W_J is there).
// Hand-offs inside the workgroup are LDS counters (pre3_chain_async.h's recipe: payload, then counter; counter read first), no barrier in the loop.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int RL_MAXT = 4;                           // tiles per wave: row blocks j = par + 4 t < nrb <= 16
enum { RLF_C0 = 0 /* + fa: C_J's planes of row half fa are in CPb (value J + 1) */, RLF_W0 = 2 /* + fa: W_J's */, RLF_DR0 = 4 /* + (J & 3): waves of panel J whose stores have drained */,
       RLF_PUB = 8 /* panels whose consumer flag is up */, RLF_N = 12 };
// a wave's 32 x 32 accumulator (rows = k of the next product) -> its granules of the B-operand planes: k-steps 2 fa + g2, this lane's eight consecutive k
template <typename F>
__device__ __forceinline__ void rl_acc_granules(const f32x16_t &v, const float sign, F &&fn)
{
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = sign * v[8 * g2 + j], hi = sign * v[8 * g2 + 4 + j];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
            const unsigned s0 = sw[0], s1 = sw[1];
            x[j] = __uint_as_float(s0); x[4 + j] = __uint_as_float(s1);
        }
        u32x4_t p0, p1, p2;
        b3_split3(x, p0, p1, p2);
        fn(g2, x, p0, p1, p2);
    }
}
__device__ __forceinline__ void strip_rl_loop(const CpArgs &a, const int nrb, const int s, frag_t *WPl, frag_t *CPb, unsigned *sf)
{
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int fa = wave & 1, par = wave >> 1, win = a.win;
    const int c0 = s * 32;
    const bool nu_strip = c0 == a.ld, publish = a.n_dd > 0 || a.xu != 0, to_wp = c0 < a.ld + NB;
    int32_t *guard = a.status + 1;
    const __amdgpu_buffer_rsrc_t rW = cp_rsrc(a.W), rSp = cp_rsrc(a.Sp), rWp = cp_rsrc(a.Wp);
    const unsigned wvoff = acc_voff(lane, a.ldw) + (unsigned)((32 * fa) * a.ldw + c0) * 4u;
    const unsigned plo = (unsigned)(fa * 64 + lane) * 16u;
    if (tid < RLF_N) sf[tid] = 0u;
    // this wave's tiles: C[t] = sum_K L(j, K) W_K - HP_j, j = par + 4 t (the raw rows, cold in HBM, are all requested now)
    f32x16_t C[RL_MAXT];
#pragma unroll
    for (int t = 0; t < RL_MAXT; ++t) {
        const int j = par + 4 * t;
#pragma unroll
        for (int e = 0; e < 16; ++e) C[t][e] = j < nrb ? -ld_f32(rW, wvoff, (unsigned)(j * NB * a.ldw) * 4u + acc_soff(e, a.ldw)) : 0.f;
    }
    __syncthreads();                                                     // (the counters are reset)
    frag_t fA[4][3];
    auto load_A = [&](const unsigned blk) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fA[q][pl] = ld_granule(rSp, plo, blk + q * 384 + pl * 128, 1);
    };
    auto lds_wait2 = [&](const int k0, const unsigned need) {            // both halves' counters
        if (!cha_wait(sf + k0, need) || !cha_wait(sf + k0 + 1, need)) { if (lane == 0) atomicExch(guard, 1); }
    };
#define RL_MMA(acc_, fa_, bb) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa_), __builtin_bit_cast(bf16x8_t, bb), acc_, 0, 0, 0)
    for (int J = 0; J < nrb; ++J) {
        // (the owner's part is instantiated per tile slot: a run-time index into the register tiles would copy one -- sixteen more registers beside
        //  four tiles and twelve fragments -- and the tiles came back as scratch traffic: +18 MB written per launch, measured)
#pragma unroll
        for (int tJ = 0; tJ < RL_MAXT; ++tJ) {
        if (J == par + 4 * tJ) {
            // ---- this wave owns C_J (row half fa): (1) its planes -> CPb
            f32x16_t &wacc = C[tJ];                                       // (C_J is spent once its planes are out: the slot takes W_J)
            rl_acc_granules(C[tJ], -1.f, [&](int g2, const float (&)[8], u32x4_t p0, u32x4_t p1, u32x4_t p2) {
                frag_t *d = CPb + (2 * fa + g2) * 192 + lane;
                d[0] = __builtin_bit_cast(frag_t, p0); d[64] = __builtin_bit_cast(frag_t, p1); d[128] = __builtin_bit_cast(frag_t, p2);
            });
            cha_store(sf + RLF_C0 + fa, (unsigned)J + 1, lane);
            if (fa == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 3);
            // (2) M_J's fragments (this row half, all four k-steps) as soon as its flag is up; (3) the other half's planes of C_J
            wave_wait(a.cf + CF_MP, a.base + (unsigned)J + 1, guard);
            if (fa == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 1);
            load_A((unsigned)(J * a.sp_stride + J) * B3_SGRAN);
            lds_wait2(RLF_C0, (unsigned)J + 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) wacc[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const frag_t b0 = CPb[q * 192 + lane], b1 = CPb[q * 192 + 64 + lane], b2 = CPb[q * 192 + 128 + lane];
                RL_MMA(wacc, fA[q][0], b0); RL_MMA(wacc, fA[q][0], b1); RL_MMA(wacc, fA[q][1], b0); RL_MMA(wacc, fA[q][1], b1); RL_MMA(wacc, fA[q][0], b2); RL_MMA(wacc, fA[q][2], b0);
            }
            if (fa == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 5);
            // (4) W_J[fa]: f32 to memory (the x-update and k_gain read it; the nu column written through: every strip reads it), planes to the ring slot
            //     and -- write-through -- to k_downdate_b3's image
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (nu_strip) st_f32_sc1(wacc[e], rW, wvoff, (unsigned)(J * NB * a.ldw) * 4u + acc_soff(e, a.ldw));
                else st_f32(wacc[e], rW, wvoff, (unsigned)(J * NB * a.ldw) * 4u + acc_soff(e, a.ldw));
            }
            rl_acc_granules(wacc, 1.f, [&](int g2, const float (&)[8], u32x4_t p0, u32x4_t p1, u32x4_t p2) {
                const int q = 2 * fa + g2;
                frag_t *d = WPl + (size_t)(J % win) * CP_WGRAN + q * 192 + lane;
                d[0] = __builtin_bit_cast(frag_t, p0); d[64] = __builtin_bit_cast(frag_t, p1); d[128] = __builtin_bit_cast(frag_t, p2);
                if (to_wp) {
                    const unsigned gb = (unsigned)((((size_t)(c0 >> 7) * a.nst_total + 4 * J + q) * B3_GRAN + ((c0 >> 5) & 3) * 64 + lane) * 16u);
                    st16_sc1(p0, rWp, gb); st16_sc1(p1, rWp, gb + 256 * 16); st16_sc1(p2, rWp, gb + 512 * 16);
                }
            });
            cha_store(sf + RLF_W0 + fa, (unsigned)J + 1, lane);
            if (fa == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 2);
            // (5) the consumers' flag: once both halves' stores have drained, in panel order.  This wave has nothing urgent now (the next panel's pair is
            //     another one): it waits for its own stores here
            if (publish) {
                drain_stores();
                unsigned old = 0;
                if (lane == 0) old = atomicAdd(&sf[RLF_DR0 + (J & 3)], 1u);
                if (__builtin_amdgcn_readfirstlane((int)old) == 1) {
                    if (!cha_wait(sf + RLF_PUB, (unsigned)J)) { if (lane == 0) atomicExch(guard, 1); }
                    if (lane == 0) { sf[RLF_DR0 + (J & 3)] = 0u; cf_store(cf_strip(a.cf, s), a.base + (unsigned)J + 1); }
                    cha_store(sf + RLF_PUB, (unsigned)J + 1, lane);
                }
            }
        }
        }
        // ---- W_J into this wave's tiles j > J (ascending: j = J + 1, the next panel's right-hand side, first)
#pragma unroll
        for (int t = 0; t < RL_MAXT; ++t) {
            const int j = par + 4 * t;
            if (j <= J || j >= nrb) continue;
            wave_wait(cf_rowL(a.cf, j), a.base + (unsigned)J + 1, guard);     // L(j, J) is out (j = J + 1: crit's; else row j's)
            if (j == J + 1 && fa == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 6);
            load_A((unsigned)(j * a.sp_stride + J) * B3_SGRAN);                // ... and on its way before W_J is waited for
            lds_wait2(RLF_W0, (unsigned)J + 1);
            const frag_t *wb = WPl + (size_t)(J % win) * CP_WGRAN + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const frag_t b0 = wb[q * 192], b1 = wb[q * 192 + 64], b2 = wb[q * 192 + 128];
                RL_MMA(C[t], fA[q][0], b0); RL_MMA(C[t], fA[q][0], b1); RL_MMA(C[t], fA[q][1], b0); RL_MMA(C[t], fA[q][1], b1); RL_MMA(C[t], fA[q][0], b2); RL_MMA(C[t], fA[q][2], b0);
            }
            if (j == J + 1 && fa == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 7);
        }
    }
#undef RL_MMA
    __syncthreads();
}

__device__ CP_ROLE void strip_body(cp_ka_t ka_v, int nrb_v, int rows_v, int s_v)
{
    const CpArgs a = args_from_kernarg<CpArgs>(ka_v, 0);
    const int nrb = __builtin_amdgcn_readfirstlane(nrb_v), s = __builtin_amdgcn_readfirstlane(s_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    // Eight waves: wave (fa, par) owns row half fa of the 64 x 32 block and every fourth operand block (or k-step) par; the four shares meet
    // through two f32 buffers.  LDS: [win] blocks of W as B-operand planes | CP (the current right-hand side's planes; Yw aliases it) | Yw2.
    // (round 4: the plane blocks are a RING of a.win slots, block K in slot K % win; an update with more panels than slots re-reads the older
    //  blocks from Wp, where the strip has written them anyway -- config 5's 40 panels take the one-launch form; Yw2 has a buffer of its own)
    const int win = a.win;
    frag_t *WPl = reinterpret_cast<frag_t *>(cp_smem);                  // [win][CP_WGRAN]
    frag_t *CP = WPl + (size_t)win * CP_WGRAN;                          // [CP_WGRAN]
    float *Yw = reinterpret_cast<float *>(CP);                          // f32 [64][CP_WS] (8.4 KB), alias of CP: used strictly before / after it
    float *Yw2 = reinterpret_cast<float *>(CP + CP_WGRAN);              // f32 [64][CP_WS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fa = wave & 1, par = wave >> 1;
    const int c0 = s * 32, lcol = lane & 31;
    const bool nu_strip = c0 == a.ld, publish = a.n_dd > 0 || a.xu != 0;
    int32_t *guard = a.status + 1;
    const __amdgpu_buffer_rsrc_t rW = cp_rsrc(a.W), rSp = cp_rsrc(a.Sp), rWp = cp_rsrc(a.Wp);
    const unsigned wvoff = acc_voff(lane, a.ldw) + (unsigned)((32 * fa) * a.ldw + c0) * 4u;
    const unsigned plo = (unsigned)(fa * 64 + lane) * 16u;               // this lane's place in a plane block's row half
    float *yrow = Yw + (32 * fa) * CP_WS + lcol, *yrow2 = Yw2 + (32 * fa) * CP_WS + lcol;
    // the four shares of a 64 x 32 tile -> their sum in the par == 0 waves (three barriers; Yw / Yw2 must be free on entry)
    auto reduce4 = [&](f32x16_t &v) {
        if (par == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) yrow[acc_row(e, lane) * CP_WS] = v[e];
        } else if (par == 3) {
#pragma unroll
            for (int e = 0; e < 16; ++e) yrow2[acc_row(e, lane) * CP_WS] = v[e];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += yrow[acc_row(e, lane) * CP_WS];
        } else if (par == 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += yrow2[acc_row(e, lane) * CP_WS];
        }
        __syncthreads();
        if (par == 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) yrow[acc_row(e, lane) * CP_WS] = v[e];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += yrow[acc_row(e, lane) * CP_WS];
        }
    };
    // f32 [64][32] tile in Yw -> B-operand planes; 256 jobs (q, lane), every thread joins the barrier
    auto tile_to_planes = [&](int J, bool keep, bool to_wp) {
        const int q = (tid >> 6) & 3, l = tid & 63, col = l & 31, h = l >> 5;
        u32x4_t p0, p1, p2;
        if (tid < 256) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = Yw[(16 * q + 8 * h + j) * CP_WS + col];
            b3_split3(x, p0, p1, p2);
        }
        __syncthreads();                                                 // every read of Yw is done: CP (its alias) may be written
        if (tid < 256) {
            if (!keep && !to_wp) {
                CP[q * 192 + l] = __builtin_bit_cast(frag_t, p0); CP[q * 192 + 64 + l] = __builtin_bit_cast(frag_t, p1); CP[q * 192 + 128 + l] = __builtin_bit_cast(frag_t, p2);
            }
            if (keep) {
                frag_t *d = WPl + (size_t)(J % win) * CP_WGRAN + q * 192 + l;
                d[0] = __builtin_bit_cast(frag_t, p0); d[64] = __builtin_bit_cast(frag_t, p1); d[128] = __builtin_bit_cast(frag_t, p2);
            }
            if (to_wp) {
                // k_downdate_b3's image: block (column block of 128, stage of 16 k) = [plane 3][fragment 4][lane 64].  Write-through stores:
                // the down-date consumers of THIS launch read the panel as soon as the strip's flag is up.
                const unsigned gb = (unsigned)((((size_t)(c0 >> 7) * a.nst_total + 4 * J + q) * B3_GRAN + ((c0 >> 5) & 3) * 64 + l) * 16u);
                st16_sc1(p0, rWp, gb); st16_sc1(p1, rWp, gb + 256 * 16); st16_sc1(p2, rWp, gb + 512 * 16);
                if (publish || a.nrb_max - 1 > win) drain_stores();        // (also when the strip itself re-reads these blocks later)
            }
        }
    };
#define ST_MMA(fa_, bb) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa_), __builtin_bit_cast(bf16x8_t, bb), acc, 0, 0, 0)
    // acc of wave (fa, par): this wave's share of  sum_K L(J, K) W_K - HP_J  for the step about to be solved (-HP_J rides with par 0; blocks
    // K = par (mod 4); the newest block K = J-1 is split by k-steps instead).  It is accumulated AHEAD of need: the terms K <= J-2 of step J
    // while the strip waits for M_{J-1}, the last term as soon as W_{J-1} exists -- when M_J arrives only one product and the epilogue are left.
    if (a.tail) gate_prefetch(rows_v, s);                             // (table entries of the rescue gate's first round: see gate_prefetch)
    if (a.strip_rl && !a.tail && nrb <= 4 * RL_MAXT) {
        strip_rl_loop(a, nrb, s, WPl, CP, reinterpret_cast<unsigned *>(Yw2));
    } else {
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = par == 0 ? -ld_f32(rW, wvoff, acc_soff(e, a.ldw)) : 0.f;
    // HP_{J+1} (raw, cold in HBM) is requested one step ahead by the par == 0 waves: nothing later in a step waits for it
    float hp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) hp[e] = (par == 0 && nrb > 1) ? ld_f32(rW, wvoff, (unsigned)(NB * a.ldw) * 4u + acc_soff(e, a.ldw)) : 0.f;
    // fragments of the first block of the next step's bulk sum, L(J+2, par): requested in (4), as soon as row J+2 is known to have published, by
    // write-through-coherent (sc1) loads -- no acquire, nothing to wait for when (2) comes  (round 5: they were plain loads behind an acquire, requested
    // in (2): 1.7 us of every panel)
    frag_t f0[4][3];
    for (int J = 0; J < nrb; ++J) {
        // (1) right-hand side C_J = HP_J - sum: the shares meet in the par == 0 waves, then C_J as B-operand planes (CP)
        reduce4(acc);
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) yrow[acc_row(e, lane) * CP_WS] = -acc[e];
        }
        __syncthreads();
        tile_to_planes(J, false, false);                                 // -> CP
        if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 3);
        // (2) while M_J is on its way: the terms K <= J-1 of step J+1
        const bool more = J + 1 < nrb;
        if (more) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = -hp[e];
            if (J + 2 < nrb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) hp[e] = par == 0 ? ld_f32(rW, wvoff, (unsigned)((J + 2) * NB * a.ldw) * 4u + acc_soff(e, a.ldw)) : 0.f;
            }
            if (J >= 1) {
                // L(J+1, K), K <= J-1, were published during panel J-1 (the previous step's (4) has seen row J+1's flag): sc1 loads
                if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 0);
                // blocks K_u = par + 4 u: the fragments of the next block are requested, k-step by k-step, into the registers the current block
                // has just multiplied from
                const int nb = par <= J - 1 ? (J - 1 - par) / 4 + 1 : 0;
                const unsigned row0 = (unsigned)((J + 1) * a.sp_stride + par) * B3_SGRAN;
                auto loadq = [&](int u, int q) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) f0[q][pl] = ld_granule(rSp, plo, row0 + (unsigned)(4 * u) * B3_SGRAN + q * 384 + pl * 128, 1);
                };
                // (block 0 is on its way since the previous step's (4))
                // blocks written so far: 0 .. J-1, the ring holds the last `win` of them; the older ones (K < J - win) come back from Wp (this
                // strip's own write-through stores, drained steps ago), their fragments requested one block ahead like L's
                const int n_old = J - win > par ? (J - win - par + 3) / 4 : 0;          // blocks K = par + 4 u < J - win
                int u = 0;
                if (n_old > 0) {
                    const unsigned wlane = (unsigned)lane * 16u;
                    const unsigned wbase = (unsigned)(((size_t)(c0 >> 7) * a.nst_total) * B3_GRAN + ((c0 >> 5) & 3) * 64);
                    frag_t w0[4][3];
                    auto loadw = [&](int uu, int q) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) w0[q][pl] = ld_granule(rWp, wlane, wbase + (unsigned)(4 * (par + 4 * uu) + q) * B3_GRAN + pl * 256, 1);
                    };
#pragma unroll
                    for (int q = 0; q < 4; ++q) loadw(0, q);
                    for (; u < n_old; ++u) {
                        const bool nx = u + 1 < nb, nw = u + 1 < n_old;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const frag_t b0 = w0[q][0], b1 = w0[q][1], b2 = w0[q][2];
                            ST_MMA(f0[q][0], b0); ST_MMA(f0[q][0], b1); ST_MMA(f0[q][1], b0); ST_MMA(f0[q][1], b1); ST_MMA(f0[q][0], b2); ST_MMA(f0[q][2], b0);
                            if (nx) loadq(u + 1, q);
                            if (nw) loadw(u + 1, q);
                        }
                    }
                }
                for (; u < nb; ++u) {
                    const frag_t *wb = WPl + (size_t)((par + 4 * u) % win) * CP_WGRAN + lane;
                    const bool nx = u + 1 < nb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const frag_t b0 = wb[q * 192], b1 = wb[q * 192 + 64], b2 = wb[q * 192 + 128];
                        ST_MMA(f0[q][0], b0); ST_MMA(f0[q][0], b1); ST_MMA(f0[q][1], b0); ST_MMA(f0[q][1], b1); ST_MMA(f0[q][0], b2); ST_MMA(f0[q][2], b0);
                        if (nx) loadq(u + 1, q);
                    }
                }
            }
        }
        // (3) W_J = M_J C_J: wave (fa, par) multiplies k-step par; M_J by sc1 loads (one lane polls, barrier, then every wave loads)
        if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 4);
        wg_wait(a.cf + CF_MP, a.base + (unsigned)J + 1, guard);
        if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 1);
        f32x16_t wacc;
        {
            frag_t fM[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fM[pl] = ld_granule(rSp, plo, (unsigned)(J * a.sp_stride + J) * B3_SGRAN + par * 384 + pl * 128, 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) wacc[e] = 0.f;
            const frag_t b0 = CP[par * 192 + lane], b1 = CP[par * 192 + 64 + lane], b2 = CP[par * 192 + 128 + lane];
#define SM_MMA(px, bb) wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fM[px]), __builtin_bit_cast(bf16x8_t, bb), wacc, 0, 0, 0)
            SM_MMA(0, b0); SM_MMA(0, b1); SM_MMA(1, b0); SM_MMA(1, b1); SM_MMA(0, b2); SM_MMA(2, b0);
        }
        __syncthreads();                                                 // CP's reads are done (Yw aliases it)
        reduce4(wacc);
        __syncthreads();
        if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 5);
        if (par == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                yrow[acc_row(e, lane) * CP_WS] = wacc[e];
                // (the strip that owns column ld = L^-1 nu writes through: every strip reads that column for its x-update)
                if (nu_strip) st_f32_sc1(wacc[e], rW, wvoff, (unsigned)(J * NB * a.ldw) * 4u + acc_soff(e, a.ldw));
                else st_f32(wacc[e], rW, wvoff, (unsigned)(J * NB * a.ldw) * 4u + acc_soff(e, a.ldw));
            }
        }
        __syncthreads();
        if (a.tail) {
            // W_J of these columns once more, column-major (four k of one column per thread): the rescue gate reads whole columns of W (y = H J W'),
            // which in the row-major image are one 128-byte line per entry.  Drained with everything else in front of the strip's x flag.
            const int col = tid & 31, kq = tid >> 5;
            const f4v_t v = { Yw[(4 * kq) * CP_WS + col], Yw[(4 * kq + 1) * CP_WS + col], Yw[(4 * kq + 2) * CP_WS + col], Yw[(4 * kq + 3) * CP_WS + col] };
            st16_sc1(__builtin_bit_cast(u32x4_t, v), cp_rsrc(a.Wt), (unsigned)((c0 + col) * a.kcap + J * NB + 4 * kq) * 4u);
        }
        tile_to_planes(J, more || a.tail != 0, c0 < a.ld + NB);          // (tail: the HI panel's right-hand side needs every block of W)
        __syncthreads();
        // W_J's planes of these 32 columns are out (every storing thread has drained): the consumers' flag
        if (tid == 0 && publish) cf_store(cf_strip(a.cf, s), a.base + (unsigned)J + 1);
        if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 2);
        // (4) the newest term of step J+1: L(J+1, J) W_J, one k-step per wave; L(J+1, J) is published about now (sc1 loads)
        if (more) {
            // the next step's (2) reads L(J+2, K), K <= J: once row J+2 has published them, the first block's fragments are requested (sc1 loads)
            if (wave == 7) {
                // rows J+1 and (if it exists) J+2 have published their L(., J): one poll loop for both flags (lane 0 / lane 1)
                wave_wait_rows(a.cf, J + 1, J + 2 < nrb ? 2 : 1, a.base + (unsigned)J + 1, lane, guard);
            }
            __syncthreads();
            if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 6);
            if (J + 2 < nrb && par <= J) {
                const unsigned row0n = (unsigned)((J + 2) * a.sp_stride + par) * B3_SGRAN;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) f0[q][pl] = ld_granule(rSp, plo, row0n + q * 384 + pl * 128, 1);
            }
            const frag_t *wb = WPl + (size_t)(J % win) * CP_WGRAN + lane;
            frag_t fL[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fL[pl] = ld_granule(rSp, plo, (unsigned)((J + 1) * a.sp_stride + J) * B3_SGRAN + par * 384 + pl * 128, 1);
            const frag_t b0 = wb[par * 192], b1 = wb[par * 192 + 64], b2 = wb[par * 192 + 128];
            ST_MMA(fL[0], b0); ST_MMA(fL[0], b1); ST_MMA(fL[1], b0); ST_MMA(fL[1], b1); ST_MMA(fL[0], b2); ST_MMA(fL[2], b0);
            if (tid == 0 && (s == 0 || s == a.n_strips - 1)) CP_STAMP(s == 0 ? 16 : 17, J, 7);
        }
    }
#undef ST_MMA
#undef SM_MMA
    }
    // ---- update.m:36,42,48 for this strip's 32 states: x_out = x_prior + W'(L^-1 nu), the normalisation Jacobian at the un-normalised quaternion
    //      -> params, the quaternion normalised.  Sixteen chains (chain g: rows a = g mod 16, in order), summed 0 .. 15, x_prior last -- term for
    //      term the sums of update_x_block / k_update_x, so x_k_k is the same bits whichever of them ran.  W is this strip's own (its stores have
    //      drained); c = L^-1 nu is the nu strip's column, written through and read here behind that strip's last flag.
    // (scratch of the x-updates: the plane slots when they are dead; with the tail they still feed the HI panel, and CP | Yw2 take it)
    unsigned char *xs_base = a.tail ? reinterpret_cast<unsigned char *>(CP) : cp_smem;
    if (a.xu && c0 < a.ld) {
        if (s == 1 && tid == 0) CP_STAMP(23, 2, 0);
        const int rows = __builtin_amdgcn_readfirstlane(rows_v);
        float *cs = reinterpret_cast<float *>(xs_base);                              // [nrb * 64]
        double *red = reinterpret_cast<double *>(xs_base + 4096);                    // [16][32]
        double *qs = red + 16 * 32;
        __syncthreads();                                                             // (the plane slots are dead)
        wg_wait(cf_strip(a.cf, a.ld / 32), a.base + (unsigned)nrb, guard);
        for (int k = tid; k < rows; k += 512) cs[k] = ld_f32_sc1(rW, (unsigned)(k * a.ldw + a.ld) * 4u, 0);
        __syncthreads();
        const int ci = tid & 31, rg = tid >> 5, i = c0 + ci;
        const float *Wc = a.W + i;
        double sx = 0;
#pragma unroll 8
        for (int k = rg; k < rows; k += 16) sx = fma((double)Wc[(size_t)k * a.ldw], (double)cs[k], sx);
        red[rg * 32 + ci] = sx;
        __syncthreads();
        sx = 0;
        if (rg == 0) {
#pragma unroll
            for (int g = 0; g < 16; ++g) sx += red[g * 32 + ci];
            if (i < a.n) sx += a.x_prior[i];
        }
        if (s == 0) {                                                                // (strip-uniform: every thread reaches the barrier)
            if (rg == 0 && i >= 3 && i < 7) qs[i - 3] = sx;
            __syncthreads();
            if (rg == 0 && i == 0) { double Jn[16]; d_normjac(qs, Jn); for (int t = 0; t < 16; ++t) { st_f64_sc1(a.params + 16 + t, Jn[t]); st_f64_sc1(a.params + 96 + t, Jn[t]); } }
            if (rg == 0 && i >= 3 && i < 7) sx = sx / sqrt(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
        }
        if (rg == 0 && i < a.n) st_f64_sc1(a.x_out + i, sx);
        if (a.tail || a.proj) {
            // x_k_k of these 32 states (and, from strip 0, the normalisation Jacobian) is out: the gate's waves wait for this value
            drain_stores();
            __syncthreads();
            if (tid == 0) cf_store(cf_strip(a.cf, s), a.base + CFV_X);
            if (s == 1 && tid == 0) CP_STAMP(23, 2, 1);
        }
    }
    if (a.tail) strip_tail_body(nrb, s, rows_v);
    else if (a.proj) strip_proj_body(s);
}

// ------------------------------------------------------------------------------------------------------------------------------
// strip s behind its x-update, when the launch carries the tail (CpTail): (B) its landmarks are projected and gated; (C) crit's list; (D) the rescued
// landmarks' rows as panel nrb: W~ = M_hi ((H J) P - Y_hi W) for its 32 columns, then the HI update's x.  Out of line and with the launch arguments
// from LDS, so that nothing of it is live in the panel loop of strip_body.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) void strip_tail_body(int nrb_v, int s_v, int rows_v)
{
    const CpArgs a = args_from_lds<CpArgs>(CP_TA_OFF);
    const int nrb = __builtin_amdgcn_readfirstlane(nrb_v), s = __builtin_amdgcn_readfirstlane(s_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    const int win = a.win;
    frag_t *WPl = reinterpret_cast<frag_t *>(cp_smem);
    frag_t *CP = WPl + (size_t)win * CP_WGRAN;
    float *Yw = reinterpret_cast<float *>(CP);
    float *Yw2 = reinterpret_cast<float *>(CP + CP_WGRAN);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fa = wave & 1, par = wave >> 1;
    const int c0 = s * 32, lcol = lane & 31;
    const bool nu_strip = c0 == a.ld;
    int32_t *guard = a.status + 1;
    const __amdgpu_buffer_rsrc_t rW = cp_rsrc(a.W), rSp = cp_rsrc(a.Sp), rWp = cp_rsrc(a.Wp);
    const unsigned wvoff = acc_voff(lane, a.ldw) + (unsigned)((32 * fa) * a.ldw + c0) * 4u;
    const unsigned plo = (unsigned)(fa * 64 + lane) * 16u;
    float *yrow = Yw + (32 * fa) * CP_WS + lcol, *yrow2 = Yw2 + (32 * fa) * CP_WS + lcol;
    unsigned char *xs_base = reinterpret_cast<unsigned char *>(CP);
    f32x16_t acc;
    // (reduce4 / tile_to_planes: as in strip_body)
    auto reduce4 = [&](f32x16_t &v) {
        if (par == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) yrow[acc_row(e, lane) * CP_WS] = v[e];
        } else if (par == 3) {
#pragma unroll
            for (int e = 0; e < 16; ++e) yrow2[acc_row(e, lane) * CP_WS] = v[e];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += yrow[acc_row(e, lane) * CP_WS];
        } else if (par == 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += yrow2[acc_row(e, lane) * CP_WS];
        }
        __syncthreads();
        if (par == 2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) yrow[acc_row(e, lane) * CP_WS] = v[e];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += yrow[acc_row(e, lane) * CP_WS];
        }
    };
    auto tile_to_planes = [&](int J, bool, bool to_wp) {
        const int q = (tid >> 6) & 3, l = tid & 63, col = l & 31, h = l >> 5;
        u32x4_t p0, p1, p2;
        if (tid < 256) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = Yw[(16 * q + 8 * h + j) * CP_WS + col];
            b3_split3(x, p0, p1, p2);
        }
        __syncthreads();
        if (tid < 256) {
            if (!to_wp) {
                CP[q * 192 + l] = __builtin_bit_cast(frag_t, p0); CP[q * 192 + 64 + l] = __builtin_bit_cast(frag_t, p1); CP[q * 192 + 128 + l] = __builtin_bit_cast(frag_t, p2);
            } else {
                const unsigned gb = (unsigned)((((size_t)(c0 >> 7) * a.nst_total + 4 * J + q) * B3_GRAN + ((c0 >> 5) & 3) * 64 + l) * 16u);
                st16_sc1(p0, rWp, gb); st16_sc1(p1, rWp, gb + 256 * 16); st16_sc1(p2, rWp, gb + 512 * 16);
                drain_stores();
            }
        }
    };
    const CpTail t = args_from_lds<CpTail>(CP_T_OFF);
    if (s == 1 && tid == 0) CP_STAMP(23, 0, 0);
    tail_gate_body(a, t, nrb, s, __builtin_amdgcn_readfirstlane(rows_v));      // (stores the strip's gate flag)
    __syncthreads();
    if (s == 1 && tid == 0) CP_STAMP(23, 0, 1);
    wg_wait(a.cf + CF_HI, a.base + CFV_HIL, guard);
    if (s == 1 && tid == 0) CP_STAMP(23, 0, 2);
    const int r_hi = __builtin_amdgcn_readfirstlane(ld_i32_sc1(reinterpret_cast<const int32_t *>(a.cf + CF_HI + 1)));
    if (r_hi == 0) return;                                       // (nothing rescued, or too much for one panel: params[96..] holds the LI update's Jacobian)
    const int J = nrb;
    TailSmem &ts = *reinterpret_cast<TailSmem *>(strip_tail_smem(win));
    {
        const __amdgpu_buffer_rsrc_t rHib = cp_rsrc(t.hib);
        if (tid < NB * 4) *reinterpret_cast<u32x4_t *>(&ts.rv[tid >> 2][4 * (tid & 3)]) = ld16_sc1(rHib, (unsigned)(64 + (tid >> 2) * 16 + 4 * (tid & 3)) * 4u);
        else if (tid < NB * 4 + HT_MAXL) {
            const int l = tid - NB * 4;
            int lm = t.N, off = 0, d = 0;
            if (2 * l < r_hi) { lm = ld_i32_sc1(t.hib + 4 + l); off = t.lm_off[lm]; d = t.lm_type[lm] == PRE3_INVDEPTH ? 6 : 3; }
            ts.lm[l] = lm; ts.off[l] = off; ts.d[l] = d;
        }
    }
    __syncthreads();
    // (D1) this wave's share of  sum_K Y_hi(:, K) W_K - (H J P)(:, these columns):  the 13 terms of the raw row are dealt over the four par groups,
    //      the blocks of W as in the panels.  The nu column is the new innovation z - h(x_k_k) itself: nothing of the LI update's is subtracted.
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = 32 * fa + acc_row(e, lane);
        float v = 0.f;
        if (nu_strip) { if (par == 0 && lcol == 0) v = ts.rv[row][13]; }
        else if (c0 + lcol < a.ld) {
            const float *rv = ts.rv[row];
            const int lo = ts.off[row >> 1];
            for (int u = par; u < 13; u += 4) v = fmaf(rv[u], a.P[(size_t)(u < 7 ? u : lo + u - 7) * a.ld + c0 + lcol], v);
        }
        acc[e] = -v;
    }
#define ST_MMA(fa_, bb) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa_), __builtin_bit_cast(bf16x8_t, bb), acc, 0, 0, 0)
    if (c0 < a.ld) {
        const int ra = 32 * fa + (lane & 31);
        const __amdgpu_buffer_rsrc_t rY = cp_rsrc(t.Yp);
        const unsigned va = yp_voff(ra < r_hi ? ts.lm[ra >> 1] : t.N, ra & 1, lane >> 5, t.ykcap);
        const unsigned wlane = (unsigned)lane * 16u;
        const unsigned wbase = (unsigned)(((size_t)(c0 >> 7) * a.nst_total) * B3_GRAN + ((c0 >> 5) & 3) * 64);
        for (int K = par; K < nrb; K += 4) {
            frag_t f0[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) f0[q][pl] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rY, va, yp_soff(pl, K, q, t.ykcap), 16));
            if (K < nrb - win) {                                 // (left the ring: this strip's own write-through planes)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const frag_t b0 = ld_granule(rWp, wlane, wbase + (unsigned)(4 * K + q) * B3_GRAN, 1), b1 = ld_granule(rWp, wlane, wbase + (unsigned)(4 * K + q) * B3_GRAN + 256, 1),
                                 b2 = ld_granule(rWp, wlane, wbase + (unsigned)(4 * K + q) * B3_GRAN + 512, 1);
                    ST_MMA(f0[q][0], b0); ST_MMA(f0[q][0], b1); ST_MMA(f0[q][1], b0); ST_MMA(f0[q][1], b1); ST_MMA(f0[q][0], b2); ST_MMA(f0[q][2], b0);
                }
            } else {
                const frag_t *wb = WPl + (size_t)(K % win) * CP_WGRAN + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const frag_t b0 = wb[q * 192], b1 = wb[q * 192 + 64], b2 = wb[q * 192 + 128];
                    ST_MMA(f0[q][0], b0); ST_MMA(f0[q][0], b1); ST_MMA(f0[q][1], b0); ST_MMA(f0[q][1], b1); ST_MMA(f0[q][0], b2); ST_MMA(f0[q][2], b0);
                }
            }
        }
    }
#undef ST_MMA
    // (D2) C_hi as planes, then W~ = M_hi C_hi once crit has published M_hi (the panel loop's (1) and (3))
    if (s == 1 && tid == 0) CP_STAMP(23, 0, 3);
    reduce4(acc);
    __syncthreads();
    if (par == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) yrow[acc_row(e, lane) * CP_WS] = -acc[e];
    }
    __syncthreads();
    tile_to_planes(J, false, false);
    wg_wait(a.cf + CF_MP, a.base + (unsigned)J + 1, guard);
    if (s == 1 && tid == 0) CP_STAMP(23, 0, 4);
    f32x16_t wacc;
    {
        frag_t fM[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fM[pl] = ld_granule(rSp, plo, (unsigned)(J * a.sp_stride + J) * B3_SGRAN + par * 384 + pl * 128, 1);
#pragma unroll
        for (int e = 0; e < 16; ++e) wacc[e] = 0.f;
        const frag_t b0 = CP[par * 192 + lane], b1 = CP[par * 192 + 64 + lane], b2 = CP[par * 192 + 128 + lane];
#define SM_MMA(px, bb) wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fM[px]), __builtin_bit_cast(bf16x8_t, bb), wacc, 0, 0, 0)
        SM_MMA(0, b0); SM_MMA(0, b1); SM_MMA(1, b0); SM_MMA(1, b1); SM_MMA(0, b2); SM_MMA(2, b0);
#undef SM_MMA
    }
    __syncthreads();
    reduce4(wacc);
    __syncthreads();
    if (par == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            yrow[acc_row(e, lane) * CP_WS] = wacc[e];
            if (nu_strip) st_f32_sc1(wacc[e], rW, wvoff, (unsigned)(J * NB * a.ldw) * 4u + acc_soff(e, a.ldw));
            else st_f32(wacc[e], rW, wvoff, (unsigned)(J * NB * a.ldw) * 4u + acc_soff(e, a.ldw));
        }
    }
    __syncthreads();
    tile_to_planes(J, false, c0 < a.ld + NB);
    __syncthreads();
    if (tid == 0) cf_store(cf_strip(a.cf, s), a.base + CFV_HIW);   // the consumers' panel nrb
    if (s == 1 && tid == 0) CP_STAMP(23, 0, 5);
    // (D3) update.m:36,42,48 of the HI update for these 32 states: x += J W~'(L_hi^-1 nu) (W~ lacks the J' of the LI update's normalisation on its
    //      quaternion columns: the down-date carries it as a pending pass, the state takes it here); then this update's own Jacobian J2, and what is
    //      left of update.m:42-46 for both updates is rows / columns 3..6 <- (J2 J) . : params[96..] (and [16..] for k_jnorm_P)
    if (c0 < a.ld) {
        float *cs = reinterpret_cast<float *>(xs_base);
        double *red = reinterpret_cast<double *>(xs_base + 4096);
        double *qs = red + 16 * 32;
        __syncthreads();
        wg_wait(cf_strip(a.cf, a.ld / 32), a.base + CFV_HIW, guard);
        if (tid < NB) cs[tid] = ld_f32_sc1(rW, (unsigned)((J * NB + tid) * a.ldw + a.ld) * 4u, 0);
        __syncthreads();
        const int ci = tid & 31, rg = tid >> 5, i = c0 + ci;
        const float *Wc = a.W + (size_t)(J * NB) * a.ldw + i;
        double sx = 0;
#pragma unroll
        for (int k = 0; k < NB / 16; ++k) sx = fma((double)Wc[(size_t)(rg + 16 * k) * a.ldw], (double)cs[rg + 16 * k], sx);
        red[rg * 32 + ci] = sx;
        __syncthreads();
        sx = 0;
        if (rg == 0) {
#pragma unroll
            for (int g = 0; g < 16; ++g) sx += red[g * 32 + ci];
        }
        const bool quat = rg == 0 && i >= 3 && i < 7;
        double J1[16];
        if (s == 0) {
            if (quat) qs[i - 3] = sx;
            __syncthreads();
            if (rg == 0 && (quat || i == 0)) {
#pragma unroll
                for (int u = 0; u < 16; ++u) J1[u] = ld_f64_sc1(a.params + 16 + u);
            }
            if (quat) sx = fma(J1[(i - 3) * 4 + 3], qs[3], fma(J1[(i - 3) * 4 + 2], qs[2], fma(J1[(i - 3) * 4 + 1], qs[1], J1[(i - 3) * 4] * qs[0])));
            __syncthreads();
        }
        double xv = 0;
        if (rg == 0 && i < a.n) xv = ld_f64_sc1(a.x_out + i) + sx;
        if (s == 0) {
            if (quat) qs[i - 3] = xv;
            __syncthreads();
            if (rg == 0 && i == 0) {
                double J2[16];
                d_normjac(qs, J2);
                for (int r1 = 0; r1 < 4; ++r1)
                    for (int c1 = 0; c1 < 4; ++c1) {
                        const double v = fma(J2[r1 * 4 + 3], J1[12 + c1], fma(J2[r1 * 4 + 2], J1[8 + c1], fma(J2[r1 * 4 + 1], J1[4 + c1], J2[r1 * 4] * J1[c1])));
                        a.params[96 + r1 * 4 + c1] = v; a.params[16 + r1 * 4 + c1] = v;
                    }
            }
            if (quat) xv = xv / sqrt(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
        }
        if (rg == 0 && i < a.n) a.x_out[i] = xv;
        if (s == 1 && tid == 0) CP_STAMP(23, 0, 6);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// down-date consumer g: up to twelve 64 x 64 tiles of P's upper triangle (update.m:37-38), one per wave
// ------------------------------------------------------------------------------------------------------------------------------
// Group record (DG_WORDS int32, built by dd_build_groups): [0] = tasks | slots << 8; [1..6] = the 64-column block of W each LDS slot holds;
// [7..9] = one byte per task: slot of the tile's row block | slot of its column block << 4.  LDS: [stage 4][slot 6][plane 3][fragment 2][lane 64]
// granules = 144 KB -- one whole panel of the group's operands; the epilogue's wave-private patches alias it.
template <int N> __device__ __forceinline__ void dd_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n of this wave's vector-memory operations are outstanding (a smaller immediate than n is only stricter)
__device__ __forceinline__ void dd_vmwait_le(int n)
{
    switch (n) {
    case 0: dd_vmwait<0>(); break;   case 1: dd_vmwait<1>(); break;   case 2: dd_vmwait<2>(); break;   case 3: dd_vmwait<3>(); break;
    case 4: dd_vmwait<4>(); break;   case 5: dd_vmwait<5>(); break;   case 6: dd_vmwait<6>(); break;   case 7: dd_vmwait<6>(); break;
    case 8: dd_vmwait<8>(); break;   case 9: dd_vmwait<9>(); break;   case 10: dd_vmwait<10>(); break; case 11: dd_vmwait<10>(); break;
    case 12: dd_vmwait<12>(); break; case 13: dd_vmwait<12>(); break; case 14: dd_vmwait<12>(); break; default: dd_vmwait<15>(); break;
    }
}

// The consumers' stores of P are write-through (sc0 sc1): the 37 MB then drain to memory while the launch still runs.  As plain stores they
// sat dirty in the eight L2s until the end-of-kernel release wrote them back -- a 6-us gap between this launch and the next one
// (tools/ab_env2.sh PRE3_DD_MODE 1 17: +1.8 % steps/s).
__device__ __forceinline__ void dd_store_wt(float *d, float v, bool wt)
{
    if (wt) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    else *d = v;
}
__device__ __forceinline__ void dd_store_wt(f4v_t *d, f4v_t v, bool wt)
{
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    else *d = v;
}

__device__ CP_ROLE void dd_body(cp_ka_t ka_v, int nrb_v, int rows_v, int g_v)
{
    const CpArgs a = args_from_kernarg<CpArgs>(ka_v, 0);
    const int nrb = __builtin_amdgcn_readfirstlane(nrb_v), rows = __builtin_amdgcn_readfirstlane(rows_v), g = __builtin_amdgcn_readfirstlane(g_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    frag_t *ops = reinterpret_cast<frag_t *>(cp_smem);
    const int32_t *rec = a.dd + (size_t)g * DG_WORDS;
    // (everything read from the record is made scalar explicitly: left to itself the compiler kept the slot table in vector registers and
    //  put a load + s_waitcnt vmcnt(0) between two LDS-DMA instructions -- 9.9 us per panel instead of 3)
    auto U = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int hdr = U(rec[0]), ntasks = hdr & 0xff, nslots = (hdr >> 8) & 0xff;
    const int tid = threadIdx.x, wave = U(tid >> 6), lane = tid & 63;
    if (wave >= ntasks) return;                                  // (an ended wave is not waited for by the barriers below)
    const int tb = (U(rec[7 + (wave >> 2)]) >> (8 * (wave & 3))) & 0xff, sa = tb & 15, sb = tb >> 4;
    const int bi = U(rec[1 + sa]), bj = U(rec[1 + sb]);
    const bool diag = bi == bj;
    int32_t *guard = a.status + 1;
    const int nst_real = (rows + B3_BK - 1) / B3_BK;              // k-stages that hold real rows (the rest of the last panel is zero padding)
    // This wave's share of a stage's LDS-DMA instructions: idx = wave, wave + ntasks, .. < 6 nslots; idx -> (slot, plane, fragment).  The
    // descriptors (source / LDS offsets in granules) are worked out once and live in scalar registers: the issue loop has no load in it.
    const int n_inst = 6 * nslots, c_w = U((n_inst - wave + ntasks - 1) / ntasks);
    unsigned src_off[DG_KMAX]; int lds_off[DG_KMAX];
#pragma unroll
    for (int k = 0; k < DG_KMAX; ++k) {
        const int idx = wave + k * ntasks, idc = idx < n_inst ? idx : 0;
        const int slot = U(idc / 6), pf = idc - 6 * slot, pl = pf >> 1, f = pf & 1;
        const int blk = U(rec[1 + slot]);
        src_off[k] = (unsigned)U((blk >> 1) * a.nst_total * B3_GRAN + pl * 256 + (2 * (blk & 1) + f) * 64);
        lds_off[k] = U(slot * DG_SLOT_GRAN + (2 * pl + f) * 64);
    }
    // wave 0, lane l < 2 nslots: the flag of strip 2 blk + (l & 1) of slot l >> 1
    const unsigned *fp = a.cf;
    if (wave == 0) fp = cf_strip(a.cf, 2 * rec[1 + ((lane < 2 * nslots ? lane : 0) >> 1)] + (lane & 1));
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const frag_t *Wp = static_cast<const frag_t *>(a.Wp), *Wpp = static_cast<const frag_t *>(a.Wp_pend);
    const int j_last = (nst_real + 3) / 4 - 1;                  // the last panel that holds real rows
    const int npp = (a.pend_ns + 3) / 4;                        // panels of a pending HI update (J = -npp .. -1: complete in memory, no flag to wait for)
    for (int J = -npp; J <= nrb; ++J) {
        int ns = J < 0 ? a.pend_ns - 4 * (J + npp) : nst_real - 4 * J;
        ns = ns > 4 ? 4 : ns;
        if (J < nrb) { if (ns <= 0) continue; }
        else {
            // panel nrb: the rescued landmarks' rows, if the tail brings any (crit's word says how many once the gate has run)
            if (!a.tail) break;
            int r_hi = 0;
            if (wave == 0) {
                if (lane == 0) cf_wait(a.cf + CF_HI, a.base + CFV_HIL, guard);
                r_hi = ld_i32_sc1(reinterpret_cast<const int32_t *>(a.cf + CF_HI + 1));
                if (lane == 0) *reinterpret_cast<volatile int *>(ops + 4 * DG_SLOTS * DG_SLOT_GRAN) = r_hi;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            r_hi = U(*reinterpret_cast<volatile int *>(ops + 4 * DG_SLOTS * DG_SLOT_GRAN));
            ns = (r_hi + B3_BK - 1) / B3_BK;
            if (ns <= 0) break;
        }
        if (J == j_last && (a.dd_mode & 4)) {
            // While the strips finish the last panel: this wave's tile of P is pulled towards the XCD's L2 (sixteen LDS-DMA requests into a 1 KB
            // scratch line behind the operand slots, contents ignored), so that the epilogue's read of P -- all consumers at once, behind the last
            // MFMA -- is served by the L2 instead of HBM.
            frag_t *sink = ops + 4 * DG_SLOTS * DG_SLOT_GRAN + wave * 64;
            const float *pt = a.P + (size_t)(bi * 64 + (lane >> 4)) * a.ld + bj * 64 + (lane & 15) * 4;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                __builtin_amdgcn_global_load_lds(pt + (size_t)(4 * t) * a.ld, (__attribute__((address_space(3))) void *)sink, 16, 0, 0);
        }
        // the strips that own the group's column blocks (two per 64-column block) have published W_J
        if (wave == 0 && J >= 0) {
            const int nfl = 2 * nslots;
            bool gave_up = true;
            for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
                const bool ok = lane >= nfl || cf_reached(cf_load(fp), a.base + (J < nrb ? (unsigned)J + 1 : CFV_HIW));
                if (__all(ok)) { gave_up = false; break; }
                if ((spin & 1023) == 1023 && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { gave_up = false; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (gave_up && lane == 0) atomicExch(guard, 1);
            // ONE agent-scope acquire for the workgroup, its invalidate complete before anybody loads: the panel then comes in by PLAIN LDS-DMA
            // loads, served by the XCD's L2 (every block of W_J is wanted by some twenty groups).  With sc1 loads instead every group fetched
            // its 144 KB from the fabric: 9.9 us per panel instead of 3 (measured, tools/probe_cholp.py).
            if (a.dd_mode & 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // (this wave's reads of the previous panel are in registers)
        __builtin_amdgcn_s_barrier();
        if (tid == 0 && (g == 0 || g == a.n_dd - 1)) CP_STAMP(g == 0 ? 20 : 21, J, 0);
        for (int st = 0; st < ns; ++st) {
            const frag_t *src = (J < 0 ? Wpp + (size_t)(4 * (J + npp) + st) * B3_GRAN : Wp + (size_t)(4 * J + st) * B3_GRAN) + lane;
            frag_t *dst = ops + st * (DG_SLOTS * DG_SLOT_GRAN);
#pragma unroll
            for (int k = 0; k < DG_KMAX; ++k)
                if (k < c_w) {
                    if (a.dd_mode & 1) __builtin_amdgcn_global_load_lds(src + src_off[k], (__attribute__((address_space(3))) void *)(dst + lds_off[k]), 16, 0, 16);
                    else __builtin_amdgcn_global_load_lds(src + src_off[k], (__attribute__((address_space(3))) void *)(dst + lds_off[k]), 16, 0, 0);
                }
        }
        for (int st = 0; st < ns; ++st) {
            dd_vmwait_le(c_w * (ns - 1 - st));
            __builtin_amdgcn_s_barrier();                        // every wave's share of stage st has landed
            const frag_t *sA = ops + (st * DG_SLOTS + sa) * DG_SLOT_GRAN + lane, *sB = ops + (st * DG_SLOTS + sb) * DG_SLOT_GRAN + lane;
            frag_t A[3][2], B[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) { A[pl][i] = sA[(2 * pl + i) * 64]; B[pl][i] = sB[(2 * pl + i) * 64]; }
#define DD_MMA(pa, pb) \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[pa][i]), __builtin_bit_cast(bf16x8_t, B[pb][j]), acc[i][j], 0, 0, 0)
            // (the six products of k_downdate_b3, in its order)
            DD_MMA(0, 0); DD_MMA(0, 1); DD_MMA(1, 0); DD_MMA(1, 1); DD_MMA(0, 2); DD_MMA(2, 0);
#undef DD_MMA
        }
        if (tid == 0 && (g == 0 || g == a.n_dd - 1)) CP_STAMP(g == 0 ? 20 : 21, J, 1);
    }
    // ---- P tile <- P tile - acc, and its mirror image (as k_downdate_b3's epilogue; blocks below the diagonal of a diagonal tile are skipped)
    if (tid == 0 && (g == 0 || g == a.n_dd - 1)) CP_STAMP(g == 0 ? 20 : 21, 15, 0);
    float *P = a.P;
    const bool wt = (a.dd_mode & 16) != 0;
    const int ld = a.ld, R0 = bi * 64, C0 = bj * 64;
    const int lrow = 4 * (lane >> 5), lcol = lane & 31;
    float pv[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (!(diag && i > j)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) pv[i][j][e] = P[(size_t)(R0 + i * 32 + (e & 3) + 8 * (e >> 2) + lrow) * ld + C0 + j * 32 + lcol];
            }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                                // the patches alias the operand slots
    float (*patch)[36] = reinterpret_cast<float (*)[36]>(reinterpret_cast<float *>(cp_smem) + wave * (2 * 32 * 36));
    float (*patchT)[36] = patch + 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (diag && i > j) continue;
            const bool dblk = diag && i == j;
            const int r0 = R0 + i * 32, c0 = C0 + j * 32;
            if (dblk) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = (e & 3) + 8 * (e >> 2) + lrow;
                    const float v = pv[i][j][e] - acc[i][j][e];
                    if (lr <= lcol) dd_store_wt(P + (size_t)(r0 + lr) * ld + c0 + lcol, v, wt);
                    patch[lr][lcol] = v;
                }
                wave_lds_sync();
                if (a.jn_q != nullptr && r0 == 0) {
                    // rows 3..6 as the Jnorm pass will read them from memory: below the diagonal that is the mirror image stored further down
                    const int ra = 3 + (lane >> 5), rb = 5 + (lane >> 5);
                    dd_store_wt(a.jn_q + (size_t)(ra - 3) * ld + c0 + lcol, lcol >= ra ? patch[ra][lcol] : patch[lcol][ra], wt);
                    dd_store_wt(a.jn_q + (size_t)(rb - 3) * ld + c0 + lcol, lcol >= rb ? patch[rb][lcol] : patch[lcol][rb], wt);
                }
                const int rr = lane & 31, half = lane >> 5;
#pragma unroll
                for (int cc = 0; cc < 32; cc += 2) {
                    const int c = cc + half;
                    if (rr < c) dd_store_wt(P + (size_t)(c0 + c) * ld + r0 + rr, patch[rr][c], wt);
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f4v_t v;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { v[t] = pv[i][j][4 * g4 + t] - acc[i][j][4 * g4 + t]; patch[8 * g4 + lrow + t][lcol] = v[t]; }
                    *reinterpret_cast<f4v_t *>(&patchT[lcol][8 * g4 + lrow]) = v;
                }
                wave_lds_sync();
                if (a.jn_q != nullptr && r0 == 0) dd_store_wt(a.jn_q + (size_t)(lane >> 5) * ld + c0 + lcol, patch[3 + (lane >> 5)][lcol], wt), dd_store_wt(a.jn_q + (size_t)(2 + (lane >> 5)) * ld + c0 + lcol, patch[5 + (lane >> 5)][lcol], wt);
                const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = rr + 8 * it;
                    f4v_t *d0 = reinterpret_cast<f4v_t *>(P + (size_t)(r0 + row) * ld + c0 + c4), *d1 = reinterpret_cast<f4v_t *>(P + (size_t)(c0 + row) * ld + r0 + c4);
                    const f4v_t v0 = *reinterpret_cast<const f4v_t *>(&patch[row][c4]), v1 = *reinterpret_cast<const f4v_t *>(&patchT[row][c4]);
                    dd_store_wt(d0, v0, wt); dd_store_wt(d1, v1, wt);
                }
            }
            wave_lds_sync();
        }
    if (tid == 0 && (g == 0 || g == a.n_dd - 1)) CP_STAMP(g == 0 ? 20 : 21, 15, 1);
}

__global__ __launch_bounds__(CP_NTH) void k_cholp(CpArgs a, CpTail t)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    int nrb = a.nrb, rows = a.rows;
    if (a.n_dev != nullptr) {                                   // the row count is still on its way to the host (LI update of a step)
        const int n = *a.n_dev;
        rows = 2 * n;
        nrb = (rows + NB - 1) / NB;
        if (nrb > a.nrb_max) nrb = a.nrb_max;
    }
    if (rows > nrb * NB) rows = nrb * NB;
    if (nrb <= 0 && a.pend_ns == 0) return;
    const bool only_pend = nrb <= 0;                             // no LI rows on the device: the consumers still owe P the pending HI down-date
    // Blocks 0, 8, 16, .. 8 nH are crit and the rows: blocks are dealt round-robin over the eight XCDs, so these share one XCD's L2 -- the rows'
    // bulk operands (each L(k, J) is wanted by every row below k) and the hand-offs with crit are then served by that L2.  Placement is a
    // speed assumption only: every hand-off is valid for any placement.  Every other block is a strip.
    const int b = blockIdx.x, nH = a.nrb_max > 2 ? a.nrb_max - 2 : 0, stride = a.stride;
    if (a.tail || a.proj) {
        // the tail's code (crit, strips) takes the launch arguments from LDS; it reads them behind many barriers of its own role.  The copy is made
        // from the kernel-argument segment, not from &a / &t: a dynamically indexed read of a by-value struct makes the compiler keep the struct in
        // scratch, stored by EVERY thread at the kernel's entry -- 240 B x 768 threads x 256 workgroups = 47 MB of write-back per launch (round 6,
        // tools/pmc_env_write.sh: the "unexplained" writes of the LI launch)
        typedef __attribute__((address_space(4))) const unsigned kw_t;
        kw_t *ka = (kw_t *)__builtin_amdgcn_kernarg_segment_ptr();
        if (threadIdx.x < sizeof(CpTail) / 4) reinterpret_cast<unsigned *>(cp_smem + CP_T_OFF)[threadIdx.x] = ka[CP_KA_TAIL / 4 + threadIdx.x];
        if (threadIdx.x >= 128 && threadIdx.x < 128 + sizeof(CpArgs) / 4) reinterpret_cast<unsigned *>(cp_smem + CP_TA_OFF)[threadIdx.x - 128] = ka[threadIdx.x - 128];
    }
#ifndef CP_TEST_ROLE
#define CP_TEST_ROLE 15
#endif
    if (b % stride == 0 && b / stride <= nH) {
        if (only_pend) return;
        const int r = b / stride;
        if (r == 0) { if (CP_TEST_ROLE & 1) crit_body(a, nrb, rows, cp_smem); return; }
        const int i = r + 1;
        if ((CP_TEST_ROLE & 2) && i < nrb) row_body(a, nrb, i);
        return;
    }
    const int sidx = b - (b / stride < nH ? b / stride + 1 : nH + 1);
    if (sidx < a.n_strips) {
        if (threadIdx.x >= 512 || only_pend) return;            // strips are eight waves
        // every strip's x-update waits for the strip that owns column ld (L^-1 nu): it takes the lowest block index of the strips, so that the one
        // workgroup all the others wait for is dispatched in front of them
        const int s_nu = a.ld / 32, s_col = sidx == 0 ? s_nu : sidx == s_nu ? 0 : sidx;
        if (CP_TEST_ROLE & 4) strip_body((cp_ka_t)__builtin_amdgcn_kernarg_segment_ptr(), nrb, rows, s_col);
    } else if (sidx - a.n_strips < a.n_dd) {
        if (CP_TEST_ROLE & 8) dd_body((cp_ka_t)__builtin_amdgcn_kernarg_segment_ptr(), nrb, rows, sidx - a.n_strips);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
size_t cholp_flag_bytes(int n_strips) { return sizeof(unsigned) * ((size_t)CF_STRIP + 32 * (size_t)(n_strips > 0 ? n_strips : 1)); }

// Contexts of this process that can launch k_cholp, per device.  A launch is a set of workgroups that wait for one another and each fills a
// CU (LDS): two launches fit the chip side by side (2 x 110 workgroups at N = 500), a third one's workgroups could interleave with theirs at
// dispatch so that none of the three is complete.  With more than two such contexts alive on a device the launch-per-panel form is used
// (PRE3_CHOL_FORM=2 overrides; several PROCESSES sharing one GPU are not seen here: run those with PRE3_CHOL_FORM=0).  A wait that never
// ends is bounded in any case (stats[7] -> PRE3_E_HIP).
static std::atomic<int> g_cholp_live[64];
void cholp_context_count(int device, int delta) { if (device >= 0 && device < 64) g_cholp_live[device].fetch_add(delta); }

// crit and the rows are blocks 0, stride, 2 stride ..: they must all be in the first wave of the dispatch (one workgroup per CU), and with
// stride 8 they share one XCD's L2; an update of many panels (config 5: 40) takes the largest stride that still fits them, 0 if none does
static int cholp_stride(const pre3_ctx *c, int nH)
{
    if (8 * nH + 1 <= c->num_cus) return 8;        // up to 31 rows: crit and the rows on one XCD (their hand-offs stay in one L2)
    // more rows than that (config 5: 38): an ODD stride deals them over all eight XCDs -- their bulk tiles read and write S at several TB/s,
    // more than the one or two L2s an even stride would put them on can serve (measured: crit waited 5-10 us per panel for its next tiles)
    for (int st = 7; st >= 1; st -= 2) if (st * nH + 1 <= c->num_cus) return st;
    return 0;
}
constexpr int CP_WIN_MAX = 11;                     // plane blocks of W a strip keeps in LDS: 11 x 12 KB + CP (12 KB) + Yw2 (8.3 KB) = 152 KB

bool cholp_usable(const pre3_ctx *c, int nrb_max)
{
    static const int form = getenv("PRE3_CHOL_FORM") ? atoi(getenv("PRE3_CHOL_FORM")) : 1;
    if (form == 0 || !c->chol_persist) return false;
    if (form != 2 && c->device >= 0 && c->device < 64 && g_cholp_live[c->device].load() > 2) return false;
    // crit and the rows wait for one another: blocks 0, 8, .. 8 nH must be resident together, each on a CU of its own (the strips only wait
    // for them, so strips beyond the chip's capacity simply start later)
    const int nH = nrb_max > 2 ? nrb_max - 2 : 0;
    if (cholp_stride(c, nH) < 1) return false;
    return c->dtype == PRE3_F32 && c->k9_b3 && c->Wp != nullptr && c->Sp != nullptr && c->cholp_flags != nullptr && c->cholp_tp != nullptr &&
           nrb_max >= 1 && nrb_max <= CP_MAX_NRB && nrb_max <= c->rcap / NB;
}

// The down-date consumers' schedule: P's upper triangle in 64-column blocks, 4 x 4 super-blocks.  A diagonal super-block is ONE group (its 10 tiles
// read 4 column blocks of W); an off-diagonal one is two groups of 4 x 2 tiles (4 + 2 column blocks).  tiles64 lists every tile in group order
// (what k_downdate_b3 takes over when a launch cannot hold all groups), tile_off[g] the first tile of group g.
void dd_build_groups(int nb, std::vector<int32_t> &rec, std::vector<int2> &tiles64, std::vector<int> &tile_off)
{
    rec.clear(); tiles64.clear(); tile_off.clear();
    struct Grp { std::vector<int> slots; std::vector<std::pair<int, int>> tasks; };
    std::vector<Grp> groups;
    const int ns = (nb + 3) / 4;
    for (int SI = 0; SI < ns; ++SI)
        for (int SJ = SI; SJ < ns; ++SJ) {
            const int r0 = 4 * SI, r1 = std::min(nb, r0 + 4), c0 = 4 * SJ, c1 = std::min(nb, c0 + 4);
            if (SI == SJ) {
                Grp g;
                for (int i = r0; i < r1; ++i) g.slots.push_back(i);
                for (int i = r0; i < r1; ++i) for (int j = i; j < r1; ++j) g.tasks.push_back({ i - r0, j - r0 });
                groups.push_back(g);
            } else {
                for (int h = c0; h < c1; h += 2) {
                    Grp g;
                    for (int i = r0; i < r1; ++i) g.slots.push_back(i);
                    const int nr = r1 - r0;
                    for (int j = h; j < std::min(c1, h + 2); ++j) g.slots.push_back(j);
                    for (int i = r0; i < r1; ++i) for (int j = h; j < std::min(c1, h + 2); ++j) g.tasks.push_back({ i - r0, nr + j - h });
                    groups.push_back(g);
                }
            }
        }
    // Which group runs where: consumer block b lands on XCD b % 8 (observed placement: a speed assumption only), so the groups are dealt into eight
    // classes, class x = every eighth group of the emitted order.  Every XCD fetches the planes of the column blocks its groups touch ONCE into
    // its L2: with the groups in plain super-block order every class touched 88 % of the columns (340 of 8 x 48 block fetches at n = 3013); a greedy
    // start + pairwise swaps (deterministic, a few ms at creation) bring that to ~46 %, i.e. half the fabric traffic of the planes.
    const int G = (int)groups.size();
    std::vector<int> cls(G, 0);
    if (G >= 16) {
        const int cap = (G + 7) / 8, nbig = G % 8 == 0 ? 8 : G % 8;      // classes 0 .. nbig-1 hold `cap` groups, the others cap - 1
        std::vector<std::vector<int>> cnt(8, std::vector<int>(nb, 0));
        std::vector<int> size(8, 0);
        auto room = [&](int x) { return size[x] < (x < nbig ? cap : cap - 1); };
        for (int g = 0; g < G; ++g) {
            int best = -1, best_new = 1 << 30;
            for (int x = 0; x < 8; ++x) {
                if (!room(x)) continue;
                int nw = 0;
                for (int b : groups[g].slots) nw += cnt[x][b] == 0;
                if (nw < best_new || (nw == best_new && size[x] < size[best])) { best = x; best_new = nw; }
            }
            cls[g] = best; ++size[best];
            for (int b : groups[g].slots) ++cnt[best][b];
        }
        auto distinct = [&](int x) { int d = 0; for (int b = 0; b < nb; ++b) d += cnt[x][b] > 0; return d; };
        unsigned long long lcg = 88172645463325252ull;
        for (int it = 0; it < 120000; ++it) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            const int g1 = (int)((lcg >> 33) % (unsigned)G);
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            const int g2 = (int)((lcg >> 33) % (unsigned)G);
            const int x1 = cls[g1], x2 = cls[g2];
            if (x1 == x2) continue;
            const int before = distinct(x1) + distinct(x2);
            for (int b : groups[g1].slots) { --cnt[x1][b]; ++cnt[x2][b]; }
            for (int b : groups[g2].slots) { --cnt[x2][b]; ++cnt[x1][b]; }
            if (distinct(x1) + distinct(x2) <= before) { cls[g1] = x2; cls[g2] = x1; }
            else {
                for (int b : groups[g1].slots) { ++cnt[x1][b]; --cnt[x2][b]; }
                for (int b : groups[g2].slots) { ++cnt[x2][b]; --cnt[x1][b]; }
            }
        }
    }
    std::vector<std::vector<int>> by(8);
    for (int g = 0; g < G; ++g) by[cls[g]].push_back(g);
    std::vector<int> order;
    for (size_t k = 0; (int)order.size() < G; ++k)
        for (int x = 0; x < 8; ++x) if (k < by[x].size()) order.push_back(by[x][k]);
    for (int gi : order) {
        const Grp &g = groups[gi];
        if (g.tasks.empty()) continue;
        int32_t r[DG_WORDS] = { 0 };
        r[0] = (int)g.tasks.size() | ((int)g.slots.size() << 8);
        for (size_t k = 0; k < g.slots.size(); ++k) r[1 + k] = g.slots[k];
        tile_off.push_back((int)tiles64.size());
        for (size_t t = 0; t < g.tasks.size(); ++t) {
            r[7 + (t >> 2)] |= (g.tasks[t].first | (g.tasks[t].second << 4)) << (8 * (t & 3));
            tiles64.push_back(make_int2(g.slots[g.tasks[t].first], g.slots[g.tasks[t].second]));
        }
        rec.insert(rec.end(), r, r + DG_WORDS);
    }
    tile_off.push_back((int)tiles64.size());
}

// algorithmic work of a bracketed fused launch once its row count is known: SYRK n(n+1)r (the graded down-date), r^3/3 + n r^2 (factorisation + solve)
void cholp_timing_rows(pre3_ctx *c, int r)
{
    c->kt.pending = false;
    c->kt.flops += (double)c->n * ((double)c->n + 1.0) * r;
    c->kt.fact_flops += (double)r * r * r / 3.0 + (double)c->n * r * (double)r;
    c->kt.bytes += 1.5 * c->n * (double)c->n * c->esz + (double)c->n * (double)r * c->esz;
}

// nrb < 0: the row count is read on the device (stats[4]); nrb_max bounds the grid and the LDS.  rows: the real row count when nrb >= 0.
// Can the launch about to go out (LI update of a step, row count on the device) also carry the rescue stage and the HI update?  It needs the
// consumers to hold every tile of P (P is then written once, behind the HI panel), the strips' x-update, one panel of room behind the LI
// update's, and the tail's buffers.
bool cholp_tail_usable(const pre3_ctx *c, int nrb_max)
{
    if ( !c->tail_yp || !c->tail_hb || !c->tail_hib || !c->tail_wt || c->N <= 0 || c->N > 65535 || c->m <= 0 || c->m > 16384) return false;
    if (nrb_max + 1 > c->rcap / NB || nrb_max + 1 > CP_MAX_NRB + 1) return false;
    return true;
}

int launch_cholp(pre3_ctx *c, int nrb, int nrb_max, int rows, int which_prior, const CholpTailReq *tail_req)
{
    const int n_strips = c->ldw / 32;
    const int nH = nrb_max > 2 ? nrb_max - 2 : 0;
    const size_t lds_crit = CP_LP_OFF + sizeof(frag_t) * B3_SGRAN, lds_row = sizeof(RowSmem);
    const int win = std::max(1, std::min(nrb_max - 1, CP_WIN_MAX)), stride = cholp_stride(c, nH);
    PRE3_CHECK(stride >= 1, PRE3_E_ARG, "launch_cholp: %d panels need more CUs than the device has", nrb_max);
    const size_t lds_strip = (size_t)win * CP_WGRAN * 16 + (size_t)CP_WGRAN * 16 + (size_t)NB * CP_WS * sizeof(float);     // ring | CP (Yw) | Yw2
    size_t lds = std::max(lds_crit, std::max(lds_row, lds_strip));
    static std::atomic<unsigned long long> attr_set{ 0 };        // one bit per device
    if (c->device >= 0 && c->device < 64 && !((attr_set.load() >> c->device) & 1ull)) {
        PRE3_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cholp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.fetch_or(1ull << c->device);
    }
    // Down-date consumers (P -= W_J' W_J behind the strips, update.m:37): as many groups as there are CUs left.  Every workgroup of the launch
    // declares the same LDS (more than half a CU's), so there is one per CU, and the grid never exceeds the CU count: all of them are resident
    // together whatever the dispatch order -- a consumer can never hold the CU a strip, a row or crit is waiting for.
    static const int ov_env = getenv("PRE3_K9_OVERLAP") ? atoi(getenv("PRE3_K9_OVERLAP")) : 1;
    int n_dd = 0;
    // (with the consumers a launch fills the chip: it must be the only such launch in flight -- one live fp32 context on the device; with two,
    //  the launches run without consumers, 2 x 110 workgroups side by side as before)
    static const int form = getenv("PRE3_CHOL_FORM") ? atoi(getenv("PRE3_CHOL_FORM")) : 1;
    const bool alone = form == 2 || !(c->device >= 0 && c->device < 64) || g_cholp_live[c->device].load() <= 1;
    if (ov_env && alone && c->k9_overlap && c->dd_groups != nullptr && c->dd_n_groups > 0) {
        n_dd = std::min(c->dd_n_groups, c->num_cus - (1 + nH + n_strips));
        static const int dd_max = getenv("PRE3_DD_MAX") ? atoi(getenv("PRE3_DD_MAX")) : -1;      // (tests: fewer groups in the launch than would fit -- the rest goes to k_downdate_b3)
        if (dd_max >= 0) n_dd = std::min(n_dd, dd_max);
        if (n_dd < 0 || std::max(1 + nH + n_strips + n_dd, stride * nH + 1) > c->num_cus) n_dd = 0;
    }
    if (n_dd > 0) lds = std::max(lds, (size_t)4 * DG_SLOTS * DG_SLOT_GRAN * 16 + 12 * 1024);      // operand slots + one scratch line per wave (P warm-up)
    if (n_dd > 0) {
        // With consumers in the launch every strip waits for the strip that owns column ld, and the consumers wait for the strips: whatever the
        // dispatch order, nobody must wait for a workgroup that has no CU.  The grid never exceeds the CU count and every workgroup takes more than
        // half a CU's LDS; what is still checked, once per device: that the runtime really places one such workgroup per CU on THIS device and
        // configuration (occupancy query), and that the CU count the context holds is the device's.  If not: no consumers (round 3's form, whose
        // strips only wait for crit and the rows, which are dispatched in front of them).
        static std::atomic<int> resident_ok[64];                   // 0 unknown, 1 yes, -1 no
        int ok = (c->device >= 0 && c->device < 64) ? resident_ok[c->device].load() : -1;
        if (ok == 0) {
            int per_cu = 0, cus = 0;
            const hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_cholp), CP_NTH, 160 * 1024 - 1024);
            const hipError_t e2 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
            ok = (e1 == hipSuccess && e2 == hipSuccess && per_cu >= 1 && cus >= c->num_cus) ? 1 : -1;
            resident_ok[c->device].store(ok);
        }
        if (ok != 1) n_dd = 0;
    }
    // the tail (rescue stage + HI update inside this launch): every group of P's tiles must be in the launch, the strips do the x-update, the row
    // count is the device's (the speculative launch of a step)
    static const int xu_env0 = getenv("PRE3_CHOLP_XU") ? atoi(getenv("PRE3_CHOLP_XU")) : 1;
    const bool tail = tail_req != nullptr && nrb < 0 && n_dd > 0 && n_dd == c->dd_n_groups && xu_env0 && which_prior == PRE3_X_K_KM1 && cholp_tail_usable(c, nrb_max);
    if (tail) {
        const size_t strip_tail = ((size_t)(win + 1) * CP_WGRAN * 16 + (size_t)NB * CP_WS * sizeof(float) + 127) / 128 * 128 + sizeof(TailSmem);
        PRE3_CHECK(strip_tail <= CP_T_OFF && CP_TAIL_OFF + sizeof(TailSmem) <= CP_T_OFF, PRE3_E_ARG, "launch_cholp: the tail's LDS does not fit");
        lds = 160 * 1024;
    }
    PRE3_CHECK(lds <= 160 * 1024, PRE3_E_ARG, "launch_cholp: %d panels do not fit the strips' LDS", nrb_max);
    // flag words are never cleared within an epoch range: values are epoch + step, compared as signed differences.  Long before a word that has
    // not been written for 2^31 / 64 launches could read as "reached", the block is cleared and the epoch restarts.
    if (c->cholp_epoch >= (1u << 28)) {
        PRE3_HIP(hipMemsetAsync(c->cholp_flags, 0, cholp_flag_bytes(n_strips), c->stream));
        c->cholp_epoch = 0;
    }
    c->cholp_epoch += 64;
    CpArgs a{};
    a.S = (float *)c->Smat; a.W = (float *)c->W; a.ldw = c->ldw; a.ld = c->ld;
    a.Sp = c->Sp; a.sp_stride = c->rcap / NB; a.Tp = c->cholp_tp; a.Wp = c->Wp; a.nst_total = c->rcap / B3_BK;
    a.cf = c->cholp_flags; a.base = c->cholp_epoch; a.status = c->stats + 6;
    a.n_dev = nrb < 0 ? c->stats + 4 : nullptr; a.nrb = nrb < 0 ? nrb_max : nrb; a.nrb_max = nrb_max; a.n_strips = n_strips;
    a.win = win; a.stride = stride;
    static const int dd_mode = getenv("PRE3_DD_MODE") ? atoi(getenv("PRE3_DD_MODE")) : 17;     // bit 0: sc1 LDS-DMA of the planes; bit 1: an acquire per panel (experiment); bit 2: P warm-up; bit 4: write-through stores of P
    a.dd_mode = dd_mode;
    static const int poll_budget = getenv("PRE3_CHOLP_POLL") ? atoi(getenv("PRE3_CHOLP_POLL")) : 0;      // crit's wave 11: shader clocks per chain step spent polling (0: one poll per step, rounds 3-4)
    a.poll_budget = poll_budget;
    static const int poll_from = getenv("PRE3_CHOLP_POLL_FROM") ? atoi(getenv("PRE3_CHOLP_POLL_FROM")) : 3;
    a.poll_from = poll_from;
    static const int row_late = getenv("PRE3_ROW_LATE") ? atoi(getenv("PRE3_ROW_LATE")) : 0;      // measured (round 6): 5625 vs 5644 steps/s -- crit does not wait for the rows at present
    a.row_late = row_late;
    static const int crit_early = getenv("PRE3_CRIT_EARLY") ? atoi(getenv("PRE3_CRIT_EARLY")) : 1;
    a.crit_early = crit_early;
    static const int strip_rl = getenv("PRE3_STRIP_RL") ? atoi(getenv("PRE3_STRIP_RL")) : 1;
    a.strip_rl = strip_rl;
    // every group of P's tiles is in this launch and the strips finish x: the consumers also leave rows 3..6 behind for the gate (GateRide, pre3_geom.hip)
    a.jn_q = (!tail && n_dd > 0 && n_dd == c->dd_n_groups && c->jn_q != nullptr) ? c->jn_q : nullptr;
    c->jn_q_valid = a.jn_q != nullptr;
    // with the consumers in the launch the strips also finish the state: x_k_k = x_prior + W'(L^-1 nu) (the K9 launch that used to carry the
    // x-update as riders has nothing left to do at N = 500)
    static const int xu_env = getenv("PRE3_CHOLP_XU") ? atoi(getenv("PRE3_CHOLP_XU")) : 1;
    a.xu = (xu_env && n_dd > 0 && which_prior >= 0) ? 1 : 0;
    a.n = c->n; a.x_prior = which_prior == PRE3_X_K_K ? c->x_kk : c->x_km1; a.x_out = c->x_kk; a.params = c->pred_params;
    a.P = (float *)c->P; a.dd = c->dd_groups; a.n_dd = n_dd; a.rows = nrb < 0 ? nrb_max * NB : (rows > 0 && rows <= nrb * NB ? rows : nrb * NB);
    CpTail t{};
    a.tail = tail ? 1 : 0;
    // without the tail: the strips still end with the rescue stage's projection when the step wants the gate to ride with the Jnorm pass (GateRide)
    static const int proj_env = getenv("PRE3_CHOLP_PROJ") ? atoi(getenv("PRE3_CHOLP_PROJ")) : 1;
    a.proj = (proj_env && !tail && c->want_gate_ride && a.jn_q != nullptr && a.xu && c->N > 0 && lds_strip <= CP_T_OFF) ? 1 : 0;
    if (a.proj) {
        t.N = c->N; t.lm_type = c->lm.type; t.lm_off = c->lm.off; t.has_h = c->lm.has_h; t.h = c->lm.h; t.Hc = c->lm.Hc; t.Hl = c->lm.Hl;
        t.cam = CamD{ c->cam.f, c->cam.Cx, c->cam.Cy, c->cam.k1, c->cam.k2, (double)c->cam.nRows, (double)c->cam.nCols };
        lds = 160 * 1024;
    }
    c->proj_in_cholp = a.proj != 0;
    if (tail) {
        t.on = 1; t.N = c->N; t.m = c->m; t.seq = tail_req->seq; t.ykcap = c->rcap; t.chi2 = tail_req->chi2;
        t.lm_type = c->lm.type; t.lm_off = c->lm.off; t.lm_ic = c->lm.ic; t.lm_li = c->lm.li; t.meas = c->meas;
        t.lm_hi = c->lm.hi; t.has_h = c->lm.has_h; t.hi_meas = c->hi_meas; t.sel_rows = c->sel_rows; t.stats = c->stats; t.mail = c->mail_dev;
        t.h = c->lm.h; t.Hc = c->lm.Hc; t.Hl = c->lm.Hl; t.z = c->lm.z;
        t.cam = CamD{ c->cam.f, c->cam.Cx, c->cam.Cy, c->cam.k1, c->cam.k2, (double)c->cam.nRows, (double)c->cam.nCols };
        t.Yp = c->tail_yp; t.Hb = c->tail_hb; t.hib = c->tail_hib;
        a.Wt = c->tail_wt; a.kcap = c->rcap;
#ifdef PRE3_TAIL_DEBUG
        t.dbgS = c->lm.S;
#endif
    }
    c->tail_launched = tail;
    // PRE3_OPT_PEND_HI: with every group of P's tiles in the launch the consumers take the pending HI down-date as the panels in front of panel 0
    // (and run even when the device finds no LI rows); otherwise it goes out as its own launch now
    if (c->pend_rows > 0) {
        if (!tail && n_dd > 0 && n_dd == c->dd_n_groups && c->Wp_pend != nullptr) { a.Wp_pend = c->Wp_pend; a.pend_ns = (c->pend_rows + B3_BK - 1) / B3_BK; c->pend_rows = 0; }
        else PRE3_TRY(pend_flush(c));
    }
    // roofline bracket (pre3_kernel_timing): the launches that carry a matrix-bound down-date -- updates of the predicted state
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = c->kt.enabled && n_dd > 0 && which_prior == PRE3_X_K_KM1 && (nrb < 0 || nrb >= 2) && (c->kt.seen++ % c->kt.every) == 0;
    if (timed) {
        if ((size_t)c->kt.used + 2 > c->kt.ev.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; PRE3_HIP(hipEventCreate(&e)); c->kt.ev.push_back(e); }
        }
        e0 = c->kt.ev[c->kt.used]; e1 = c->kt.ev[c->kt.used + 1];
        c->kt.used += 2;
        PRE3_HIP(hipEventRecord(e0, c->stream));
    }
    hipLaunchKernelGGL(k_cholp, dim3(std::max(1 + nH + n_strips + n_dd, stride * nH + 1)), dim3(CP_NTH), lds, c->stream, a, t);
    if (timed) {
        PRE3_HIP(hipEventRecord(e1, c->stream));
        c->kt.fused += 1;
        if (nrb < 0) c->kt.pending = true;                      // the row count arrives with the mailbox: cholp_timing_rows()
        else cholp_timing_rows(c, a.rows);
    }
    PRE3_HIP(hipGetLastError());
    c->split_rows = (nrb < 0 ? nrb_max : nrb) * NB;             // the strips' epilogues have written every plane k_downdate_b3 reads
    c->x_done = a.xu != 0;
    c->dd_done = n_dd;                                          // the next launch_downdate only covers the groups behind these
    return PRE3_OK;
}

#ifdef PRE3_PROBE
extern "C" __attribute__((visibility("default"))) int pre3_debug_cholp(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cp), sizeof(unsigned long long) * 24 * 16 * 8) == hipSuccess ? 0 : -3;
}
// the flag-driven chain's per-role stamps of crit's LAST chain (pre3_chain_async.h: [role 8][step 12][slot 4], shader clocks)
#ifdef PRE3_PROBE_CHA
extern "C" __attribute__((visibility("default"))) int pre3_debug_cha(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cha), sizeof(unsigned long long) * 8 * 12 * 4) == hipSuccess ? 0 : -3;
}
#endif
#endif

}  // namespace pre3
