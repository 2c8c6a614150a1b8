// pre3_chain.h -- the dependent chain of one 64-column Cholesky panel (shared by the launch-per-panel kernel k_chol_step in
// pre3_update.hip and the persistent factorisation in pre3_cholp.hip), the MFMA traits and the bf16-plane helpers.
#pragma once
#include "pre3_internal.h"
#include "pre3_geomdev.h"

namespace pre3 {

// three-way bf16 split of W into the stage image of k_downdate_b3 (layout: see "K9 on the bf16 matrix cores" below)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define B3_NBUF 3                  // LDS ring slots of k_downdate_b3 (the register pipeline runs two stages ahead)
constexpr int B3_T = 128, B3_BK = 16, B3_GRAN = 3 * 4 * 64;        // granules (16 B) of one operand block of one stage
// x[0..7] = eight consecutive k of one column -> this lane's granule of the three planes (dst: plane 0; planes are 256 granules apart)
template <int PSTRIDE = 256>
__device__ __forceinline__ void b3_split_store(const float (&x)[8], bf16x8_t *__restrict__ dst)
{
    bf16x8_t a, b, c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 ha = (__bf16)x[j];
        const float r1 = x[j] - (float)ha;          // exact
        const __bf16 hb = (__bf16)r1;
        const float r2 = r1 - (float)hb;            // exact
        a[j] = ha; b[j] = hb; c[j] = (__bf16)r2;
    }
    dst[0] = a; dst[PSTRIDE] = b; dst[2 * PSTRIDE] = c;
}
constexpr int B3_SGRAN = 4 * 3 * 2 * 64;          // granules of one 64x64 S block's planes: [k-step 4][plane 3][32-row half 2][lane 64]
// one (128 columns x 16 k) block from W in global memory; 256 threads
__device__ __forceinline__ void b3_split_block(const float *__restrict__ W, int ldw, bf16x8_t *__restrict__ Wp, int nst_total, int cb, int st, int tid)
{
    const int f = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
    const float *src = W + (size_t)(st * B3_BK + 8 * h) * ldw + cb * B3_T + f * 32 + r;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = src[(size_t)j * ldw];
    b3_split_store(x, Wp + ((size_t)cb * nst_total + st) * B3_GRAN + f * 64 + l);
}

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    static constexpr int BLK = 32, KS = 2, NREG = 16;
    typedef float acc_t __attribute__((ext_vector_type(16)));
    static __device__ inline void mma(float a, float b, acc_t &c) { c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
    static __device__ inline int col(int lane) { return lane & 31; }
    static __device__ inline int kk(int lane) { return lane >> 5; }
};
template <> struct Mfma<double> {
    static constexpr int BLK = 16, KS = 4, NREG = 4;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ inline void mma(double a, double b, acc_t &c) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
    static __device__ inline int col(int lane) { return lane & 15; }
    static __device__ inline int kk(int lane) { return lane >> 4; }
};

// 1/sqrt(x): hardware estimate + one Newton step (fp32: v_rsq_f32, fp64: v_rsq_f64) -- keeps the dependent
// chain of the 8x8 factorisation short; error <= 2 ulp, far inside the tolerances of DESIGN.md
__device__ inline float fast_rsqrt(float x) { float y = __builtin_amdgcn_rsqf(x); return y * (1.5f - 0.5f * x * y * y); }
__device__ inline double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}

#ifdef PRE3_PROBE
static __device__ unsigned long long g_probe[16];
static __device__ int g_probe_block = 5;
static __device__ int g_rt_on = 0;                         // pre3_debug_rt(1): stamp every workgroup of the panel-2 launch                   // the workgroup whose panel phases are stamped
static __device__ unsigned long long g_k9[64 * 8 * 4];
static __device__ unsigned long long g_k9rt[2048 * 4];     // s_memrealtime (100 MHz, chip-wide) per workgroup of the one-tile kernel
static __device__ unsigned int g_k9hw[2048];               // HW_ID of wave 0 (CU / SE / XCC placement)
#define PROBE_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == g_probe_block) { g_probe[k] = __builtin_amdgcn_s_memtime(); g_k9[256 + J * 8 + (k)] = g_probe[k]; } } while (0)    /* inside chol_panel_body: per panel J */
#define PROBE_ACC(k, t0) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_probe[k] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#define PROBE_T(k, T_) do { if (threadIdx.x == (T_) && blockIdx.x == g_probe_block) g_probe[k] = __builtin_amdgcn_s_memtime(); } while (0)
#ifdef PRE3_PROBE_STEPS
// per pipeline step of the panel chain: start / end-of-work stamps of the factor wave (0), the z wave (1) and the first worker wave (2)
#define PROBE_STEP(k, e) do { if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) >= 8) && blockIdx.x == g_probe_block) g_k9[((k) + 1) * 8 + ((threadIdx.x >> 6) == 8 ? 0 : (threadIdx.x >> 6) == 9 ? 1 : 2) * 2 + (e)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PROBE_F(k, j) do { if ((k) == 3 && (threadIdx.x & 63) == 0 && blockIdx.x == g_probe_block) { __builtin_amdgcn_sched_barrier(0); g_k9[100 + (j)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PROBE_F(k, j)
#define PROBE_STEP(k, e)
#endif
#else
#define PROBE_F(k, j)
#define PROBE_STEP(k, e)
#define PROBE_STAMP(k)
#define PROBE_ACC(k, t0)
#define PROBE_T(k, T_)
#endif

// wave-uniform lane read (v_readlane_b32): a few cycles, no LDS round trip
__device__ inline float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline double rdlane(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// hardware 1/sqrt: fp32 takes v_rsq_f32 as is (1 ulp), fp64 refines v_rsq_f64 twice
__device__ inline float chain_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline double chain_rsqrt(double x) { return fast_rsqrt(x); }
// hardware 1/x for the pivots: fp32 takes v_rcp_f32 as is (1 ulp), fp64 refines v_rcp_f64 twice
__device__ inline float chain_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline double chain_rcp(double x) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); return r * (2.0 - x * r); }

// Panel kernel.  What bounds it is the dependent chain of the 64 columns, not flops, so the chain runs on a
// dedicated wave and everything else is kept off it.  The 64 columns are swept in 8 sub-panels of MB = 8; a workgroup is TEN waves:
//   wave 8  (factor wave; lane = row i of the L block): per sub-panel s it takes the columns as published by the
//           D workers (updated through sub-panel s-2), applies sub-panel s-1's update itself (lookahead; the 8x8
//           multipliers are a broadcast LDS read of what the wave wrote one step earlier -- no cross-lane VALU work),
//           then factors right-looking inside the sub-panel, division-free: per column one pivot v_readlane, v_rcp, v_rsq and
//           7-c (v_readlane, multiply, fma) triples.  Writes its row of L into Ls and the eight 1/sqrt(pivot) into Rs.
//   wave 9  (z wave; lane = column i of the workgroup's X block), ONE step behind: x <- L8^-1 (x - lookahead), all
//           coefficients (the 8x8 diagonal sub-block, the 8x8 block left of it, Rs) broadcast from LDS.
//   waves 0-3 (D workers) and 4-7 (X workers): wave w holds 32x32 tile (w>>1 & 1, w&1) of the diagonal block D / of the workgroup's X
//           block as MFMA accumulators and applies the rank-8 updates of finished sub-panels on the matrix cores (ch_worker_step below),
//           then publishes the next sub-panel's columns of L (Pn) / rows of X (Xr).  The pending update of panel J-1 (the launch's
//           prologue) is accumulated by the same waves straight into these tiles: acc = raw tile - (products from the bf16 planes).
// ONE workgroup barrier per step, 10 steps per 64 columns (was 18 with 4-column micro-panels and the X solve on the
// factor wave).  Every workgroup factors the diagonal block redundantly; workgroup b then owns X = its 64-row block
// of [S ; HP'] (b = 0: the diagonal block itself).
constexpr int CH_MB = 8, CH_NSP = NB / CH_MB, CH_NTH = 640;
#ifndef CH_EXP_Z
#define CH_EXP_Z 1          // timing experiments (tools/probe_panel.hip): 0 = the z wave idles
#define CH_EXP_WX 1         // 0 = the workers skip the X tiles
#define CH_EXP_F 1          // 0 = the factor wave idles
#endif
template <typename T> struct ChLs { static constexpr int STRIDE = sizeof(T) == 4 ? NB + 4 : NB + 2; };   // rows 16-byte aligned
template <typename T>
struct ChPipe {
    __attribute__((aligned(16))) T Pn[3][NB][CH_MB];   // (the lock-step chain uses two slots, the flag-driven one up to three) published sub-panel columns: Pn[par][i][t] = A[i][C+t], updated through sub-panel s-2
    __attribute__((aligned(16))) T Zt[2][NB][CH_MB];   // final X[C+t][i], transposed for the workers
    __attribute__((aligned(16))) T Xr[2][CH_MB][NB];   // published rows of X: Xr[par][t][i] = X[C+t][i]
    __attribute__((aligned(16))) T Rs[2][CH_MB];       // 1/sqrt(pivot) of the sub-panel's columns
    __attribute__((aligned(16))) T RsA[CH_NSP][CH_MB]; // the flag-driven chain (pre3_chain_async.h): the factor wave runs ahead of the z wave, every sub-panel keeps its slot
    unsigned fl[16];                                   // its progress counters
};
template <typename T>
struct ChSmem {
    __attribute__((aligned(16))) T Ls[NB][ChLs<T>::STRIDE];
    T Xs[NB][NB + 1];                              // Xs[a][i]
    union {
        T As[NB][NB + 1];                          // operand tiles of the trailing update / of the fused prologue
        ChPipe<T> pipe;                            // the chain's hand-off buffers (the prologue is over by then)
    };
    T Bs[NB][NB + 1];                              // Bs[j][a]
};

// Worker waves of the panel chain.  The four 32x32 tiles of the diagonal block D and of the workgroup's X block live as MFMA
// accumulators (wave WV owns tile (WV>>1, WV&1) of both; D's tile above the diagonal is dead).  A finished sub-panel is a rank-8
// update: D -= Y Y', X -= Y Z on the fp32 / fp64 matrix cores (the f32-input MFMA is an exact fma chain).  MFMA step j multiplies
// k = NJ*kk(lane) + j, so that a lane's NJ operand values are contiguous: ONE 16-byte LDS read per operand and rank-8 update,
// against 26 per lane and step with register patches on the VALU -- the LDS queue, not arithmetic, was what the chain waited for.
template <typename T> struct ChW {
    using M = Mfma<T>;
    static constexpr int NBLK = 32 / M::BLK, NJ = CH_MB / M::KS;
    typedef T vk_t __attribute__((ext_vector_type(CH_MB / M::KS), aligned(16)));
    typedef typename M::acc_t acc_t;
};

// raw tile from LDS (the D tile of Ls, or the X tile of Xs[a][i]: tile rows = panel columns a, tile columns = i)
template <typename T, int WV, bool XSIDE>
__device__ __forceinline__ void ch_worker_load(ChSmem<T> &sm, typename ChW<T>::acc_t (&acc)[ChW<T>::NBLK][ChW<T>::NBLK], const int lane, const bool live)
{
    using M = Mfma<T>;
    constexpr int NBLK = ChW<T>::NBLK, w0 = (WV >> 1) * 32, w1 = (WV & 1) * 32;
    const int cl = M::col(lane);
    // (the dead tile's zero is opaque: as a constant the compiler built ONE hoisted all-zero accumulator tuple for every such use in the kernel, took
    //  its first register for "the zero register" of unrelated address arithmetic, and spilled and reloaded the whole tuple around the chain)
    T zero = (T)0;
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int p = 0; p < NBLK; ++p)
#pragma unroll
        for (int q = 0; q < NBLK; ++q)
#pragma unroll
            for (int e = 0; e < M::NREG; ++e) {
                const int r = p * M::BLK + M::row(lane, e), c = q * M::BLK + cl;
                acc[p][q][e] = !live ? zero : XSIDE ? sm.Xs[w0 + r][w1 + c] : sm.Ls[w0 + r][w1 + c];
            }
}

// One pipeline step of a worker wave.  D side (waves 0-3; the tile above the diagonal, WV = 1, is dead): D -= Y(k-1) Y(k-1)' where the tile
// still has columns >= 8(k+1) (sub-panel k's own columns get it from the factor wave's lookahead), then publish sub-panel k+1's columns.
// X side (waves 4-7): X -= Y(k-2) Z(k-2) where the tile still has rows >= 8k (sub-panel k-1's rows get it from the z wave's lookahead), then
// publish rows 8k..8k+7.  One LDS round trip and one MFMA chain per wave and step (the two sides used to share a wave, back to back).
template <typename T, int WV, bool XSIDE, bool RELAX = false>
__device__ __forceinline__ void ch_worker_step(const int k, ChSmem<T> &sm, typename ChW<T>::acc_t (&acc)[ChW<T>::NBLK][ChW<T>::NBLK], const int lane, const bool live)
{
    using M = Mfma<T>;
    typedef typename ChW<T>::vk_t vk_t;
    constexpr int NBLK = ChW<T>::NBLK, NJ = ChW<T>::NJ, MB = CH_MB, NSP = CH_NSP, w0 = (WV >> 1) * 32, w1 = (WV & 1) * 32;
    const int cl = M::col(lane), kq = M::kk(lane) * NJ;
    if (!live) return;
    if constexpr (!XSIDE) {
        if (WV == 1) return;
        if (k >= 1 && k + 1 <= NSP - 1 && w1 + 32 > MB * (k + 1)) {
            const int C = MB * (k - 1);
            vk_t a[NBLK], bb[NBLK];
#pragma unroll
            for (int p = 0; p < NBLK; ++p) {
                a[p] = *reinterpret_cast<const vk_t *>(&sm.Ls[w0 + p * M::BLK + cl][C + kq]);
                bb[p] = *reinterpret_cast<const vk_t *>(&sm.Ls[w1 + p * M::BLK + cl][C + kq]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int q = 0; q < NBLK; ++q) {
                        // fp64 (16 x 16 blocks, an MFMA every 64 cycles): a block whose columns are all factored, or that lies above the diagonal, is dead
                        if constexpr (sizeof(T) == 8) {
                            const int c_lo = w1 + q * M::BLK > MB * (k + 1) ? w1 + q * M::BLK : MB * (k + 1);
                            if (w1 + (q + 1) * M::BLK <= MB * (k + 1) || w0 + (p + 1) * M::BLK - 1 < c_lo) continue;
                        }
                        M::mma(-a[p][j], bb[q][j], acc[p][q]);
                    }
        }
        if (k + 1 <= NSP - 1) {
            // publish sub-panel k+1: columns Cn..Cn+7 of D (the tile's rows), from the accumulators
            const int Cn = MB * (k + 1), par = (k + 1) & 1;
            if (Cn >= w1 && Cn < w1 + 32) {
                const int q = (Cn - w1) / M::BLK, c8 = (Cn - w1) % M::BLK;
                if (cl >= c8 && cl < c8 + MB) {
#pragma unroll
                    for (int p = 0; p < NBLK; ++p)
#pragma unroll
                        for (int e = 0; e < M::NREG; ++e) sm.pipe.Pn[par][w0 + p * M::BLK + M::row(lane, e)][cl - c8] = acc[p][q][e];
                }
            }
        }
    } else {
        if (CH_EXP_WX && k >= 2 && k <= NSP - 1 && w0 + 32 > MB * k) {
            const int C = MB * (k - 2), par = (k - 2) & 1;
            vk_t a[NBLK], bb[NBLK];
            // (Zt's address is rebuilt from the lane id at every step -- three vector instructions -- instead of living in a register for the
            //  whole chain: the chain takes every register the kernel has, and inside the persistent kernel that one value was the spill)
            int clz = cl;
            if constexpr (RELAX) asm volatile("" : "+v"(clz));
#pragma unroll
            for (int p = 0; p < NBLK; ++p) {
                a[p] = *reinterpret_cast<const vk_t *>(&sm.Ls[w0 + p * M::BLK + cl][C + kq]);
                bb[p] = *reinterpret_cast<const vk_t *>(&sm.pipe.Zt[par][w1 + p * M::BLK + clz][kq]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int q = 0; q < NBLK; ++q) {
                        if constexpr (sizeof(T) == 8) { if (w0 + (p + 1) * M::BLK <= MB * k) continue; }      // (fp64: the block's rows are all solved)
                        M::mma(-a[p][j], bb[q][j], acc[p][q]);
                    }
        }
        if (CH_EXP_WX && k >= 0 && k <= NSP - 1) {
            // publish rows Cr..Cr+7 of X
            const int Cr = MB * k, par = k & 1;
            if (Cr >= w0 && Cr < w0 + 32) {
                const int r8 = Cr - w0;
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int e = 0; e < M::NREG; ++e)
                        if ((p * M::BLK + M::row(0, e)) / MB * MB == r8) {       // the 8-row group this register belongs to (same for all lanes)
                            const int t = p * M::BLK + M::row(lane, e) - r8;
#pragma unroll
                            for (int q = 0; q < NBLK; ++q) sm.pipe.Xr[par][t][w1 + q * M::BLK + cl] = acc[p][q][e];
                        }
            }
        }
    }
}

// The chain of one panel.  On entry Ls holds the (fully updated) diagonal block and Xs the workgroup's X block (hasX), or the
// worker waves already carry their tiles in `acc` (acc_loaded: the fused prologue of k_chol_step); on exit Ls holds L_JJ and
// Xs[a][i] the solved block L_JJ^-1 X.  Waves 0-7 workers, 8 factor wave, 9 z wave; any further wave of the workgroup (the
// persistent kernel's publisher / fetcher waves) runs side_step(k) once per pipeline step and joins the step's barrier.
// EARLY: the panel's last `NSP - nsp_eff` sub-panels are padding (identity rows and columns of S against zero rows of the right-hand side: the
// last panel of an update whose row count is not a multiple of 64).  Their steps change nothing -- L keeps its identity columns, the
// right-hand side its rows -- so the steps behind the one in which sub-panel nsp_eff-1 passes the z wave (k = nsp_eff) are skipped: an update of 20 rows
// runs 5 of the 10 steps.  nsp_eff is workgroup-uniform.
// z_tail: run by the z wave behind its last sub-panel (the whole of X is final in Xs for this wave: LDS operations of one wave are in order),
// before the chain's last barrier -- crit turns M_J's last rows into planes there instead of behind the chain.
// worker_init(acc, xside, wv): run by a worker wave INSTEAD of the load from LDS when acc_loaded is set and the caller has no tiles in `acc` yet
// (ChNoInit: the caller's `acc` is taken as it is).  The persistent kernel's crit computes D_{J+1} there -- a tile defined inside the worker branch
// is dead on the factor and z waves' paths; defined outside the chain it would be sixteen live registers on every path through it.
struct ChNoTail { __device__ __forceinline__ void operator()() const {} };
struct ChNoInit { static constexpr bool none = true; template <typename A> __device__ __forceinline__ void operator()(A &, bool, int) const {} };
template <typename T, bool RELAX = false, bool EARLY = false, typename SideStep, typename ZTail = ChNoTail, typename WInit = ChNoInit>
__device__ __forceinline__ void chol_chain(ChSmem<T> &sm, typename ChW<T>::acc_t (&acc)[ChW<T>::NBLK][ChW<T>::NBLK], const bool acc_loaded,
                                           const bool hasX, bool &bad, SideStep &&side_step, const int nsp_eff = CH_NSP, ZTail &&z_tail = ChNoTail{},
                                           WInit &&worker_init = ChNoInit{})
{
    constexpr int MB = CH_MB, NSP = CH_NSP;
    typedef T v4_t __attribute__((ext_vector_type(4), aligned(16)));
    typedef T T2 __attribute__((ext_vector_type(2)));
    auto &Ls = sm.Ls; auto &Xs = sm.Xs; auto &Pn = sm.pipe.Pn; auto &Zt = sm.pipe.Zt; auto &Xr = sm.pipe.Xr; auto &Rs = sm.pipe.Rs;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int role = wave == 8 ? 0 : wave == 9 ? 1 : wave < 8 ? 2 : 3;        // 0: factor wave, 1: z wave, 2: worker, 3: side wave
    const bool worker = role == 2, xside = wave >= 4 && wave < 8;
    const int wv = wave & 3;
    const bool tile_live = worker && (xside ? hasX : wv != 1);
    if (worker && !acc_loaded) {
        if (xside) {
            if (wv == 0) ch_worker_load<T, 0, true>(sm, acc, lane, tile_live);
            else if (wv == 1) ch_worker_load<T, 1, true>(sm, acc, lane, tile_live);
            else if (wv == 2) ch_worker_load<T, 2, true>(sm, acc, lane, tile_live);
            else ch_worker_load<T, 3, true>(sm, acc, lane, tile_live);
        } else {
            if (wv == 0) ch_worker_load<T, 0, false>(sm, acc, lane, tile_live);
            else if (wv == 1) ch_worker_load<T, 1, false>(sm, acc, lane, tile_live);
            else if (wv == 2) ch_worker_load<T, 2, false>(sm, acc, lane, tile_live);
            else ch_worker_load<T, 3, false>(sm, acc, lane, tile_live);
        }
    } else if (worker) {
        worker_init(acc, xside, wv);
    }
    T yprev[MB], zprev[MB];                          // factor wave: its row of Y(s-1); z wave: its column of Z(s-1)
#pragma unroll
    for (int t = 0; t < MB; ++t) { yprev[t] = (T)0; zprev[t] = (T)0; }
    // Software pipeline, fully unrolled (every index below is a compile-time constant).  Step k:
    //   factor wave: sub-panel k (0 <= k < NSP);   z wave: sub-panel k-1 (1 <= k <= NSP);
    //   workers (ch_worker_step): D tiles -= Y(k-1) Y(k-1)' for columns >= 8(k+1) (sub-panel k's own columns get it from the factor
    //            wave's lookahead), publish Pn(k+1);  X tiles -= Y(k-2) Z(k-2) for rows >= 8k, publish Xr(k).
    // LDS latency is what a step costs: every role issues ALL the reads of a phase first and waits once (LDS_GROUP keeps the
    // compiler from re-interleaving reads, waits and arithmetic, which serialised a dozen LDS round trips per step).
#define LDS_GROUP() do { if constexpr (sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0); } while (0)      // (fp64: the register budget does not allow it)
#pragma unroll
    for (int k = -1; k <= NSP; ++k) {
        if constexpr (EARLY) { if (k > nsp_eff) continue; }       // (not `break`: a second loop exit keeps the optimizer from unrolling, and every index below is a constant only unrolled)
        PROBE_STEP(k, 0);
        if (worker) {
            if (xside) {
                if (wv == 0) ch_worker_step<T, 0, true, RELAX>(k, sm, acc, lane, tile_live);
                else if (wv == 1) ch_worker_step<T, 1, true, RELAX>(k, sm, acc, lane, tile_live);
                else if (wv == 2) ch_worker_step<T, 2, true, RELAX>(k, sm, acc, lane, tile_live);
                else ch_worker_step<T, 3, true, RELAX>(k, sm, acc, lane, tile_live);
            } else {
                if (wv == 0) ch_worker_step<T, 0, false>(k, sm, acc, lane, tile_live);
                else if (wv == 1) ch_worker_step<T, 1, false>(k, sm, acc, lane, tile_live);
                else if (wv == 2) ch_worker_step<T, 2, false>(k, sm, acc, lane, tile_live);
                else ch_worker_step<T, 3, false>(k, sm, acc, lane, tile_live);
            }
        } else if (role == 0) {
            if (k >= 0 && k < NSP && CH_EXP_F) {
                const int C = MB * k, par = k & 1, i = tid & 63;
                T2 a2[MB / 2];
                T y[MB], rsv[MB];
                PROBE_F(k, 0);
                // the published columns (updated through sub-panel k-2) and, for the lookahead, rows C..C+7 of what this wave stored one
                // step ago (broadcast reads): all requested at once, one wait
                const v4_t v0 = *reinterpret_cast<const v4_t *>(&Pn[par][i][0]), v1 = *reinterpret_cast<const v4_t *>(&Pn[par][i][4]);
                v4_t Lh[MB][2];
                if (k > 0 && sizeof(T) == 4) {
#pragma unroll
                    for (int t = 0; t < MB; ++t) {
                        Lh[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB]); Lh[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB + 4]);
                    }
                }
                LDS_GROUP();
                a2[0] = T2{ v0[0], v0[1] }; a2[1] = T2{ v0[2], v0[3] }; a2[2] = T2{ v1[0], v1[1] }; a2[3] = T2{ v1[2], v1[3] };
                PROBE_F(k, 1);
                if (k > 0) {
                    // lookahead: sub-panel k-1's update of these eight columns, a[t] -= sum_u y[u] L[C+t][C-8+u]; adjacent u pair up in
                    // packed fmas (operands are register pairs as loaded)
                    const T2 yp[4] = { T2{ yprev[0], yprev[1] }, T2{ yprev[2], yprev[3] }, T2{ yprev[4], yprev[5] }, T2{ yprev[6], yprev[7] } };
#pragma unroll
                    for (int t = 0; t < MB; ++t) {
                        if constexpr (sizeof(T) == 8) { Lh[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB]); Lh[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB + 4]); }
                        T2 acc = yp[0] * T2{ Lh[t][0][0], Lh[t][0][1] };
                        acc += yp[1] * T2{ Lh[t][0][2], Lh[t][0][3] };
                        acc += yp[2] * T2{ Lh[t][1][0], Lh[t][1][1] };
                        acc += yp[3] * T2{ Lh[t][1][2], Lh[t][1][3] };
                        a2[t >> 1][t & 1] -= acc[0] + acc[1];
                    }
                }
                PROBE_F(k, 2);
                // right-looking inside the sub-panel, division-free on the dependent chain: per column ONE v_readlane of the pivot,
                // v_rcp, and a[t] -= a[c] * (a[c]@row(C+t) / pivot), two columns t per packed instruction; y = a * rsqrt(pivot) is off the
                // chain (measured on one wave, tools/probe_chain.hip: 467 cycles per sub-panel against 738 for rsq -> mul -> readlane ->
                // fma).  Rows above the diagonal compute garbage that nothing reads; a non-positive pivot poisons the values and raises
                // the status flag.
#pragma unroll
                for (int c = 0; c < MB; ++c) {
                    const T ac = a2[c >> 1][c & 1];
                    const T piv = rdlane(ac, C + c);
                    T so = (T)0;
                    T2 st2[MB / 2];
                    if ((c & 1) == 0) so = rdlane(ac, C + c + 1);
#pragma unroll
                    for (int m = (c >> 1) + 1; m < MB / 2; ++m) st2[m] = T2{ rdlane(ac, C + 2 * m), rdlane(ac, C + 2 * m + 1) };
                    const T rinv = chain_rcp(piv);
                    rsv[c] = chain_rsqrt(piv);
                    bad |= !(piv > (T)0);
                    if ((c & 1) == 0) a2[c >> 1][1] -= ac * (so * rinv);
#pragma unroll
                    for (int m = (c >> 1) + 1; m < MB / 2; ++m) a2[m] -= T2{ ac, ac } * (st2[m] * T2{ rinv, rinv });
                    y[c] = ac * rsv[c];
                }
                PROBE_F(k, 3);
                *reinterpret_cast<v4_t *>(&Ls[i][C]) = v4_t{ y[0], y[1], y[2], y[3] };
                *reinterpret_cast<v4_t *>(&Ls[i][C + 4]) = v4_t{ y[4], y[5], y[6], y[7] };
                if (i == 0) {
                    *reinterpret_cast<v4_t *>(&Rs[par][0]) = v4_t{ rsv[0], rsv[1], rsv[2], rsv[3] };
                    *reinterpret_cast<v4_t *>(&Rs[par][4]) = v4_t{ rsv[4], rsv[5], rsv[6], rsv[7] };
                }
#pragma unroll
                for (int t = 0; t < MB; ++t) yprev[t] = y[t];
                PROBE_F(k, 4);
            }
        } else if (role == 3) {
            side_step(k);
        } else if (hasX && k >= 1 && k <= NSP && CH_EXP_Z) {
            // z wave, sub-panel k-1: z <- L8^-1 (x - lookahead)
            const int s2 = k - 1, C = MB * s2, par = s2 & 1, i = tid & 63;
            T x[MB], z[MB];
            v4_t Lh[MB][2], Ld[MB][2];
#pragma unroll
            for (int t = 0; t < MB; ++t) x[t] = Xr[par][t][i];
            if (s2 > 0 && sizeof(T) == 4) {
#pragma unroll
                for (int t = 0; t < MB; ++t) {
                    Lh[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB]); Lh[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB + 4]);
                }
            }
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int t = 1; t < MB; ++t) {
                    Ld[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C]);
                    if (t > 4) Ld[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C + 4]);
                }
            }
            const v4_t r0 = *reinterpret_cast<const v4_t *>(&Rs[par][0]), r1 = *reinterpret_cast<const v4_t *>(&Rs[par][4]);
            LDS_GROUP();
            if (s2 > 0) {
                const T2 zp[4] = { T2{ zprev[0], zprev[1] }, T2{ zprev[2], zprev[3] }, T2{ zprev[4], zprev[5] }, T2{ zprev[6], zprev[7] } };
#pragma unroll
                for (int t = 0; t < MB; ++t) {
                    if constexpr (sizeof(T) == 8) { Lh[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB]); Lh[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB + 4]); }
                    T2 acc = zp[0] * T2{ Lh[t][0][0], Lh[t][0][1] };
                    acc += zp[1] * T2{ Lh[t][0][2], Lh[t][0][3] };
                    acc += zp[2] * T2{ Lh[t][1][0], Lh[t][1][1] };
                    acc += zp[3] * T2{ Lh[t][1][2], Lh[t][1][3] };
                    x[t] -= acc[0] + acc[1];
                }
            }
#pragma unroll
            for (int t = 0; t < MB; ++t) {
                T acc = x[t];
                if constexpr (sizeof(T) == 8) {
                    if (t > 0) Ld[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C]);
                    if (t > 4) Ld[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C + 4]);
                }
#pragma unroll
                for (int u = 0; u < 4 && u < t; ++u) acc -= Ld[t][0][u] * z[u];
#pragma unroll
                for (int u = 4; u < t; ++u) acc -= Ld[t][1][u - 4] * z[u];
                z[t] = acc * (t < 4 ? r0[t & 3] : r1[t & 3]);
            }
            *reinterpret_cast<v4_t *>(&Zt[par][i][0]) = v4_t{ z[0], z[1], z[2], z[3] };
            *reinterpret_cast<v4_t *>(&Zt[par][i][4]) = v4_t{ z[4], z[5], z[6], z[7] };
#pragma unroll
            for (int t = 0; t < MB; ++t) { Xs[C + t][i] = z[t]; zprev[t] = z[t]; }
            if (k == NSP) z_tail();
        }
        PROBE_STEP(k, 1);
        __syncthreads();
    }
#undef LDS_GROUP
}

}  // namespace pre3
