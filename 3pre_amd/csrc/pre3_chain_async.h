// pre3_chain_async.h -- the dependent chain of one 64-column Cholesky panel WITHOUT a workgroup barrier per pipeline step (round 6).
//
// chol_chain (pre3_chain.h) runs ten lock-step pipeline steps per panel, one s_barrier each: a step lasts as long as its slowest wave plus
// the barrier's round trip, ~1570 shader clocks, of which the factor wave -- the only wave every other one waits for -- works ~1000
// (tools/probe_chain2.hip).  Here every hand-off is a flag word in LDS instead: a producer writes its payload and then its progress
// counter (LDS operations of one wave complete in issue order, so a reader that has seen the counter sees the payload), a consumer reads
// the counter FIRST and the payload right behind it in the same batch of LDS reads -- when the producer is ahead, which it normally is,
// a hand-off costs no extra round trip.  The roles then run at their own pace:
//   D side:  factor wave F (sub-panel s needs Pn(s), published by the D workers one F step earlier)  <->  D workers (step k needs Y(k-1))
//   X side:  z wave (sub-panel s needs Xr(s), Ls through sub-panel s, Rs(s))  <->  X workers (step k needs Y(k-2) and Z(k-2))
// The D side never waits for the X side; the X side trails it by about two steps.
//
// fp32: the factor wave's lookahead  a[t] -= sum_u Y(s-1)[i][u] Y(s-1)[C+t][u]  (its own row against rows C..C+7 of the sub-panel it has just
// factored) runs on the matrix core as sixteen v_mfma_f32_4x4x1 with the A operand BROADCAST from one block of four lanes (CBSZ = 4,
// ABID = the block that holds rows C+4h .. C+4h+3): both operands are the wave's own registers, the 16 broadcast LDS reads of the
// lock-step form and the LDS write -> read round trip between two sub-panels are gone.  Lane l's result register i is
// sum_u y_l[u] * y_{C+4h+i}[u]: its own row -- exactly where the column loop wants it.  (The f32-input MFMA is an exact fma chain.)
// The z wave's lookahead takes its A operand from LDS in the same layout (lane l: row C + 4h + (l & 3)): 4 reads instead of 16.
//
// Waves: 0-3 D workers, 4-7 X workers, 8 z wave, 9 factor wave (the factor wave shares its SIMD with the dead D tile above the diagonal),
// wave 1 (the slot of the dead D tile) and any wave >= 10 run side(flags) once and join the barrier behind the chain.
#pragma once
#include <type_traits>
#include "pre3_chain.h"

namespace pre3 {

// progress counters (LDS words inside ChPipe): F / Z = sub-panels finished; D[w] / X[w] = pipeline steps finished by worker w (+2 / +1: see below)
enum { CHF_F = 0, CHF_Z = 1, CHF_D0 = 2, CHF_X0 = 6, CHF_DI0 = 10 /* + WV: D worker WV has its whole tile in registers (DInit's sources may be overwritten) */, CHF_N = 14 };

__device__ __forceinline__ unsigned cha_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void cha_store(unsigned *p, unsigned v, int lane)
{
    asm volatile("" ::: "memory");                   // the payload's LDS writes are issued before the counter's (and complete before it: one wave's LDS operations are in order)
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr int CHA_SPIN = 1 << 16;      // x (one LDS round trip + s_sleep 1): a few milliseconds; a hand-off inside one workgroup takes well under a microsecond
// wait until *p >= need (wave-uniform); false if the bound ran out (a hand-off that never comes: the caller raises the status flag and goes on)
// a blocking read of one counter, as ONE opaque instruction pair: around the builtin atomic load the compiler also waits for every vector-memory
// operation of the wave (s_waitcnt vmcnt(0) at the poll loop's head) -- the persistent kernel's publisher wave then stood ~900 clocks per poll behind its
// own write-through stores (tools/probe_cholp.py, round 6)
__device__ __forceinline__ unsigned cha_load_wait(const unsigned *p)
{
    unsigned v;
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned *)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// BUSY: no s_sleep between polls (the D workers: the factor wave's next sub-panel waits for what they publish)
template <bool BUSY = false>
__device__ __forceinline__ bool cha_wait(const unsigned *p, unsigned need)
{
    for (int spin = 0; spin < CHA_SPIN; ++spin) {
        if (__builtin_amdgcn_readfirstlane((int)cha_load_wait(p)) >= (int)need) return true;
        if (!BUSY) __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

#ifdef PRE3_PROBE_CHA
static __device__ unsigned long long g_cha[8 * 12 * 4];         // [role: 0 F, 1 z, 2 D0, 3 D2, 4 D3, 5 X0 .. 7 X2(sic: X2 = wave 6)][step + 1][slot]
#if PRE3_PROBE_CHA == 2        // lite: the factor and z waves' first and last stamps only (every stamp waits for the wave's LDS operations: the full set stretches the chain by a quarter)
#define CHA_STAMP(role, k, slot) do { if ((role) <= 1 && (((k) == 0 && (slot) == 0) || ((k) == 7 && (slot) == 3)) && (threadIdx.x & 63) == 0) g_cha[((role) * 12 + (k) + 1) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CHA_STAMP(role, k, slot) do { if ((threadIdx.x & 63) == 0) g_cha[((role) * 12 + (k) + 1) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define CHA_STAMP(role, k, slot)
#endif

#ifndef CHA_DBUSY
#define CHA_DBUSY true
#endif
#ifndef CHA_PREFETCH_POS
#define CHA_PREFETCH_POS 0  // where the factor wave requests the next sub-panel's columns: 0 in front of its own LDS writes (end of the step), 1 in front of the column loop (CHA_LA = 2: they are out a step earlier)
#endif
#ifndef CHA_PREFETCH
#define CHA_PREFETCH 1
#endif
#ifndef CHA_ABL
#define CHA_ABL 0           // ablations of the D workers (tools/probe_chain3.hip): 1 no MFMA, 2 no operand reads either, 3 no publish writes either, 4 no polling either
#endif
#ifndef CHA_ZWAVE
#define CHA_ZWAVE 8
#endif
#ifndef CHA_ZBATCH
#define CHA_ZBATCH 1
#endif
#ifndef CHA_LA
#define CHA_LA 1            // sub-panels of lookahead the factor wave applies itself: the D workers publish sub-panel k + CHA_LA at step k
#endif
struct ChaNoSide { __device__ __forceinline__ void operator()(unsigned *) const {} };

// D-side matrix-core traits: 16 x 16 blocks in both precisions.  fp32 takes v_mfma_f32_16x16x4_f32 (8 passes) instead of the 32 x 32 x 2 form of
// the lock-step chain (16 passes): the two blocks that hold the NEXT sub-panel's columns are updated and published first -- the factor wave's
// next step waits for exactly those -- and blocks that are factored already or lie above the diagonal are skipped.
template <typename T> struct MfmaD;
template <> struct MfmaD<double> : Mfma<double> {};
template <> struct MfmaD<float> {
    static constexpr int BLK = 16, KS = 4, NREG = 4;
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ inline void mma(float a, float b, acc_t &c) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ inline int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
    static __device__ inline int col(int lane) { return lane & 15; }
    static __device__ inline int kk(int lane) { return lane >> 4; }
};

// D worker wave WV (tile (WV >> 1, WV & 1) of the diagonal block, read from Ls on entry: the caller's raw / pre-updated block), pipeline step k
// (-LA .. NSP-1-LA): D -= Y(k-1) Y(k-1)' where the tile still has columns >= 8(k+LA), sub-panel k+LA published as soon as its blocks are done
// DInit: where a D worker's 16 x 16 blocks come from.  Default: the diagonal block in Ls.  The persistent kernel's crit computes them there and then
// (D_{J+1} = A - L L' on the bf16 matrix cores, block column 0 first): sub-panels 0 and 1 are published -- and the factor wave starts -- while the
// second block column is still being multiplied.
struct ChaFromLs {
    static constexpr bool from_ls = true;
    template <typename ACC> __device__ __forceinline__ void operator()(ACC &, ACC &, bool, bool, int, int, int) const {}
};
template <typename T, int WV, typename DInit>
__device__ __forceinline__ void cha_dworker(ChSmem<T> &sm, const int lane, bool &bad, DInit &&dinit, const int nsp_eff)
{
    using M = MfmaD<T>;
    constexpr int NBLK = 32 / M::BLK, NJ = CH_MB / M::KS, MB = CH_MB, NSP = CH_NSP, w0 = (WV >> 1) * 32, w1 = (WV & 1) * 32;
    typedef T vk_t __attribute__((ext_vector_type(NJ), aligned(NJ * sizeof(T))));
    typedef typename M::acc_t acc_t;
    const int cl = M::col(lane), kq = M::kk(lane) * NJ;
    unsigned *fl = sm.pipe.fl;
    acc_t acc[NBLK][NBLK];
    constexpr bool from_ls = std::remove_reference<DInit>::type::from_ls;
    auto load_q = [&](const int q) {
        // (NBLK = 2: both row blocks of the block column at once -- DInit interleaves their matrix-core chains)
        const bool live0 = !(w0 + M::BLK - 1 < w1 + q * M::BLK), live1 = !(w0 + 2 * M::BLK - 1 < w1 + q * M::BLK);
        if constexpr (from_ls) {
#pragma unroll
            for (int p = 0; p < NBLK; ++p)
#pragma unroll
                for (int e = 0; e < M::NREG; ++e)
                    acc[p][q][e] = (p == 0 ? live0 : live1) ? sm.Ls[w0 + p * M::BLK + M::row(lane, e)][w1 + q * M::BLK + cl] : (T)0;      // (a block above the diagonal is never read)
        } else dinit(acc[0][q], acc[1][q], live0, live1, w0, w1 + q * M::BLK, lane);
    };
    load_q(0);
    if constexpr (from_ls || w1 != 0) { load_q(1); cha_store(fl + CHF_DI0 + WV, 1u, lane); }      // (a tile whose first columns are sub-panel 0's publishes them before it takes up its second block column)
    // (the tile is in registers before this wave publishes anything, and the factor wave writes a column of Ls only after it has been published)
#pragma unroll
    for (int k = -CHA_LA; k <= NSP - 1 - CHA_LA; ++k) {
        const int c_first = MB * (k + CHA_LA);                             // first column the workers still own at this step; also the sub-panel to publish
        const bool upd = k >= 1 && w1 + 32 > c_first;
        const bool pub = c_first >= w1 && c_first < w1 + 32;
        if constexpr (!from_ls && w1 == 0) { if (k == 1 - CHA_LA + 1) { load_q(1); cha_store(fl + CHF_DI0 + WV, 1u, lane); } }      // (behind the publication of sub-panels 0 and 1 = the columns of block column 0)
        if (!upd && !pub) continue;
        if (k + CHA_LA >= nsp_eff) continue;                                // (sub-panel k + LA is padding: nobody reads it)
        const int qs = pub ? (c_first - w1) / M::BLK : 0, c8 = pub ? (c_first - w1) % M::BLK : 0, par = (k + CHA_LA) % (CHA_LA + 1);
        auto publish = [&]() {
            if (CHA_ABL < 3 && cl >= c8 && cl < c8 + MB) {
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int e = 0; e < M::NREG; ++e) sm.pipe.Pn[par][w0 + p * M::BLK + M::row(lane, e)][cl - c8] = acc[p][qs][e];
            }
            cha_store(fl + CHF_D0 + WV, (unsigned)(k + CHA_LA + 1), lane);      // sub-panels 0 .. k + LA of this tile's rows are out
            CHA_STAMP(2 + (WV == 0 ? 0 : WV == 2 ? 1 : 2), k, 2);
        };
        if (upd) {
            const int C = MB * (k - 1);
            CHA_STAMP(2 + (WV == 0 ? 0 : WV == 2 ? 1 : 2), k, 0);
            if (CHA_ABL < 4) { if (!cha_wait<CHA_DBUSY>(fl + CHF_F, (unsigned)k)) bad = true; }
            CHA_STAMP(2 + (WV == 0 ? 0 : WV == 2 ? 1 : 2), k, 1);
            vk_t a[NBLK], bb[NBLK];
#pragma unroll
            for (int p = 0; p < NBLK; ++p) {
                a[p] = *reinterpret_cast<const vk_t *>(&sm.Ls[w0 + p * M::BLK + cl][C + kq]);
                bb[p] = *reinterpret_cast<const vk_t *>(&sm.Ls[w1 + p * M::BLK + cl][C + kq]);
            }
#pragma unroll
            for (int qi = 0; qi < NBLK; ++qi) {
                const int q = (qs + qi) % NBLK;                               // the block column to publish first
                const int c_lo = w1 + q * M::BLK > c_first ? w1 + q * M::BLK : c_first;
                if (w1 + (q + 1) * M::BLK > c_first) {                        // (some column of the block is still the workers')
#pragma unroll
                    for (int p = 0; p < NBLK; ++p) {
                        if (w0 + (p + 1) * M::BLK - 1 < c_lo) continue;       // (above the diagonal of what is left)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            if (CHA_ABL < 1) M::mma(-a[p][j], bb[q][j], acc[p][q]);
                    }
                }
                if (qi == 0 && pub) publish();
            }
        } else publish();
    }
}

// X worker wave WV, pipeline step k (0 .. NSP-1): X -= Y(k-2) Z(k-2) where the tile still has rows >= 8k, then publish rows 8k .. 8k+7.
// The counter goes up at EVERY live step: it also tells the z wave that Zt(k-2) has been consumed (its slot is rewritten two sub-panels later).
template <typename T, int WV, bool RELAX>
__device__ __forceinline__ void cha_xworker(ChSmem<T> &sm, typename ChW<T>::acc_t (&acc)[ChW<T>::NBLK][ChW<T>::NBLK], const int lane, bool &bad, const int nsp_eff)
{
    using M = Mfma<T>;
    typedef typename ChW<T>::vk_t vk_t;
    constexpr int NBLK = ChW<T>::NBLK, NJ = ChW<T>::NJ, MB = CH_MB, NSP = CH_NSP, w0 = (WV >> 1) * 32, w1 = (WV & 1) * 32;
    const int cl = M::col(lane), kq = M::kk(lane) * NJ;
    unsigned *fl = sm.pipe.fl;
#pragma unroll
    for (int k = 0; k <= NSP - 1; ++k) {
        if (k >= nsp_eff) continue;                                         // (rows 8k.. are padding: the right-hand side keeps them)
        if (!(w0 + 32 > MB * k)) {                                      // the tile's rows are all solved: nothing of Zt is read any more
            if (k == (w0 + 32) / MB) cha_store(fl + CHF_X0 + WV, 0x7fffffffu, lane);
            continue;
        }
        if (k >= 2) {
            const int C = MB * (k - 2), par = (k - 2) & 1;
            if (WV < 3) CHA_STAMP(5 + WV, k, 0);
            if (!cha_wait(fl + CHF_Z, (unsigned)(k - 1))) bad = true;  // Z(k-2) is out (and with it Y(k-2): the z wave has read sub-panel k-2 of Ls)
            vk_t a[NBLK], bb[NBLK];
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
                        if constexpr (sizeof(T) == 8) { if (w0 + (p + 1) * M::BLK <= MB * k) continue; }
                        M::mma(-a[p][j], bb[q][j], acc[p][q]);
                    }
        }
        const int Cr = MB * k, par = k & 1;
        if (Cr >= w0 && Cr < w0 + 32) {
            // the flag-driven chain keeps Xr transposed, [column i][row t of the sub-panel]: a lane of the 32 x 32 accumulator layout holds four consecutive
            // t of one column -- ONE 16-byte write instead of four, and the z wave reads its column's eight values in two reads instead of eight
            T (*XrT)[NB][CH_MB] = reinterpret_cast<T (*)[NB][CH_MB]>(&sm.pipe.Xr[0][0][0]);
            const int r8 = Cr - w0;
            if constexpr (sizeof(T) == 4) {
                typedef T v4_t __attribute__((ext_vector_type(4), aligned(16)));
                constexpr int g = 0;
                (void)g;
                const int e0 = 4 * (r8 / 8);
                *reinterpret_cast<v4_t *>(&XrT[par][w1 + cl][4 * (lane >> 5)]) = v4_t{ acc[0][0][e0], acc[0][0][e0 + 1], acc[0][0][e0 + 2], acc[0][0][e0 + 3] };
            } else {
#pragma unroll
                for (int p = 0; p < NBLK; ++p)
#pragma unroll
                    for (int e = 0; e < M::NREG; ++e)
                        if ((p * M::BLK + M::row(0, e)) / MB * MB == r8) {
                            const int t = p * M::BLK + M::row(lane, e) - r8;
#pragma unroll
                            for (int q = 0; q < NBLK; ++q) XrT[par][w1 + q * M::BLK + cl][t] = acc[p][q][e];
                        }
            }
        }
        cha_store(fl + CHF_X0 + WV, (unsigned)(k + 1), lane);
        if (WV < 3) CHA_STAMP(5 + WV, k, 2);
    }
    if (w0 + 32 > MB * (NSP - 1)) cha_store(fl + CHF_X0 + WV, 0x7fffffffu, lane);
}

// On entry Ls holds the (fully updated) diagonal block and Xs the workgroup's X block (hasX), or the worker waves carry their tiles in `acc`
// (acc_loaded) / build them in worker_init; on exit Ls holds L_JJ and Xs[a][i] the solved block L_JJ^-1 X.  Barriers: one on entry (the counters
// are reset in front of it), one on exit.  side(fl): run once by wave 1 and every wave >= 10 (the persistent kernel's publisher / fetcher waves), with the
// counters to poll: fl[CHF_F] = columns 0 .. 8 fl - 1 of L are final in Ls, fl[CHF_Z] = rows 0 .. 8 fl - 1 of the solved block are final in Xs.
// XTRI: the X block starts as the identity (the persistent kernel's crit: M = L^-1 is lower triangular): its tile above the diagonal stays zero, no wave works on it.
template <typename T, bool RELAX = false, bool XTRI = false, typename Side = ChaNoSide, typename WInit = ChNoInit, typename DInit = ChaFromLs>
__device__ __forceinline__ void chol_chain_async(ChSmem<T> &sm, typename ChW<T>::acc_t (&acc)[ChW<T>::NBLK][ChW<T>::NBLK], const bool acc_loaded,
                                                 const bool hasX, bool &bad, Side &&side = ChaNoSide{}, WInit &&worker_init = ChNoInit{}, DInit &&dinit = ChaFromLs{},
                                                 const int nsp_eff = CH_NSP /* wave-uniform: sub-panels >= nsp_eff are padding (identity columns of the block against zero rows of the right-hand side) and are skipped; the counters end at NSP all the same */)
{
    constexpr int MB = CH_MB, NSP = CH_NSP;
    typedef T v4_t __attribute__((ext_vector_type(4), aligned(16)));
    typedef T T2 __attribute__((ext_vector_type(2)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    auto &Ls = sm.Ls; auto &Xs = sm.Xs; auto &Pn = sm.pipe.Pn; auto &Zt = sm.pipe.Zt; auto &Xr = sm.pipe.Xr; auto &Rs = sm.pipe.RsA;
    unsigned *fl = sm.pipe.fl;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int role = wave == 9 ? 0 : wave == CHA_ZWAVE ? 1 : (wave < 8 && wave != 1) ? 2 : 3;        // 0: factor wave, 1: z wave (the slot of the dead D tile: it shares the factor wave's SIMD, whose matrix core no 32 x 32 product occupies), 2: worker, 3: side wave (8, 10, 11)
    const bool worker = role == 2, xside = wave >= 4 && wave < 8;
    const int wv = wave & 3;
    const bool tile_live = worker && (xside ? (hasX && !(XTRI && wv == 1)) : wv != 1);
    if (tid < CHF_N) fl[tid] = ((tid >= CHF_X0 && !hasX) || (XTRI && tid == CHF_X0 + 1)) ? 0x7fffffffu : 0u;
    // X tiles: from Xs, or built by worker_init (acc_loaded).  D tiles are always read from Ls (by cha_dworker, in its own block layout): with
    // acc_loaded the D workers' worker_init must leave the block in Ls (each wave its own tile: no barrier in between)
    if (worker && xside && !acc_loaded) {
        if (wv == 0) ch_worker_load<T, 0, true>(sm, acc, lane, tile_live);
        else if (wv == 1) ch_worker_load<T, 1, true>(sm, acc, lane, tile_live);
        else if (wv == 2) ch_worker_load<T, 2, true>(sm, acc, lane, tile_live);
        else ch_worker_load<T, 3, true>(sm, acc, lane, tile_live);
    } else if (worker && acc_loaded) {
        worker_init(acc, xside, wv);
    }
    // (raw barriers: only this wave's LDS operations are waited for.  __syncthreads() would also wait for every vector-memory operation in flight --
    //  the persistent kernel's publisher wave arrives with write-through stores on their way, 1-1.4 us each way)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                     // counters reset; the raw tiles are in registers (the hand-off buffers alias the prologue's operand tiles)
#ifdef CHA_F_ONLY
    if (role != 0) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); return; }
#endif
#ifdef CHA_NO_X
    if (role == 1 || xside) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); return; }
#endif
    if (worker) {
        if (tile_live) {
            if (xside) {
                if (wv == 0) cha_xworker<T, 0, RELAX>(sm, acc, lane, bad, nsp_eff);
                else if (wv == 1) cha_xworker<T, 1, RELAX>(sm, acc, lane, bad, nsp_eff);
                else if (wv == 2) cha_xworker<T, 2, RELAX>(sm, acc, lane, bad, nsp_eff);
                else cha_xworker<T, 3, RELAX>(sm, acc, lane, bad, nsp_eff);
            } else {
                if (wv == 0) cha_dworker<T, 0>(sm, lane, bad, dinit, nsp_eff);
                else if (wv == 2) cha_dworker<T, 2>(sm, lane, bad, dinit, nsp_eff);
                else cha_dworker<T, 3>(sm, lane, bad, dinit, nsp_eff);
            }
        }
    } else if (role == 0) {
        // ---- factor wave: lane = row i of the L block
        __builtin_amdgcn_s_setprio(3);
        const int i = lane;
        T yprev[MB];
#pragma unroll
        for (int t = 0; t < MB; ++t) yprev[t] = (T)0;
        v4_t pv0 = v4_t{ (T)0, (T)0, (T)0, (T)0 }, pv1 = pv0;       // prefetched columns of the next sub-panel and the counters read in front of them
        unsigned pf0 = 0, pf1 = 0;
        T yprev2[MB];
#pragma unroll
        for (int t = 0; t < MB; ++t) yprev2[t] = (T)0;
#pragma unroll
        for (int k = 0; k < NSP; ++k) {
            if (k >= nsp_eff) continue;
            const int C = MB * k, par = k % (CHA_LA + 1), parn = (k + 1) % (CHA_LA + 1);
            T2 a2[MB / 2];
            T y[MB], rsv[MB];
            // the lookahead needs nothing but this wave's registers (fp32): it is issued before the published columns are waited for
            f4_t la[2] = { f4_t{ 0.f, 0.f, 0.f, 0.f }, f4_t{ 0.f, 0.f, 0.f, 0.f } };
            if constexpr (sizeof(T) == 4) {
                if (k > 0) {
#pragma unroll
                    for (int u = 0; u < MB; ++u) {
                        // (ABID is an immediate: C is a compile-time constant in the unrolled loop)
                        switch (C / 4) {
#define CHA_LA1(Q) case Q: la[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(yprev[u], yprev[u], la[0], 4, Q, 0); la[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(yprev[u], yprev[u], la[1], 4, Q + 1, 0); break;
                        CHA_LA1(2) CHA_LA1(4) CHA_LA1(6) CHA_LA1(8) CHA_LA1(10) CHA_LA1(12) CHA_LA1(14)
#undef CHA_LA1
                        default: break;
                        }
                    }
                }
                if (CHA_LA == 2 && k > 1) {
#pragma unroll
                    for (int u = 0; u < MB; ++u) {
                        switch (C / 4) {
#define CHA_LA2(Q) case Q: la[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(yprev2[u], yprev2[u], la[0], 4, Q, 0); la[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(yprev2[u], yprev2[u], la[1], 4, Q + 1, 0); break;
                        CHA_LA2(4) CHA_LA2(6) CHA_LA2(8) CHA_LA2(10) CHA_LA2(12) CHA_LA2(14)
#undef CHA_LA2
                        default: break;
                        }
                    }
                }
            }
            CHA_STAMP(0, k, 0);
            // the published columns (updated through sub-panel k-2): counters first, payload right behind them
            const unsigned *fd0 = fl + CHF_D0 + (C < 32 ? 0 : 3), *fd1 = fl + CHF_D0 + (C < 32 ? 2 : 3);
            v4_t v0 = pv0, v1 = pv1;
            v4_t Lh[MB][2];
            // (the previous step has requested this sub-panel's columns in front of its own LDS writes: if the D workers were ahead -- they normally
            //  are -- the step starts from registers)
            if (!(CHA_PREFETCH && k > 0 && __builtin_amdgcn_readfirstlane((int)pf0) >= k + 1 && __builtin_amdgcn_readfirstlane((int)pf1) >= k + 1)) {
                for (int spin = 0; ; ++spin) {
                    const unsigned f0v = cha_load(fd0), f1v = cha_load(fd1);
                    asm volatile("" ::: "memory");
                    v0 = *reinterpret_cast<const v4_t *>(&Pn[par][i][0]); v1 = *reinterpret_cast<const v4_t *>(&Pn[par][i][4]);
                    __builtin_amdgcn_sched_barrier(0);                      // (all four reads are in flight before the first one is waited for)
                    const int f0 = __builtin_amdgcn_readfirstlane((int)f0v), f1 = __builtin_amdgcn_readfirstlane((int)f1v);
#ifdef CHA_F_ONLY
                    break;
#endif
                    if (f0 >= k + 1 && f1 >= k + 1) break;
                    if (spin >= CHA_SPIN) { bad = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            asm volatile("" ::: "memory");
            a2[0] = T2{ v0[0], v0[1] }; a2[1] = T2{ v0[2], v0[3] }; a2[2] = T2{ v1[0], v1[1] }; a2[3] = T2{ v1[2], v1[3] };
            CHA_STAMP(0, k, 1);
            if (k > 0) {
                if constexpr (sizeof(T) == 4) {
                    a2[0] -= T2{ la[0][0], la[0][1] }; a2[1] -= T2{ la[0][2], la[0][3] }; a2[2] -= T2{ la[1][0], la[1][1] }; a2[3] -= T2{ la[1][2], la[1][3] };
                } else {
                    const T2 yp[4] = { T2{ yprev[0], yprev[1] }, T2{ yprev[2], yprev[3] }, T2{ yprev[4], yprev[5] }, T2{ yprev[6], yprev[7] } };
#pragma unroll
                    for (int t = 0; t < MB; ++t) {
                        // (fp64: the lookahead's multipliers are read where they are used -- sixty-four doubles at once do not fit the register budget;
                        //  the fences keep the reads from being hoisted as IR or gathered by the scheduler)
                        if (t & 1) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                        Lh[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB]); Lh[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB + 4]);
                        T2 s2 = yp[0] * T2{ Lh[t][0][0], Lh[t][0][1] };
                        s2 += yp[1] * T2{ Lh[t][0][2], Lh[t][0][3] };
                        s2 += yp[2] * T2{ Lh[t][1][0], Lh[t][1][1] };
                        s2 += yp[3] * T2{ Lh[t][1][2], Lh[t][1][3] };
                        a2[t >> 1][t & 1] -= s2[0] + s2[1];
                    }
                }
            }
            auto prefetch_next = [&]() {
                if (CHA_PREFETCH && k + 1 < NSP && k + 1 < nsp_eff) {
                    const int Cn = C + MB;
                    pf0 = cha_load(fl + CHF_D0 + (Cn < 32 ? 0 : 3)); pf1 = cha_load(fl + CHF_D0 + (Cn < 32 ? 2 : 3));
                    asm volatile("" ::: "memory");
                    pv0 = *reinterpret_cast<const v4_t *>(&Pn[parn][i][0]); pv1 = *reinterpret_cast<const v4_t *>(&Pn[parn][i][4]);
                    asm volatile("" ::: "memory");
                }
            };
            if (CHA_PREFETCH_POS == 1) prefetch_next();
            // right-looking inside the sub-panel, division-free on the dependent chain (as chol_chain)
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
            CHA_STAMP(0, k, 2);
            if (CHA_PREFETCH_POS == 0) prefetch_next();
            *reinterpret_cast<v4_t *>(&Ls[i][C]) = v4_t{ y[0], y[1], y[2], y[3] };
            *reinterpret_cast<v4_t *>(&Ls[i][C + 4]) = v4_t{ y[4], y[5], y[6], y[7] };
            if (i == 0) {
                *reinterpret_cast<v4_t *>(&Rs[k][0]) = v4_t{ rsv[0], rsv[1], rsv[2], rsv[3] };
                *reinterpret_cast<v4_t *>(&Rs[k][4]) = v4_t{ rsv[4], rsv[5], rsv[6], rsv[7] };
            }
            cha_store(fl + CHF_F, (unsigned)(k + 1), lane);
            CHA_STAMP(0, k, 3);
#pragma unroll
            for (int t = 0; t < MB; ++t) { yprev2[t] = yprev[t]; yprev[t] = y[t]; }
        }
        if (nsp_eff < NSP) cha_store(fl + CHF_F, (unsigned)NSP, lane);     // (the padding's columns of L are the block's own identity columns)
    } else if (role == 1) {
        if (hasX) {
            // ---- z wave: lane = column i of the workgroup's X block; sub-panel s: z <- L8^-1 (x - lookahead).
            // Software-pipelined against LDS latency (a round trip costs 250-400 clocks while the other waves use the LDS): the reads of phase B
            // (the 8 x 8 diagonal sub-block, Rs) travel while the lookahead runs on the matrix core, the reads of the NEXT sub-panel's phase A
            // (its published rows, the lookahead's operand, the counters) while the triangular solve runs.
            __builtin_amdgcn_s_setprio(2);
            const int i = lane;
            T (*XrT)[NB][CH_MB] = reinterpret_cast<T (*)[NB][CH_MB]>(&Xr[0][0][0]);
            T zprev[MB];
#pragma unroll
            for (int t = 0; t < MB; ++t) zprev[t] = (T)0;
            // phase A operands of the sub-panel about to start (prefetched) and the counters read in front of them
            v4_t xa = v4_t{ (T)0, (T)0, (T)0, (T)0 }, xb = xa, La[2][2] = { { xa, xa }, { xa, xa } }, Lh[MB][2];
            unsigned aF = 0, ax0 = 0, ax1 = 0, ax2 = 0, ax3 = 0;
            auto issue_a = [&](const int s2) {
                const int C = MB * s2, par = s2 & 1;
                aF = cha_load(fl + CHF_F); ax0 = cha_load(fl + CHF_X0 + 0); ax1 = cha_load(fl + CHF_X0 + 1); ax2 = cha_load(fl + CHF_X0 + 2); ax3 = cha_load(fl + CHF_X0 + 3);
                asm volatile("" ::: "memory");
                xa = *reinterpret_cast<const v4_t *>(&XrT[par][i][0]); xb = *reinterpret_cast<const v4_t *>(&XrT[par][i][4]);
                if (s2 > 0) {
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            La[h][0] = *reinterpret_cast<const v4_t *>(&Ls[C + 4 * h + (lane & 3)][C - MB]); La[h][1] = *reinterpret_cast<const v4_t *>(&Ls[C + 4 * h + (lane & 3)][C - MB + 4]);
                        }
                    }
                }
                asm volatile("" ::: "memory");
            };
            if constexpr (sizeof(T) == 4) issue_a(0);
#pragma unroll
            for (int s2 = 0; s2 < NSP; ++s2) {
                if (s2 >= nsp_eff) continue;
                if constexpr (sizeof(T) == 8) issue_a(s2);          // (fp64: no prefetch across the solve -- the register budget)
                const int C = MB * s2, par = s2 & 1;
                T x[MB], z[MB];
                v4_t Ld[MB][2], r0, r1;
                CHA_STAMP(1, s2, 0);
                // phase A: Xr(s2) and the release of Zt's slot (X workers), rows C .. C+7 of Y(s2-1) (factor wave through sub-panel s2-1)
                for (int spin = 0; ; ++spin) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int fF = __builtin_amdgcn_readfirstlane((int)aF), needx = s2 + 1;
                    const int fx0 = __builtin_amdgcn_readfirstlane((int)ax0), fx1 = __builtin_amdgcn_readfirstlane((int)ax1), fx2 = __builtin_amdgcn_readfirstlane((int)ax2), fx3 = __builtin_amdgcn_readfirstlane((int)ax3);
                    if (fF >= s2 && fx0 >= needx && fx1 >= needx && fx2 >= needx && fx3 >= needx) break;
                    if (spin >= CHA_SPIN) { bad = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    issue_a(s2);
                }
                asm volatile("" ::: "memory");          // (nothing that reads the producers' payload moves above the counters' check)
                CHA_STAMP(1, s2, 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) { x[t] = xa[t]; x[4 + t] = xb[t]; }
                if constexpr (XTRI) {
                    if (s2 < NSP / 2 && i >= 32) {               // (the dead tile's rows: nothing was published)
#pragma unroll
                        for (int t = 0; t < MB; ++t) x[t] = (T)0;
                    }
                }
                // phase B's reads go out in front of the lookahead
                unsigned bF;
                auto issue_b = [&]() {
                    bF = cha_load(fl + CHF_F);
                    asm volatile("" ::: "memory");
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int t = 1; t < MB; ++t) {
                            Ld[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C]);
                            if (t > 4) Ld[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C + 4]);
                        }
                    }
                    r0 = *reinterpret_cast<const v4_t *>(&Rs[s2][0]); r1 = *reinterpret_cast<const v4_t *>(&Rs[s2][4]);
                    asm volatile("" ::: "memory");
                };
                issue_b();
                if (s2 > 0) {
                    if constexpr (sizeof(T) == 4) {
                        f4_t la[2] = { f4_t{ 0.f, 0.f, 0.f, 0.f }, f4_t{ 0.f, 0.f, 0.f, 0.f } };
#pragma unroll
                        for (int u = 0; u < MB; ++u)
#pragma unroll
                            for (int h = 0; h < 2; ++h) la[h] = __builtin_amdgcn_mfma_f32_4x4x1f32(La[h][u >> 2][u & 3], zprev[u], la[h], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < MB; ++t) x[t] -= la[t >> 2][t & 3];
                    } else {
                        const T2 zp[4] = { T2{ zprev[0], zprev[1] }, T2{ zprev[2], zprev[3] }, T2{ zprev[4], zprev[5] }, T2{ zprev[6], zprev[7] } };
#pragma unroll
                        for (int t = 0; t < MB; ++t) {
                            if (t & 1) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }      // (keeps the reads where they are used -- neither hoisted as IR nor sunk by the scheduler: batched up front they spill)
                            Lh[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB]); Lh[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C - MB + 4]);
                            T2 s2v = zp[0] * T2{ Lh[t][0][0], Lh[t][0][1] };
                            s2v += zp[1] * T2{ Lh[t][0][2], Lh[t][0][3] };
                            s2v += zp[2] * T2{ Lh[t][1][0], Lh[t][1][1] };
                            s2v += zp[3] * T2{ Lh[t][1][2], Lh[t][1][3] };
                            x[t] -= s2v[0] + s2v[1];
                        }
                    }
                }
                // phase B: the 8 x 8 diagonal sub-block of L and Rs(s2) (factor wave through sub-panel s2)
                for (int spin = 0; ; ++spin) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (__builtin_amdgcn_readfirstlane((int)bF) >= s2 + 1) break;
                    if (spin >= CHA_SPIN) { bad = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    issue_b();
                }
                asm volatile("" ::: "memory");
                CHA_STAMP(1, s2, 2);
                // the next sub-panel's phase A travels while the solve runs (its operand of the lookahead, rows C+8 .. C+15 of Y(s2), is final: see phase B)
                if constexpr (sizeof(T) == 4) { if (s2 + 1 < NSP && s2 + 1 < nsp_eff) issue_a(s2 + 1); }
                if constexpr (sizeof(T) == 4) {
                    // column-oriented: once z[u] is known every later row takes its term -- the dependent chain is one multiply and one fma per row
#pragma unroll
                    for (int u = 0; u < MB; ++u) {
                        z[u] = x[u] * (u < 4 ? r0[u & 3] : r1[u & 3]);
#pragma unroll
                        for (int t = u + 1; t < MB; ++t) x[t] -= (u < 4 ? Ld[t][0][u] : Ld[t][1][u - 4]) * z[u];
                    }
                } else {
                    // fp64: row by row, a row's multipliers read where they are used (the same terms in the same order)
#pragma unroll
                    for (int t = 0; t < MB; ++t) {
                        T accz = x[t];
                        if (t & 1) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                        if (t > 0) Ld[t][0] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C]);
                        if (t > 4) Ld[t][1] = *reinterpret_cast<const v4_t *>(&Ls[C + t][C + 4]);
#pragma unroll
                        for (int u = 0; u < 4 && u < t; ++u) accz -= Ld[t][0][u] * z[u];
#pragma unroll
                        for (int u = 4; u < t; ++u) accz -= Ld[t][1][u - 4] * z[u];
                        z[t] = accz * (t < 4 ? r0[t & 3] : r1[t & 3]);
                    }
                }
                *reinterpret_cast<v4_t *>(&Zt[par][i][0]) = v4_t{ z[0], z[1], z[2], z[3] };
                *reinterpret_cast<v4_t *>(&Zt[par][i][4]) = v4_t{ z[4], z[5], z[6], z[7] };
#pragma unroll
                for (int t = 0; t < MB; ++t) { Xs[C + t][i] = z[t]; zprev[t] = z[t]; }
                cha_store(fl + CHF_Z, (unsigned)(s2 + 1), lane);
                CHA_STAMP(1, s2, 3);
            }
            if (nsp_eff < NSP) cha_store(fl + CHF_Z, (unsigned)NSP, lane);
        }
    } else {
        side(fl);
    }
    if (role <= 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace pre3
