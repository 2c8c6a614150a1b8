// pre3_match.hip -- descriptor matching (sift/siftmatch.c:83-132) and kNearestNeighbors.m:29-39 on gfx950.
//
// Exactness contract: the reference accumulates (a-b)^2 bin by bin in the promoted class (double, float,
// int, int) with a separate multiply and add, keeps the first index on ties, and applies Lowe's ratio test
// in float.  (best, second best, first arg-best) of a sequential scan is order independent as a multiset
// statistic, so it can be reduced in parallel and merged across database shards.
//
//   * float / double classes: one lane per (query, database) pair walks the 128 bins in the reference's
//     order with contraction off -> bit-identical distances.
//   * uint8 / int8 classes: distances are integers; d2 = |a|^2 + |b|^2 - 2 a.b is evaluated exactly in
//     int32 on the matrix cores (v_mfma_i32_32x32x32_i8; uint8 is re-centred by -128, which leaves a-b
//     unchanged) with the best/second-best scan fused into the epilogue.
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "pre3_internal.h"
#include "pre3_geomdev.h"

namespace pre3 {

// generic exact kernel: block = one query column; lanes stride over database columns.
// L1: ND x K1, L2: ND x K2 column-major.  Outputs per query: best, second (as double), arg (global index).
template <typename T, typename ACC>
__global__ __launch_bounds__(256) void k_match_exact(int ND, int K2, const T *__restrict__ L1, const T *__restrict__ L2, int k2_offset,
                                                     double *__restrict__ obest, double *__restrict__ osecond, int32_t *__restrict__ oarg,
                                                     const int32_t *__restrict__ qidx = nullptr, const int32_t *__restrict__ nq = nullptr)
{
#pragma clang fp contract(off)
    extern __shared__ unsigned char smem_raw[];
    ACC *q = reinterpret_cast<ACC *>(smem_raw);
    const int k1 = blockIdx.x, tid = threadIdx.x;
    if (nq && k1 >= *nq) return;                       // IC search: the query count is only known on the device
    const size_t qcol = qidx ? (size_t)qidx[k1] : (size_t)k1;
    for (int b = tid; b < ND; b += blockDim.x) q[b] = (ACC)L1[qcol * ND + b];
    __syncthreads();
    ACC best = acc_max<ACC>(), second = acc_max<ACC>();
    int bk = -1;
    for (int k2 = tid; k2 < K2; k2 += blockDim.x) {
        const T *b = L2 + (size_t)k2 * ND;
        ACC acc = 0;
        for (int bin = 0; bin < ND; ++bin) {
            ACC delta = q[bin] - (ACC)b[bin];
            ACC sq = delta * delta;
            acc = acc + sq;
        }
        push3(best, second, bk, acc, k2);
    }
    // wave reduction then block reduction
    for (int o = 32; o > 0; o >>= 1) {
        ACC ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
        int ok = __shfl_xor(bk, o, 64);
        merge3(best, second, bk, ob, os, ok);
    }
    __shared__ int sk[4];
    __shared__ ACC sbb[4], sss[4];
    int wv = tid >> 6;
    if ((tid & 63) == 0) { sbb[wv] = best; sss[wv] = second; sk[wv] = bk; }
    __syncthreads();
    if (tid == 0) {
        ACC B = sbb[0], S2 = sss[0]; int K = sk[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) merge3(B, S2, K, sbb[w], sss[w], sk[w]);
        obest[k1] = (double)B; osecond[k1] = (double)S2; oarg[k1] = K < 0 ? -1 : K + k2_offset;
    }
}

// ------------------------------------------------------------------------------------------------
// Tiled exact kernel for the float classes (siftmatch.c:97-116 keeps every pair's accumulation order: bins 0..ND-1, one
// subtract, one multiply, one add each, contraction off -> bit-identical squared distances).  64 queries x 64 database
// columns per workgroup, 4 x 4 pairs per lane, both operand tiles staged through LDS in 32-bin chunks (bin-major, rows
// padded to 68 so that the 16-byte reads are aligned and conflict-free).  Per (query, column tile) the lanes scan their
// four columns in index order and merge across the tile with the order-independent (best, second, first-arg) statistic;
// k_match_reduce_f merges the column tiles.  ~20x (fp32) / 40x (fp64) faster than the one-query-per-block form at 4096^2.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_match_exact_tiled(int ND, int K1, int K2, const T *__restrict__ L1, const T *__restrict__ L2,
                                                           T *__restrict__ pbest, T *__restrict__ psecond, int32_t *__restrict__ parg, int ntile_n,
                                                           const int32_t *__restrict__ qidx = nullptr, const int32_t *__restrict__ nq = nullptr)
{
#pragma clang fp contract(off)
    constexpr int TQ = 64, TK = 64, CB = 32, LDP = 68;
    __shared__ __attribute__((aligned(16))) T Qs[CB][LDP];
    __shared__ __attribute__((aligned(16))) T Bs[CB][LDP];
    typedef T v4_t __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, tq = tid >> 4, tk = tid & 15;
    const int q0 = blockIdx.y * TQ, k0 = blockIdx.x * TK;
    const int nQ = nq ? *nq : K1;                          // IC search: the query count is only known on the device (K1 bounds the grid)
    if (q0 >= nQ) return;
    T acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (T)0;
    const int lq = tid >> 2, lb = (tid & 3) * 8;           // loader: descriptor lq of the tile, bins lb..lb+7 of the chunk
    const size_t qcol = q0 + lq < nQ ? (qidx ? (size_t)qidx[q0 + lq] : (size_t)(q0 + lq)) : 0;
    for (int c0 = 0; c0 < ND; c0 += CB) {
        T qv[8], bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int bin = c0 + lb + e;
            qv[e] = (q0 + lq < nQ && bin < ND) ? L1[qcol * ND + bin] : (T)0;
            bv[e] = (k0 + lq < K2 && bin < ND) ? L2[(size_t)(k0 + lq) * ND + bin] : (T)0;
        }
        __syncthreads();                                    // the previous chunk has been consumed
#pragma unroll
        for (int e = 0; e < 8; ++e) { Qs[lb + e][lq] = qv[e]; Bs[lb + e][lq] = bv[e]; }
        __syncthreads();
#pragma unroll 8
        for (int bin = 0; bin < CB; ++bin) {
            const v4_t q4 = *reinterpret_cast<const v4_t *>(&Qs[bin][tq * 4]);
            const v4_t b4 = *reinterpret_cast<const v4_t *>(&Bs[bin][tk * 4]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const T delta = q4[a] - b4[b];
                    const T sq = delta * delta;
                    acc[a][b] = acc[a][b] + sq;
                }
        }
    }
    // per query: this lane's four columns in increasing index, then the 16 lanes of the row group
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        T best = acc_max<T>(), second = acc_max<T>();
        int bk = -1;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k2 = k0 + tk * 4 + b;
            if (k2 < K2) push3(best, second, bk, acc[a][b], k2);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const T ob = __shfl_xor(best, o, 16), os = __shfl_xor(second, o, 16);
            const int ok = __shfl_xor(bk, o, 16);
            merge3(best, second, bk, ob, os, ok);
        }
        const int q = q0 + tq * 4 + a;
        if (tk == 0 && q < nQ) {
            const size_t o = (size_t)blockIdx.x * K1 + q;             // [column tile][query]: the reduce kernel reads it coalesced
            pbest[o] = best; psecond[o] = second; parg[o] = bk;
        }
    }
}

// sixteen lanes per query, each merging every sixteenth column tile with its loads issued 8 at a time (the merge is a dependent chain:
// one tile per load latency cost 42 us at 64 tiles), then four shuffle merges; launch with 16 * K1 threads
template <typename T>
__global__ __launch_bounds__(256) void k_match_reduce_f(int K1, int ntile_n, const T *__restrict__ pbest, const T *__restrict__ psecond, const int32_t *__restrict__ parg,
                                 int k2_offset, double *__restrict__ obest, double *__restrict__ osecond, int32_t *__restrict__ oarg)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int k1 = g >> 4, sub = g & 15;
    const bool live = k1 < K1;
    T best = acc_max<T>(), second = acc_max<T>();
    int bk = -1;
    if (live) {
        for (int t0 = sub; t0 < ntile_n; t0 += 128) {
            T vb[8], vs[8];
            int va[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 16 * u;
                const size_t o = (size_t)(t < ntile_n ? t : 0) * K1 + k1;
                vb[u] = pbest[o]; vs[u] = psecond[o]; va[u] = t < ntile_n ? parg[o] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) merge3(best, second, bk, vb[u], vs[u], va[u]);
        }
    }
#pragma unroll
    for (int o = 1; o <= 8; o <<= 1) {
        const T ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
        const int ok = __shfl_xor(bk, o, 64);
        merge3(best, second, bk, ob, os, ok);
    }
    if (live && sub == 0) { obest[k1] = (double)best; osecond[k1] = (double)second; oarg[k1] = bk < 0 ? -1 : bk + k2_offset; }
}

// ------------------------------------------------------------------------------------------------
// int8 MFMA distance kernel.  A (queries) and B (database) are int8, ND padded to a multiple of 32 with
// zeros; norms precomputed.  Tile: 128 queries x 128 database columns per workgroup (4 waves, 2x2 of 64x64,
// each 2x2 MFMA 32x32 blocks); K loop over ND in steps of 32 (one v_mfma_i32_32x32x32_i8 per block per step).
// Epilogue: d2 = na + nb - 2*dot, per-row scan over this tile's 128 columns, merged across column tiles by
// a second pass (k_match_reduce) over the per-tile partials.
// Operand layout for v_mfma_i32_32x32x32_i8: lane l holds 16 consecutive k (int8x16 = 4 VGPRs) of
// row/col (l&31), k-group (l>>5): k = 16*(l>>5) + j.
// ------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_match_i8_mfma(int NDp, int K1, int K2, const int8_t *__restrict__ A, const int8_t *__restrict__ B,
                                                       const int *__restrict__ na, const int *__restrict__ nb, int ntile_n,
                                                       int *__restrict__ pbest, int *__restrict__ psecond, int *__restrict__ parg)
{
    // A: K1p x NDp row-major (one descriptor per row), B: K2p x NDp row-major; both padded to 128 rows.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int I0 = blockIdx.y * 128, J0 = blockIdx.x * 128;
    v16i acc[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[p][q][e] = 0;
    const int r = lane & 31, kg = lane >> 5;
    if (NDp == 128) {
        // SIFT's 128 bins: all 16 fragment loads of the tile are issued before the first MFMA (one memory latency instead of four)
        v4i av[4][2], bv[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                av[ks][p] = *reinterpret_cast<const v4i *>(A + (size_t)(I0 + wi * 64 + p * 32 + r) * 128 + ks * 32 + 16 * kg);
                bv[ks][p] = *reinterpret_cast<const v4i *>(B + (size_t)(J0 + wj * 64 + p * 32 + r) * 128 + ks * 32 + 16 * kg);
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks][p], bv[ks][q], acc[p][q], 0, 0, 0);
    } else
    for (int k0 = 0; k0 < NDp; k0 += 32) {
        v4i av[2], bv[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            av[p] = *reinterpret_cast<const v4i *>(A + (size_t)(I0 + wi * 64 + p * 32 + r) * NDp + k0 + 16 * kg);
            bv[p] = *reinterpret_cast<const v4i *>(B + (size_t)(J0 + wj * 64 + p * 32 + r) * NDp + k0 + 16 * kg);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[p], bv[q], acc[p][q], 0, 0, 0);
    }
    // epilogue: each lane holds, per block, column (lane&31) and 16 rows.  Stage d2 through LDS (64 rows at a time); then wave q scans
    // columns [32q, 32q+32) of every row in increasing column order -- one lane per row, conflict-free (row stride 129) -- and writes its
    // own partial: 4 partials per 128-column tile, stored [partial][query] so that the merge kernel reads them coalesced.
    __shared__ int sd[64][129];
    const int K1p = gridDim.y * 128;
    // |a|^2 is constant along a row: the scan orders nb - 2 a.b and the row's norm is added to (best, second) when they are written
    const int nbq[2] = { nb[J0 + wj * 64 + (lane & 31)], nb[J0 + wj * 64 + 32 + (lane & 31)] };
    for (int half = 0; half < 2; ++half) {
        const int na_row = na[I0 + half * 64 + lane];
        if (wi == half) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        int row = p * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        int col = wj * 64 + q * 32 + (lane & 31);
                        sd[row][col] = nbq[q] - 2 * acc[p][q][e];
                    }
        }
        __syncthreads();
        {
            const int row = lane, c0 = wave * 32;
            int best = 0x7fffffff, second = 0x7fffffff, bk = -1;
            const int ncol = K2 - J0 < 128 ? K2 - J0 : 128;
            const int cend = c0 + 32 < ncol ? c0 + 32 : ncol;
            for (int cidx = c0; cidx < cend; ++cidx) push3(best, second, bk, sd[row][cidx], J0 + cidx);
            const size_t o = (size_t)(blockIdx.x * 4 + wave) * K1p + I0 + half * 64 + row;
            pbest[o] = best == 0x7fffffff ? best : best + na_row; psecond[o] = second == 0x7fffffff ? second : second + na_row; parg[o] = bk;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// int8 MFMA distance kernel, query-per-lane form (the default for 128-bin descriptors).
// The operands are swapped with respect to k_match_i8_mfma: the DATABASE block is the A operand and the QUERY block the B operand of
// v_mfma_i32_16x16x64_i8, so the accumulator's lane index (column l & 15) is the query and its 4 registers are 4 database columns
// (rows 4(l>>4) + e).  A lane therefore runs siftmatch.c:110-116's scan over ITS OWN registers:
//     d = nb - 2 a.b ; arg' = d < best ? e : arg ; second' = med3(best, d, second) ; best' = min(best, d)        (5 vector ops per pair)
// with no LDS round trip per tile, no serial one-lane-per-row walk and no second launch.
// Workgroup = 16 waves that share ONE block of 16 queries (its fragments, all 128 bins, stay in 8 VGPRs for the kernel's life) and split
// the database between them (wave w takes 16-column blocks w, w+16, ...; the next three blocks' fragments are in flight behind the current
// block's MFMAs, four register sets in rotation); operands are pre-packed in fragment order (k_pack_i8_v), so a fragment load is 1 KB
// contiguous.  At the end the four lane groups of a query (interleaved database rows) and the 16 waves
// are merged with the order-independent (best, second, first-arg) statistic through 3 KB of LDS.  4096 queries = 256 workgroups: one
// per CU.  Measured at 4096 x 4096 x 128 (tools/match_ab.py): 11.4 us against 25.6 us for the row-scan form; of that 2.6 us is an empty
// launch, ~3 us the scan's vector instructions, ~1 us the matrix pipe, the rest the stream of the database through L2 (every workgroup
// reads all of it: K2 x 128 B at ~64 B/clk/CU) and the prologue's dependent loads.  (Tried: 32-query workgroups on the 32x32x32 MFMA with
// the database split between workgroups + a ticket merge -- 15-20 us; norms in LDS -- no change; three blocks in flight -- this form.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int med3i(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
constexpr int I8_NONE = 0x3fffffff;       // = I8_NONE_NORM, the norm the pack kernels give padded database rows: never below a real distance (<= 128 * 255^2)
constexpr int I8Q_WAVES = 16, I8Q_Q = 16;

__global__ __launch_bounds__(64 * I8Q_WAVES) void k_match_i8_q(int K1, int K2p, const int8_t *__restrict__ Q, const int8_t *__restrict__ D,
                                                     const int *__restrict__ nq, const int *__restrict__ nd, int k2_offset,
                                                     double *__restrict__ obest, double *__restrict__ osecond, int32_t *__restrict__ oarg)
{
    __shared__ int mg[3 * I8Q_WAVES * I8Q_Q];                              // merge area [best | second | arg][16 waves][16 queries]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int nblk = K2p / 16;
    // Q and D are in fragment order (k_pack_i8_v, frag = 1): [block of 16][k-step of 64 bins][lane] x 16 bytes
    v4i fq[2], fa[2], fb[2], fc[2], fd[2], na_, nb_, nc_, nd_, nq_;
    auto load = [&](const int8_t *base, int blk, v4i (&f)[2], v4i &n, const int *norms) {
        const v4i *src = reinterpret_cast<const v4i *>(base) + (size_t)blk * 128 + lane;
        f[0] = src[0]; f[1] = src[64];
        n = *reinterpret_cast<const v4i *>(norms + blk * 16 + 4 * g);          // norms of this lane's 4 rows (16 lanes share an address)
    };
    load(Q, blockIdx.x, fq, nq_, nq);
    if (wave < nblk) load(D, wave, fa, na_, nd);
    int best = 0x7fffffff, second = 0x7fffffff, bk = -1, be = 0;
    auto scan = [&](int blk, const v4i (&f)[2], const v4i &n) {
        v4i acc = { 0, 0, 0, 0 };
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[0], fq[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(f[1], fq[1], acc, 0, 0, 0);
        const int best0 = best;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int d = n[e] - 2 * acc[e];                           // |b|^2 - 2 a.b (the query's norm is added once at the end)
            be = d < best ? e : be;                                    // strict: the first index wins ties (siftmatch.c:110); e is an inline constant
            second = med3i(best, d, second);                           // best <= second always: the median is the new second best
            best = d < best ? d : best;
        }
        bk = best < best0 ? blk : bk;                                  // the block the running best lives in (its row is rebuilt from (bk, be) at the end)
    };
    // (the prefetches are unconditional -- past the end they re-read the wave's last block -- because a branch around a load makes
    //  the compiler's s_waitcnt cover the not-taken path, i.e. wait for the prefetch itself)
    if (wave < nblk) {
        // four register sets in rotation, three blocks in flight: one L2 round trip is longer than three blocks' worth of MFMA + scan
        const int last = wave + (nblk - 1 - wave) / I8Q_WAVES * I8Q_WAVES;
        auto at = [&](int b) { return b < last ? b : last; };
        load(D, at(wave + I8Q_WAVES), fb, nb_, nd);
        load(D, at(wave + 2 * I8Q_WAVES), fc, nc_, nd);
        for (int blk = wave; blk < nblk; blk += 4 * I8Q_WAVES) {
            load(D, at(blk + 3 * I8Q_WAVES), fd, nd_, nd);
            scan(blk, fa, na_);
            if (blk + I8Q_WAVES >= nblk) break;
            load(D, at(blk + 4 * I8Q_WAVES), fa, na_, nd);
            scan(blk + I8Q_WAVES, fb, nb_);
            if (blk + 2 * I8Q_WAVES >= nblk) break;
            load(D, at(blk + 5 * I8Q_WAVES), fb, nb_, nd);
            scan(blk + 2 * I8Q_WAVES, fc, nc_);
            if (blk + 3 * I8Q_WAVES >= nblk) break;
            load(D, at(blk + 6 * I8Q_WAVES), fc, nc_, nd);
            scan(blk + 3 * I8Q_WAVES, fd, nd_);
        }
    }
    bk = bk < 0 ? -1 : bk * 16 + 4 * g + be;                             // database row of the best: block, lane group, register
    // the other three lane groups hold the interleaved database rows of the same query
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const int ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64), ok = __shfl_xor(bk, o, 64);
        merge3(best, second, bk, ob, os, ok);
    }
    if (g == 0) { mg[wave * 16 + lane] = best; mg[256 + wave * 16 + lane] = second; mg[512 + wave * 16 + lane] = bk; }
    __syncthreads();
    const int q = blockIdx.x * I8Q_Q + tid;
    if (tid < I8Q_Q && q < K1) {
        int B = mg[tid], S2 = mg[256 + tid], Kk = mg[512 + tid];
#pragma unroll
        for (int w = 1; w < I8Q_WAVES; ++w) merge3(B, S2, Kk, mg[w * 16 + tid], mg[256 + w * 16 + tid], mg[512 + w * 16 + tid]);
        // padded database rows carry the norm I8_NONE: a best / second at or above it means "no such column"
        const int qn = nq[q];
        const bool hb = Kk >= 0 && B < I8_NONE / 2, hs = S2 < I8_NONE / 2;
        obest[q] = hb ? (double)(B + qn) : 2147483647.0; osecond[q] = hs ? (double)(S2 + qn) : 2147483647.0; oarg[q] = hb ? Kk + k2_offset : -1;
    }
}

// sixteen lanes per query, each merging every sixteenth partial, loads issued 8 at a time (the merge is a dependent chain: one partial
// per load latency was 0.46 us each), then four shuffle merges
__global__ __launch_bounds__(256) void k_match_reduce_i32(int K1, int K1p, int npart, const int *__restrict__ pbest, const int *__restrict__ psecond,
                                   const int *__restrict__ parg, int k2_offset, double *__restrict__ obest, double *__restrict__ osecond,
                                   int32_t *__restrict__ oarg)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int k1 = g >> 4, sub = g & 15;
    const bool live = k1 < K1;
    int best = 0x7fffffff, second = 0x7fffffff, bk = -1;
    if (live) {
        for (int t0 = sub; t0 < npart; t0 += 128) {
            int vb[8], vs[8], va[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 16 * u;
                const size_t o = (size_t)(t < npart ? t : 0) * K1p + k1;
                vb[u] = pbest[o]; vs[u] = psecond[o]; va[u] = t < npart ? parg[o] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) merge3(best, second, bk, vb[u], vs[u], va[u]);
        }
    }
#pragma unroll
    for (int o = 1; o <= 8; o <<= 1) {
        const int ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64), ok = __shfl_xor(bk, o, 64);
        merge3(best, second, bk, ob, os, ok);
    }
    if (live && sub == 0) { obest[k1] = (double)best; osecond[k1] = (double)second; oarg[k1] = bk < 0 ? -1 : bk + k2_offset; }
}

constexpr int I8_NONE_NORM = 0x3fffffff;
// pack ND x K (column-major, one descriptor per column) uint8/int8 into K_pad x NDp int8 rows (+ norms): one thread per 16 bins (one
// 16-byte store), the NDp/16 threads of a descriptor are neighbours in a wave and add up the norm with shuffles (NDp/16 is a power
// of two <= 64 here: NDp is a multiple of 32; other shapes take the generic one-thread-per-descriptor path)
template <typename T>
__global__ __launch_bounds__(256) void k_pack_i8_v(int ND, int NDp, int K, int Kp, const T *__restrict__ L, int center, int8_t *__restrict__ out,
                                                    int *__restrict__ norm, int frag)
{
    const int cpd = NDp / 16;                                   // chunks per descriptor
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int kcol = g / cpd, ch = g - kcol * cpd;
    if (kcol >= Kp) return;
    int8_t v[16];
    int s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int b = ch * 16 + j;
        const int x = (kcol < K && b < ND) ? (int)L[(size_t)kcol * ND + b] - center : 0;
        v[j] = (int8_t)x; s += x * x;
    }
    // frag: the operand order of v_mfma_i32_16x16x64_i8 for NDp = 128 -- block of 16 descriptors, k-step of 64 bins, lane (descriptor & 15)
    // + 16 * (16-bin quarter of the k-step): a wave's fragment load is then 1 KB contiguous instead of 16 B out of each of 64 places
    const size_t o = frag ? ((size_t)((kcol >> 4) * 2 + (ch >> 2)) * 64 + (kcol & 15) + 16 * (ch & 3)) * 16 : (size_t)kcol * NDp + ch * 16;
    *reinterpret_cast<int4 *>(out + o) = *reinterpret_cast<const int4 *>(v);
    for (int o = cpd >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (ch == 0) norm[kcol] = kcol < K ? s : I8_NONE_NORM;          // padded rows: a norm no real distance reaches (k_match_i8_q)
}

template <typename T>
__global__ void k_pack_i8(int ND, int NDp, int K, int Kp, const T *__restrict__ L, int center, int8_t *__restrict__ out, int *__restrict__ norm)
{
    int kcol = blockIdx.x * blockDim.x + threadIdx.x;
    if (kcol >= Kp) return;
    int s = 0;
    for (int b = 0; b < NDp; ++b) {
        int v = (kcol < K && b < ND) ? (int)L[(size_t)kcol * ND + b] - center : 0;
        out[(size_t)kcol * NDp + b] = (int8_t)v;
        s += v * v;
    }
    norm[kcol] = kcol < K ? s : I8_NONE_NORM;
}

template <typename T>
static void launch_pack_i8(int ND, int NDp, int K, int Kp, const T *L, int center, int8_t *out, int *norm, bool frag = false)
{
    const int cpd = NDp / 16;
    if (cpd >= 2 && cpd <= 64 && (cpd & (cpd - 1)) == 0 && ((size_t)Kp * cpd) % 256 == 0)
        hipLaunchKernelGGL(k_pack_i8_v<T>, dim3((unsigned)((size_t)Kp * cpd / 256)), dim3(256), 0, 0, ND, NDp, K, Kp, L, center, out, norm, frag ? 1 : 0);
    else
        hipLaunchKernelGGL(k_pack_i8<T>, dim3(ceil_div(Kp, 64)), dim3(64), 0, 0, ND, NDp, K, Kp, L, center, out, norm);
}

// ------------------------------------------------------------------------------------------------
// float / double classes on the matrix cores: rank with a distance GEMM, then re-evaluate exactly.
//
// matching_sift_based.m:104-118 hands siftmatch DOUBLE descriptors.  The reference's result for a query is the order-independent statistic
// (best, second best counting duplicates, first arg-best) of the distances d_j accumulated bin by bin in the class's own arithmetic
// (siftmatch.c:97-116).  Here:
//   pack    every descriptor x is split into two bf16 planes, hi = bf16(x), lo = bf16(x - hi)  (|x - hi - lo| <= 2^-16 |x|), stored in the
//           fragment order of v_mfma_f32_16x16x32_bf16, with its squared norm n (accumulated in double) as the two floats n(1+E), n(1-E);
//   phase 1 dt_j = n_j - 2 (qh.bh + qh.bl + ql.bh)  (three bf16 products per 16x16 block, f32 accumulation; the query's own norm is a
//           per-query constant and stays out of the scan).  |d_j - dt_j - nq| <= eps_j = E (nq + n_j): 3*2^-16 |q||b| from the planes and
//           the dropped ql.bl, <= 2*384*2^-24 |q||b| from the f32 accumulation of 384 products (a bound linear in the count, truncation
//           allowed), 2^-16 (nq + n_j) for the float class's own bin-by-bin rounding, 2^-21 for the norms; together < 2^-13.2 (nq + n_j),
//           E = 1.8e-4 > 2^-12.5 leaves a further 1.6x.  Each lane (= query, as in k_match_i8_q) keeps the two smallest UPPER bounds
//           dt_j + eps_j over its share of the database; merged over the workgroup that gives U2, an upper bound of the true second best;
//   phase 2 the same products again (the database is L2 resident): every column whose LOWER bound dt_j - eps_j is <= U2 is a candidate.
//           The true best, the true second best and everything tied with them have d_j <= U2, so they are all candidates;
//   tail    one wave per query re-evaluates its candidates (<= 64, one per lane) from the ORIGINAL descriptors in the reference's
//           accumulation order and arithmetic (bin 0..ND-1, subtract, multiply, add, contraction off) and merges them with merge3: the
//           outputs are bit-identical to k_match_exact's.  A query with more than 64 candidates is scanned in full by its wave.
// Data the bounds do not cover (NaN / Inf, |x| > 2^60, 0 < |x| < 2^-40: squares or planes leaving the f32 range) set a flag in the pack
// kernel and the host takes the exact kernels; descriptors that are integers in [0, 255] in every bin (Lowe-format descriptor files such as
// sift/data/box.sift read into doubles; NOT what matching_sift_based.m:104-118 passes -- those are siftdescriptor.c:125-141's unit-norm real
// values and take the ranked route above) set another and go to the int8 kernel, whose integer distances are exactly the reference's sums.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr double RK_E = 1.8e-4;
constexpr float RK_PAD_NORM = 3.0e38f;
constexpr int RK_WAVES = 16, RK_Q = 16, RK_CAP = 64;

// One wave re-evaluates one query's candidates in the reference's arithmetic (siftmatch.c:97-116: bin 0..ND-1, subtract, multiply, add,
// contraction off).  The accumulation is a chain over the bins, the loads are not: the wave fetches RK_CH candidate columns at a time into
// LDS with coalesced loads (all in flight together), then lane c walks column c and the query (LDS broadcast), eight bins' reads ahead of
// their use.  More than RK_CAP candidates (duplicated descriptors): the wave scans the whole database, 16 bins of 64 columns in flight.
template <typename T> struct RankTail { static constexpr int CH = 32 / sizeof(T), LDT = 128 + 16 / sizeof(T); };
template <typename T>
__device__ __forceinline__ void rank_tail_wave(int ND, int K2, int n, const int *cd /* LDS: the wave's candidate list */, const T *qrow /* LDS */, T *cb /* LDS [CH][LDT] */,
                                               const T *__restrict__ L2, int lane, T &eb, T &es, int &ek, int *__restrict__ stats)
{
#pragma clang fp contract(off)
    constexpr int RK_CH = RankTail<T>::CH, RK_LDT = RankTail<T>::LDT;
    eb = acc_max<T>(); es = acc_max<T>(); ek = -1;
    if (n <= RK_CAP) {
        for (int c0 = 0; c0 < n; c0 += RK_CH) {
            const int nc = n - c0 < RK_CH ? n - c0 : RK_CH;
            for (int i = lane; i < nc * ND; i += 64) {
                const int c = i / ND, b = i - c * ND;
                cb[c * RK_LDT + b] = L2[(size_t)cd[c0 + c] * ND + b];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < nc) {
                const T *bp = cb + lane * RK_LDT;
                T acc = 0;
                int bin = 0;
                for (; bin + 8 <= ND; bin += 8) {
                    T qv[8], bv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { qv[j] = qrow[bin + j]; bv[j] = bp[bin + j]; }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const T delta = qv[j] - bv[j];
                        const T sq = delta * delta;
                        acc = acc + sq;
                    }
                }
                for (; bin < ND; ++bin) {
                    const T delta = qrow[bin] - bp[bin];
                    const T sq = delta * delta;
                    acc = acc + sq;
                }
                merge3(eb, es, ek, acc, acc_max<T>(), cd[c0 + lane]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (int k0 = 0; k0 < K2; k0 += 64) {
            const int k2 = k0 + lane;
            const T *bp = L2 + (size_t)(k2 < K2 ? k2 : 0) * ND;
            T acc = 0;
            for (int b0 = 0; b0 < ND; b0 += 16) {
                T bv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) bv[j] = b0 + j < ND ? bp[b0 + j] : (T)0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (b0 + j < ND) {
                        const T delta = qrow[b0 + j] - bv[j];
                        const T sq = delta * delta;
                        acc = acc + sq;
                    }
                }
            }
            if (k2 < K2) push3(eb, es, ek, acc, k2);
        }
        if (lane == 0 && stats) atomicAdd(stats, 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T ob = __shfl_xor(eb, o, 64), os = __shfl_xor(es, o, 64);
        const int ok = __shfl_xor(ek, o, 64);
        merge3(eb, es, ek, ob, os, ok);
    }
}

// one thread per 16 bins of one descriptor (8 threads per descriptor: 128 bins, zero padded); K_pad descriptors
template <typename T>
__global__ __launch_bounds__(256) void k_rank_pack(int ND, int K, int Kp, const T *__restrict__ L, v4i *__restrict__ hi, v4i *__restrict__ lo,
                                                    float *__restrict__ nU, float *__restrict__ nL, float *__restrict__ nrm,
                                                    int8_t *__restrict__ i8, int *__restrict__ i8norm, int *__restrict__ flags)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int kcol = g >> 3, ch = g & 7;
    if (kcol >= Kp) return;
    double s = 0;
    int si = 0, fl = 0;
    int8_t v8[16];
    bf16x8 h[2], l[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int b = ch * 16 + j;
        const double x = (kcol < K && b < ND) ? (double)L[(size_t)kcol * ND + b] : 0.0;
        const double ax = fabs(x);
        if (!(ax <= 0x1p60) || (ax != 0.0 && ax < 0x1p-40)) fl |= 1;
        const bool isb = x >= 0.0 && x <= 255.0 && x == floor(x);
        if (!isb) fl |= 2;
        const int xi = (isb ? (int)x : 0) - 128;                   // the int8 route re-centres by -128 like the uint8 class (a - b unchanged)
        v8[j] = (int8_t)xi; si += xi * xi;
        const __bf16 hb = (__bf16)(float)x;
        const double r = x - (double)(float)hb;
        h[j >> 3][j & 7] = hb; l[j >> 3][j & 7] = (__bf16)(float)r;
        s += x * x;
    }
    if (kcol >= K) si = 0;
    // fragment order, 8-bin chunks c8 = 2 ch, 2 ch + 1: [block of 16 descriptors][k-step of 32 bins][lane = (descriptor & 15) + 16 (chunk & 3)]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c8 = 2 * ch + t;
        const size_t o = ((size_t)(kcol >> 4) * 4 + (c8 >> 2)) * 64 + (kcol & 15) + 16 * (c8 & 3);
        hi[o] = __builtin_bit_cast(v4i, h[t]); lo[o] = __builtin_bit_cast(v4i, l[t]);
    }
    // the int8 operands in k_match_i8_q's order (k_pack_i8_v, frag = 1)
    *reinterpret_cast<int4 *>(i8 + (((size_t)(kcol >> 4) * 2 + (ch >> 2)) * 64 + (kcol & 15) + 16 * (ch & 3)) * 16) = *reinterpret_cast<const int4 *>(v8);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); si += __shfl_xor(si, o, 64); }
    if (ch == 0) {
        const bool real = kcol < K;
        nU[kcol] = real ? (float)(s * (1.0 + RK_E)) : RK_PAD_NORM; nL[kcol] = real ? (float)(s * (1.0 - RK_E)) : RK_PAD_NORM;
        nrm[kcol] = real ? (float)s : 0.f;
        i8norm[kcol] = real ? si : I8_NONE_NORM;
    }
    if (fl) atomicOr(flags, fl);
}

template <typename T>
__global__ __launch_bounds__(64 * RK_WAVES) void k_match_rank(int ND, int K1, int K2, int K2p, const v4i *__restrict__ Qh, const v4i *__restrict__ Ql,
                                                                const v4i *__restrict__ Dh, const v4i *__restrict__ Dl, const float *__restrict__ nq,
                                                                const float *__restrict__ nU, const float *__restrict__ nL, const T *__restrict__ L1,
                                                                const T *__restrict__ L2, int k2_offset, double *__restrict__ obest,
                                                                double *__restrict__ osecond, int32_t *__restrict__ oarg, int *__restrict__ stats)
{
#pragma clang fp contract(off)
    __shared__ float mb[RK_WAVES * RK_Q], ms[RK_WAVES * RK_Q], thr_s[RK_Q];
    __shared__ int cnt[RK_Q], cand[RK_Q][RK_CAP];
    constexpr int RK_CH = 32 / sizeof(T), RK_LDT = 128 + 16 / sizeof(T);    // candidate columns per tail pass (1 KB each) and their LDS stride
    __shared__ T qs[RK_Q][128], cols[RK_WAVES][RK_CH][RK_LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, qi = lane & 15;
    for (int i = tid; i < RK_Q * ND; i += 64 * RK_WAVES) {                  // the workgroup's queries as the caller gave them, for the tail
        const int w = i / ND, b = i - w * ND, qq = blockIdx.x * RK_Q + w;
        qs[w][b] = qq < K1 ? L1[(size_t)qq * ND + b] : (T)0;
    }
    const int nblk = K2p / 16;
    struct Set { v4i h[4], l[4]; v4f n; };
    v4i fqh[4], fql[4];
    {
        const v4i *qh = Qh + (size_t)blockIdx.x * 256 + lane, *ql = Ql + (size_t)blockIdx.x * 256 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) { fqh[s] = qh[64 * s]; fql[s] = ql[64 * s]; }
    }
    auto load = [&](int blk, Set &S, const float *__restrict__ norms) {
        const v4i *dh = Dh + (size_t)blk * 256 + lane, *dl = Dl + (size_t)blk * 256 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) { S.h[s] = dh[64 * s]; S.l[s] = dl[64 * s]; }
        S.n = *reinterpret_cast<const v4f *>(norms + blk * 16 + 4 * g);
    };
    // database block = A operand (accumulator rows 4 (lane >> 4) + e), query block = B operand (accumulator column lane & 15)
    auto dots = [&](const Set &S) {
        v4f acc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.h[s]), __builtin_bit_cast(bf16x8, fqh[s]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.h[s]), __builtin_bit_cast(bf16x8, fql[s]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.l[s]), __builtin_bit_cast(bf16x8, fqh[s]), acc, 0, 0, 0);
        }
        return acc;
    };
    const int last = wave < nblk ? wave + (nblk - 1 - wave) / RK_WAVES * RK_WAVES : 0;
    auto at = [&](int b) { return b < last ? b : last; };
    Set A, B;
    // ---- phase 1: the two smallest upper bounds per lane
    float best = INFINITY, second = INFINITY;
    if (wave < nblk) {
        auto scan = [&](const Set &S) {
            const v4f acc = dots(S);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = fmaf(-2.f, acc[e], S.n[e]);
                second = __builtin_amdgcn_fmed3f(best, u, second);
                best = fminf(best, u);
            }
        };
        load(wave, A, nU);
        for (int blk = wave; blk < nblk; blk += 2 * RK_WAVES) {
            load(at(blk + RK_WAVES), B, nU);
            scan(A);
            if (blk + RK_WAVES >= nblk) break;
            load(at(blk + 2 * RK_WAVES), A, nU);
            scan(B);
        }
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
        second = fminf(fmaxf(best, ob), fminf(second, os));
        best = fminf(best, ob);
    }
    if (g == 0) { mb[wave * RK_Q + lane] = best; ms[wave * RK_Q + lane] = second; }
    __syncthreads();
    if (tid < RK_Q) {
        float b = mb[tid], s2 = ms[tid];
#pragma unroll
        for (int w = 1; w < RK_WAVES; ++w) {
            const float ob = mb[w * RK_Q + tid], os = ms[w * RK_Q + tid];
            s2 = fminf(fmaxf(b, ob), fminf(s2, os));
            b = fminf(b, ob);
        }
        const int q = blockIdx.x * RK_Q + tid;
        thr_s[tid] = s2 + (float)(2.0 * RK_E) * (q < K1 ? nq[q] : 0.f);       // lower bound <= upper bound of the second best, the query's norm moved across
        cnt[tid] = 0;
    }
    __syncthreads();
    // ---- phase 2: the candidates
    if (wave < nblk) {
        const float thr = thr_s[qi];
        auto emit = [&](int blk, const Set &S) {
            const v4f acc = dots(S);
            float lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) lw[e] = fmaf(-2.f, acc[e], S.n[e]);
            if (fminf(fminf(lw[0], lw[1]), fminf(lw[2], lw[3])) <= thr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = blk * 16 + 4 * g + e;
                    if (lw[e] <= thr && idx < K2) {
                        const int p = atomicAdd(&cnt[qi], 1);
                        if (p < RK_CAP) cand[qi][p] = idx;
                    }
                }
            }
        };
        load(wave, A, nL);
        for (int blk = wave; blk < nblk; blk += 2 * RK_WAVES) {
            load(at(blk + RK_WAVES), B, nL);
            emit(blk, A);
            if (blk + RK_WAVES >= nblk) break;
            load(at(blk + 2 * RK_WAVES), A, nL);
            emit(blk + RK_WAVES, B);
        }
    }
    __syncthreads();
    // ---- tail: wave w re-evaluates query w's candidates (rank_tail_wave)
    const int q = blockIdx.x * RK_Q + wave;
    if (q >= K1) return;
    const int n = cnt[wave];
    T eb, es; int ek;
    rank_tail_wave<T>(ND, K2, n, cand[wave], qs[wave], &cols[wave][0][0], L2, lane, eb, es, ek, stats);
    if (lane == 0) { obest[q] = (double)eb; osecond[q] = (double)es; oarg[q] = ek < 0 ? -1 : ek + k2_offset; if (stats) atomicAdd(stats + 1, n < RK_CAP ? n : RK_CAP); }
}

// ---- the same two phases tiled BOTH ways (the form used when there is enough work for it) -----------------------------------------
// k_match_rank streams the whole database (K2 x 512 B of planes) into every 16-query workgroup, twice: at 4096 x 4096 that is 2 x 512 MB
// through the L2 -> CU paths and sets its time (phase 1 23 us, phase 2 20 us).  Here a workgroup (8 waves) owns RT_NQB x 16 = 64 queries and
// ONE slice of the database; every wave keeps the fragments of all 64 queries in registers (4 query blocks x 2 planes x 4 k-steps = 128 VGPRs)
// and walks its share of the slice's blocks, so one 8 KB database fragment set feeds 4 x 12 = 48 MFMAs instead of 12 and the launch is bound by
// the matrix pipe, not by operand delivery.  The price is that a query's database is spread over several workgroups:
//   k_rank_bounds   phase 1 per (query group, slice): the two smallest upper bounds -> pb / ps [slice][query]; zeroes the candidate counters
//   k_rank_emit     merges the slices' bounds into U2, phase 2 on its slice, candidates appended to per-query lists in global memory
//   k_rank_tail     one wave per query: exact re-evaluation of the listed candidates (the tail of k_match_rank)
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int RT_NQB = 4, RT_WAVES = 8, RT_Q = 16 * RT_NQB;

struct RtSet { v4i h[4], l[4]; v4f n; };

// phase: 0 = bounds (norms = nU), 1 = emit (norms = nL)
template <int PHASE>
__global__ __launch_bounds__(64 * RT_WAVES) void k_rank_tiled(int K1, int K2, int K2p, int nsl, const v4i *__restrict__ Qh, const v4i *__restrict__ Ql,
                                                                const v4i *__restrict__ Dh, const v4i *__restrict__ Dl, const float *__restrict__ nq,
                                                                const float *__restrict__ norms, float *__restrict__ pb, float *__restrict__ ps, int K1p,
                                                                int *__restrict__ gcnt, int *__restrict__ gcand)
{
    __shared__ float mb[RT_WAVES][RT_Q], ms[RT_WAVES][RT_Q];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, qi = lane & 15;
    const int qg = blockIdx.x, sl = blockIdx.y;
    const int nblk = K2p / 16;
    const int b0 = (int)((long long)nblk * sl / nsl), b1 = (int)((long long)nblk * (sl + 1) / nsl);      // this slice's 16-column blocks
    auto load = [&](int blk, RtSet &S) {
        const v4i *dh = Dh + (size_t)blk * 256 + lane, *dl = Dl + (size_t)blk * 256 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) { S.h[s] = dh[64 * s]; S.l[s] = dl[64 * s]; }
        S.n = *reinterpret_cast<const v4f *>(norms + blk * 16 + 4 * g);
    };
    RtSet A, B;
    const int first = b0 + wave;
    if (first < b1) load(first, A);                   // in flight behind the staging of the queries
    // the 64 queries' fragments (32 KB) cross L2 -> CU once per workgroup and reach the eight waves' registers through LDS
    __shared__ v4i qst[2][RT_NQB * 256];
    for (int i = tid; i < RT_NQB * 256; i += 64 * RT_WAVES) {
        qst[0][i] = Qh[(size_t)qg * RT_NQB * 256 + i]; qst[1][i] = Ql[(size_t)qg * RT_NQB * 256 + i];
    }
    __syncthreads();
    v4i fqh[RT_NQB][4], fql[RT_NQB][4];
#pragma unroll
    for (int t = 0; t < RT_NQB; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) { fqh[t][s] = qst[0][t * 256 + 64 * s + lane]; fql[t][s] = qst[1][t * 256 + 64 * s + lane]; }
    float thr[RT_NQB];
    if (PHASE == 1) {
        // U2 of a query = second smallest upper bound over all slices; the query's own norm moved across (see k_match_rank)
#pragma unroll
        for (int t = 0; t < RT_NQB; ++t) {
            const int q = qg * RT_Q + 16 * t + qi;
            float b = INFINITY, s2 = INFINITY;
            for (int x = 0; x < nsl; ++x) {
                const float ob = pb[(size_t)x * K1p + q], os = ps[(size_t)x * K1p + q];
                s2 = fminf(fmaxf(b, ob), fminf(s2, os));
                b = fminf(b, ob);
            }
            thr[t] = s2 + (float)(2.0 * RK_E) * (q < K1 ? nq[q] : 0.f);
        }
    } else if (sl == 0 && tid < RT_Q) gcnt[qg * RT_Q + tid] = 0;
    float best[RT_NQB], second[RT_NQB];
#pragma unroll
    for (int t = 0; t < RT_NQB; ++t) best[t] = second[t] = INFINITY;
    auto work = [&](int blk, const RtSet &S) {
        // four independent accumulator chains (one per query block), interleaved: a dependent MFMA never issues right behind its producer
        v4f acc[RT_NQB];
#pragma unroll
        for (int t = 0; t < RT_NQB; ++t) acc[t] = v4f{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < RT_NQB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.h[s]), __builtin_bit_cast(bf16x8, fqh[t][s]), acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT_NQB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.h[s]), __builtin_bit_cast(bf16x8, fql[t][s]), acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT_NQB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.l[s]), __builtin_bit_cast(bf16x8, fqh[t][s]), acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < RT_NQB; ++t) {
            if (PHASE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = fmaf(-2.f, acc[t][e], S.n[e]);
                    second[t] = __builtin_amdgcn_fmed3f(best[t], u, second[t]);
                    best[t] = fminf(best[t], u);
                }
            } else {
                float lw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) lw[e] = fmaf(-2.f, acc[t][e], S.n[e]);
                if (fminf(fminf(lw[0], lw[1]), fminf(lw[2], lw[3])) <= thr[t]) {
                    const int q = qg * RT_Q + 16 * t + qi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = blk * 16 + 4 * g + e;
                        if (lw[e] <= thr[t] && idx < K2 && q < K1) {
                            const int p = atomicAdd(&gcnt[q], 1);
                            if (p < RK_CAP) gcand[(size_t)q * RK_CAP + p] = idx;
                        }
                    }
                }
            }
        }
    };
    if (first < b1) {
        const int last = first + (b1 - 1 - first) / RT_WAVES * RT_WAVES;
        auto at = [&](int b) { return b < last ? b : last; };
        for (int blk = first; blk < b1; blk += 2 * RT_WAVES) {
            load(at(blk + RT_WAVES), B);
            work(blk, A);
            if (blk + RT_WAVES >= b1) break;
            load(at(blk + 2 * RT_WAVES), A);
            work(blk + RT_WAVES, B);
        }
    }
    if (PHASE == 0) {
#pragma unroll
        for (int t = 0; t < RT_NQB; ++t) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ob = __shfl_xor(best[t], o, 64), os = __shfl_xor(second[t], o, 64);
                second[t] = fminf(fmaxf(best[t], ob), fminf(second[t], os));
                best[t] = fminf(best[t], ob);
            }
            if (g == 0) { mb[wave][16 * t + lane] = best[t]; ms[wave][16 * t + lane] = second[t]; }
        }
        __syncthreads();
        if (tid < RT_Q) {
            float b = mb[0][tid], s2 = ms[0][tid];
#pragma unroll
            for (int w = 1; w < RT_WAVES; ++w) {
                const float ob = mb[w][tid], os = ms[w][tid];
                s2 = fminf(fmaxf(b, ob), fminf(s2, os));
                b = fminf(b, ob);
            }
            pb[(size_t)sl * K1p + qg * RT_Q + tid] = b; ps[(size_t)sl * K1p + qg * RT_Q + tid] = s2;
        }
    }
}

// ---- round 6: the same two phases with 128 queries per workgroup and the database slice staged through LDS --------------------------------------------
// k_rank_tiled moves 2 MB of database planes into every one of its K1p / 64 query groups and 2 MB of query planes into every slice: 136 MB through the
// L2 -> CU paths per phase at 4096 x 4096 (64 groups x 4 slices), which is what sets its time.  Here a workgroup owns TWO sets of four query blocks -- waves
// 0-3 hold the fragments of queries 0..63 of the group, waves 4-7 those of 64..127 -- and every block of the slice is fetched ONCE (LDS-DMA, three-slot ring
// of four blocks) and read from LDS by one wave of each half: 32 groups x 8 slices, 80 MB per phase.  Same MFMA chains per (query block, database block)
// as k_rank_tiled: the same bounds, the same candidate sets.
// *(measured, round 6)* 52.2 us per match against k_rank_tiled's 43.5 (unit-norm doubles), 47.3 against 40.0 (floats), with a three- or a four-slot ring alike:
// the phases are NOT bound by operand delivery at this size -- both forms sit at three times their MFMA time, and the lock-step of eight waves per step
// costs more than the halved traffic saves.  Kept as PRE3_MATCH_RANK_FORM=2 (bit-identical: tests/test_gpu_match_rank.py), not the default.
constexpr int R2_Q = 128, R2_STEP = 4;                // queries per workgroup; database blocks per step (one per wave of a half)
template <int PHASE>
__global__ __launch_bounds__(64 * RT_WAVES) void k_rank_tiled2(int K1, int K2, int K2p, int nsl, const v4i *__restrict__ Qh, const v4i *__restrict__ Ql,
                                                                 const v4i *__restrict__ Dh, const v4i *__restrict__ Dl, const float *__restrict__ nq,
                                                                 const float *__restrict__ norms, float *__restrict__ pb, float *__restrict__ ps, int K1p,
                                                                 int *__restrict__ gcnt, int *__restrict__ gcand)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char r2_smem[];
    v4i *ring = reinterpret_cast<v4i *>(r2_smem);                            // [4 slots][plane 2][R2_STEP blocks][256] v4i = 4 x 32 KB
    __shared__ float mb[RT_WAVES][RT_Q], ms[RT_WAVES][RT_Q];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, qi = lane & 15;
    const int half = wave >> 2, wq = wave & 3;
    const int qg = blockIdx.x, sl = blockIdx.y;
    const int nblk = K2p / 16;
    const int b0 = (int)((long long)nblk * sl / nsl), b1 = (int)((long long)nblk * (sl + 1) / nsl);
    const int nstep = (b1 - b0 + R2_STEP - 1) / R2_STEP;
    constexpr int SLOT = 2 * R2_STEP * 256;                                  // v4i per slot
    // this thread's share of a step's 2048 granules: granule i = tid + 512 u (u < 4): plane i >> 10, the rest linear in the step's blocks
    auto issue = [&](const int step, const int slot) {
        const int blk0 = b0 + step * R2_STEP;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 512 * u, pl = i >> 10, w = i & 1023;
            int blk = blk0 + (w >> 8);
            blk = blk < b1 ? blk : b1 - 1;                                   // (a short last step re-reads the slice's last block: never used)
            const v4i *src = (pl ? Dl : Dh) + (size_t)blk * 256 + (w & 255);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(ring + slot * SLOT + (i & ~63)), 16, 0, 0);
        }
    };
    // the 128 queries' fragments (64 KB) go through slots 2 and 3 into registers; slots 0 and 1 take the first two steps meanwhile
    if (nstep > 0) issue(0, 0);
    if (nstep > 1) issue(1, 1);
    {
        v4i *qst = ring + 2 * SLOT;
        for (int i = tid; i < 2 * RT_NQB * 256; i += 64 * RT_WAVES) {
            qst[i] = Qh[(size_t)qg * 2 * RT_NQB * 256 + i]; qst[2 * RT_NQB * 256 + i] = Ql[(size_t)qg * 2 * RT_NQB * 256 + i];
        }
    }
    __syncthreads();
    v4i fqh[RT_NQB][4], fql[RT_NQB][4];
    {
        const v4i *qst = ring + 2 * SLOT;
#pragma unroll
        for (int t = 0; t < RT_NQB; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                fqh[t][s2] = qst[(half * RT_NQB + t) * 256 + 64 * s2 + lane];
                fql[t][s2] = qst[2 * RT_NQB * 256 + (half * RT_NQB + t) * 256 + 64 * s2 + lane];
            }
    }
    __syncthreads();                                                         // (slots 2 and 3 are free)
    if (nstep > 2) issue(2, 2);
    float thr[RT_NQB];
    if (PHASE == 1) {
#pragma unroll
        for (int t = 0; t < RT_NQB; ++t) {
            const int q = qg * R2_Q + half * RT_Q + 16 * t + qi;
            float b = INFINITY, s2 = INFINITY;
            for (int x = 0; x < nsl; ++x) {
                const float ob = pb[(size_t)x * K1p + q], os = ps[(size_t)x * K1p + q];
                s2 = fminf(fmaxf(b, ob), fminf(s2, os));
                b = fminf(b, ob);
            }
            thr[t] = s2 + (float)(2.0 * RK_E) * (q < K1 ? nq[q] : 0.f);
        }
    } else if (sl == 0 && tid < R2_Q) gcnt[qg * R2_Q + tid] = 0;
    float best[RT_NQB], second[RT_NQB];
#pragma unroll
    for (int t = 0; t < RT_NQB; ++t) best[t] = second[t] = INFINITY;
    for (int step = 0; step < nstep; ++step) {
        const int slot = step & 3;
        // this step's granules have landed (the next two steps' four per thread each may still fly), everybody's: the slot of step - 1 is free for step + 3
        if (step + 2 < nstep) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (step + 1 < nstep) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);                                  // (raw barrier: __syncthreads() would wait for the next steps' granules too)
        __builtin_amdgcn_s_barrier();
        if (step + 3 < nstep) issue(step + 3, (step + 3) & 3);
        const int blk = b0 + step * R2_STEP + wq;
        if (blk < b1) {
            RtSet S;
            const v4i *sh = ring + slot * SLOT + wq * 256 + lane, *sl2 = sh + R2_STEP * 256;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) { S.h[s2] = sh[64 * s2]; S.l[s2] = sl2[64 * s2]; }
            S.n = *reinterpret_cast<const v4f *>(norms + blk * 16 + 4 * g);
            v4f acc[RT_NQB];
#pragma unroll
            for (int t = 0; t < RT_NQB; ++t) acc[t] = v4f{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
                for (int t = 0; t < RT_NQB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.h[s2]), __builtin_bit_cast(bf16x8, fqh[t][s2]), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < RT_NQB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.h[s2]), __builtin_bit_cast(bf16x8, fql[t][s2]), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < RT_NQB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, S.l[s2]), __builtin_bit_cast(bf16x8, fqh[t][s2]), acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < RT_NQB; ++t) {
                if (PHASE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = fmaf(-2.f, acc[t][e], S.n[e]);
                        second[t] = __builtin_amdgcn_fmed3f(best[t], u, second[t]);
                        best[t] = fminf(best[t], u);
                    }
                } else {
                    float lw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) lw[e] = fmaf(-2.f, acc[t][e], S.n[e]);
                    if (fminf(fminf(lw[0], lw[1]), fminf(lw[2], lw[3])) <= thr[t]) {
                        const int q = qg * R2_Q + half * RT_Q + 16 * t + qi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int idx = blk * 16 + 4 * g + e;
                            if (lw[e] <= thr[t] && idx < K2 && q < K1) {
                                const int p2 = atomicAdd(&gcnt[q], 1);
                                if (p2 < RK_CAP) gcand[(size_t)q * RK_CAP + p2] = idx;
                            }
                        }
                    }
                }
            }
        }
    }
    if (PHASE == 0) {
#pragma unroll
        for (int t = 0; t < RT_NQB; ++t) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ob = __shfl_xor(best[t], o, 64), os = __shfl_xor(second[t], o, 64);
                second[t] = fminf(fmaxf(best[t], ob), fminf(second[t], os));
                best[t] = fminf(best[t], ob);
            }
            if (g == 0) { mb[wave][16 * t + lane] = best[t]; ms[wave][16 * t + lane] = second[t]; }
        }
        __syncthreads();
        if (tid < R2_Q) {
            const int hf = tid >> 6, qq = tid & 63;
            float b = mb[4 * hf][qq], s2 = ms[4 * hf][qq];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float ob = mb[4 * hf + w][qq], os = ms[4 * hf + w][qq];
                s2 = fminf(fmaxf(b, ob), fminf(s2, os));
                b = fminf(b, ob);
            }
            pb[(size_t)sl * K1p + qg * R2_Q + tid] = b; ps[(size_t)sl * K1p + qg * R2_Q + tid] = s2;
        }
    }
}

// one wave per query: the candidates of k_rank_tiled<1>, re-evaluated as the tail of k_match_rank does
template <typename T>
__global__ __launch_bounds__(256) void k_rank_tail(int ND, int K1, int K2, const T *__restrict__ L1, const T *__restrict__ L2, const int *__restrict__ gcnt,
                                                    const int *__restrict__ gcand, int k2_offset, double *__restrict__ obest, double *__restrict__ osecond,
                                                    int32_t *__restrict__ oarg, int *__restrict__ stats)
{
#pragma clang fp contract(off)
    constexpr int RK_CH = 32 / sizeof(T), RK_LDT = 128 + 16 / sizeof(T);
    __shared__ T qs[4][128], cols[4][RK_CH][RK_LDT];
    __shared__ int cd[4][RK_CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= K1) return;
    const int n = gcnt[q];
    static_assert(RK_CAP == 64, "one candidate slot per lane");
    for (int b = lane; b < ND; b += 64) qs[wave][b] = L1[(size_t)q * ND + b];
    cd[wave][lane] = gcand[(size_t)q * RK_CAP + lane];          // (the whole list, without waiting for the count: entries >= n are never read)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T eb, es; int ek;
    rank_tail_wave<T>(ND, K2, n, cd[wave], qs[wave], &cols[wave][0][0], L2, lane, eb, es, ek, stats);
    if (lane == 0) { obest[q] = (double)eb; osecond[q] = (double)es; oarg[q] = ek < 0 ? -1 : ek + k2_offset; if (stats) atomicAdd(stats + 1, n < RK_CAP ? n : RK_CAP); }
}

// ------------------------------------------------------------------------------------------------
// kNearestNeighbors.m:29-39: block per query; distances in the reference's accumulation order, then k
// selection rounds (stable: lowest index first on ties).  data N x D, query M x D column-major.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_knn(int D, int N, int M, const double *__restrict__ data, const double *__restrict__ query, int k,
                                             double *__restrict__ scratch, double *__restrict__ ids, double *__restrict__ dist)
{
#pragma clang fp contract(off)
    const int qi = blockIdx.x, tid = threadIdx.x;
    double *dsq = scratch + (size_t)qi * N;
    for (int j = tid; j < N; j += blockDim.x) {
        double s = 0;
        for (int d = 0; d < D; ++d) {
            double t = query[(size_t)d * M + qi] - data[(size_t)d * N + j];
            double sq = t * t;
            s = s + sq;
        }
        dsq[j] = s;
    }
    __syncthreads();
    __shared__ double sv[4];
    __shared__ int si[4];
    __shared__ double lastv;
    __shared__ int lasti;
    if (tid == 0) { lastv = -INFINITY; lasti = -1; }
    __syncthreads();
    for (int c = 0; c < k; ++c) {
        double bv = INFINITY; int bi = 0x7fffffff;
        double lv = lastv; int li = lasti;
        for (int j = tid; j < N; j += blockDim.x) {
            double v = dsq[j];
            // strictly after (lv, li) in (value, index) order
            bool after = (v > lv) || (v == lv && j > li);
            if (after && (v < bv || (v == bv && j < bi))) { bv = v; bi = j; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            double ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) if (sv[w] < sv[0] || (sv[w] == sv[0] && si[w] < si[0])) { sv[0] = sv[w]; si[0] = si[w]; }
            lastv = sv[0]; lasti = si[0];
            ids[(size_t)c * M + qi] = (double)(si[0] + 1);
            dist[(size_t)c * M + qi] = sqrt(sv[0]);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
// Scratch for the stateless matcher / kNN / VO entry points.  hipMalloc + hipFree cost ~0.2 ms a pair, eight pairs per call dwarf a
// 45 us kernel, and the reference calls siftmatch once per frame: buffers are kept in a small per-device pool
// (grow-only, at most 32 entries, shared by all threads under a mutex, released by pre3_release_scratch or at exit).
struct ScratchPool {
    struct Ent { void *p; size_t cap; int dev; bool used; };
    std::vector<Ent> ents;
    ~ScratchPool() { for (Ent &e : ents) if (e.p) (void)hipFree(e.p); }
    void release() { for (Ent &e : ents) if (e.p && !e.used) { (void)hipFree(e.p); e.p = nullptr; e.cap = 0; } }
};
static ScratchPool g_scratch;
static std::mutex g_scratch_mu;

// pooled allocation (also used by pre3_vo.hip): *slot >= 0 -> return it with scratch_release(); *slot < 0 -> plain hipMalloc
int scratch_acquire(size_t bytes, void **p_out, int *slot_out)
{
    if (bytes == 0) bytes = 16;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    int best = -1, freeslot = -1;
    for (int i = 0; i < (int)g_scratch.ents.size(); ++i) {
        ScratchPool::Ent &e = g_scratch.ents[i];
        if (e.used) continue;
        if (!e.p) { freeslot = i; continue; }
        if (e.dev == dev && e.cap >= bytes && (best < 0 || e.cap < g_scratch.ents[best].cap)) best = i;
    }
    if (best >= 0) { *slot_out = best; g_scratch.ents[best].used = true; *p_out = g_scratch.ents[best].p; return PRE3_OK; }
    const size_t cap = bytes + bytes / 4;                         // a little headroom: frame-to-frame sizes vary
    void *p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) { *p_out = nullptr; *slot_out = -1; set_error("hipMalloc of %zu bytes failed", cap); return PRE3_E_NOMEM; }
    if (freeslot < 0 && g_scratch.ents.size() < 32) { g_scratch.ents.push_back({ nullptr, 0, 0, false }); freeslot = (int)g_scratch.ents.size() - 1; }
    if (freeslot < 0) {                                              // pool full: recycle the smallest idle entry
        for (int i = 0; i < (int)g_scratch.ents.size(); ++i)
            if (!g_scratch.ents[i].used && (freeslot < 0 || g_scratch.ents[i].cap < g_scratch.ents[freeslot].cap)) freeslot = i;
        if (freeslot >= 0 && g_scratch.ents[freeslot].p) (void)hipFree(g_scratch.ents[freeslot].p);
    }
    if (freeslot >= 0) g_scratch.ents[freeslot] = { p, cap, dev, true };
    *p_out = p; *slot_out = freeslot;                                // (slot < 0: not pooled, the caller frees it)
    return PRE3_OK;
}

void scratch_release(int slot, void *p)
{
    if (slot >= 0) { std::lock_guard<std::mutex> lk(g_scratch_mu); g_scratch.ents[slot].used = false; }
    else if (p) (void)hipFree(p);
}

struct DevBuf {
    void *p = nullptr;
    int slot = -1;
    bool borrowed = false;                                   // p points into another DevBuf's allocation
    ~DevBuf() { if (!borrowed) scratch_release(slot, p); }
    int alloc(size_t bytes) { return scratch_acquire(bytes, &p, &slot); }
    void borrow(void *q) { if (!borrowed) scratch_release(slot, p); p = q; slot = -1; borrowed = true; }
};

void release_scratch() { std::lock_guard<std::mutex> lk(g_scratch_mu); g_scratch.release(); }

template <typename T, typename ACC>
static int partial_exact(int ND, int K1, const T *L1, int K2, const T *L2, int k2_offset, double *best, double *second, int32_t *arg)
{
    DevBuf d1, d2, db, ds, da;
    PRE3_TRY(d1.alloc(sizeof(T) * (size_t)ND * K1));
    PRE3_TRY(d2.alloc(sizeof(T) * (size_t)ND * K2));
    PRE3_TRY(db.alloc(sizeof(double) * K1)); PRE3_TRY(ds.alloc(sizeof(double) * K1)); PRE3_TRY(da.alloc(sizeof(int32_t) * K1));
    PRE3_HIP(hipMemcpy(d1.p, L1, sizeof(T) * (size_t)ND * K1, hipMemcpyHostToDevice));
    if (K2) PRE3_HIP(hipMemcpy(d2.p, L2, sizeof(T) * (size_t)ND * K2, hipMemcpyHostToDevice));
    if ((size_t)K1 * K2 >= (size_t)64 * 64 * 16) {
        const int ntn = ceil_div(K2, 64);
        DevBuf pb, ps, pa;
        PRE3_TRY(pb.alloc(sizeof(T) * (size_t)K1 * ntn)); PRE3_TRY(ps.alloc(sizeof(T) * (size_t)K1 * ntn)); PRE3_TRY(pa.alloc(sizeof(int32_t) * (size_t)K1 * ntn));
        hipLaunchKernelGGL((k_match_exact_tiled<T>), dim3(ntn, ceil_div(K1, 64)), dim3(256), 0, 0, ND, K1, K2, (const T *)d1.p, (const T *)d2.p,
                           (T *)pb.p, (T *)ps.p, (int32_t *)pa.p, ntn);
        hipLaunchKernelGGL((k_match_reduce_f<T>), dim3(ceil_div(16 * K1, 256)), dim3(256), 0, 0, K1, ntn, (const T *)pb.p, (const T *)ps.p, (const int32_t *)pa.p,
                           k2_offset, (double *)db.p, (double *)ds.p, (int32_t *)da.p);
        PRE3_HIP(hipGetLastError());
        PRE3_HIP(hipDeviceSynchronize());          // the partial buffers go out of scope below
    } else {
        hipLaunchKernelGGL((k_match_exact<T, ACC>), dim3(K1), dim3(256), sizeof(ACC) * ND, 0, ND, K2, (const T *)d1.p, (const T *)d2.p, k2_offset,
                           (double *)db.p, (double *)ds.p, (int32_t *)da.p);
    }
    PRE3_HIP(hipGetLastError());
    PRE3_HIP(hipMemcpy(best, db.p, sizeof(double) * K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(second, ds.p, sizeof(double) * K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(arg, da.p, sizeof(int32_t) * K1, hipMemcpyDeviceToHost));
    return PRE3_OK;
}

// device-resident int8 MFMA matcher state, reusable across calls (bench: inputs resident in HBM)
struct I8Match {
    int ND = 0, NDp = 0, K1 = 0, K2 = 0, K1p = 0, K2p = 0, ntn = 0;
    bool frag = false;                            // operands packed in MFMA fragment order for k_match_i8_q (128-bin descriptors)
    DevBuf A, B, na, nb, pb, ps, pa, ob, os, oa;
};

template <typename T>
static int i8_prepare(I8Match &m, int ND, int K1, const T *L1, int K2, const T *L2, int center)
{
    m.ND = ND; m.NDp = round_up(ND, 32); m.K1 = K1; m.K2 = K2; m.K1p = round_up(K1, 128); m.K2p = round_up(K2 ? K2 : 1, 128);
    m.ntn = m.K2p / 128;
    DevBuf r1, r2;
    PRE3_TRY(r1.alloc((size_t)ND * K1)); PRE3_TRY(r2.alloc((size_t)ND * (K2 ? K2 : 1)));
    PRE3_HIP(hipMemcpy(r1.p, L1, (size_t)ND * K1, hipMemcpyHostToDevice));
    if (K2) PRE3_HIP(hipMemcpy(r2.p, L2, (size_t)ND * K2, hipMemcpyHostToDevice));
    PRE3_TRY(m.A.alloc((size_t)m.K1p * m.NDp)); PRE3_TRY(m.B.alloc((size_t)m.K2p * m.NDp));
    PRE3_TRY(m.na.alloc(sizeof(int) * m.K1p)); PRE3_TRY(m.nb.alloc(sizeof(int) * m.K2p));
    size_t np = (size_t)m.K1p * m.ntn * 4;            // four partials (32-column chunks) per 128-column tile (row-scan form)
    PRE3_TRY(m.pb.alloc(sizeof(int) * np)); PRE3_TRY(m.ps.alloc(sizeof(int) * np)); PRE3_TRY(m.pa.alloc(sizeof(int) * np));
    PRE3_TRY(m.ob.alloc(sizeof(double) * K1)); PRE3_TRY(m.os.alloc(sizeof(double) * K1)); PRE3_TRY(m.oa.alloc(sizeof(int32_t) * K1));
    static const int form = getenv("PRE3_MATCH_I8_FORM") ? atoi(getenv("PRE3_MATCH_I8_FORM")) : 1;     // 0: row-scan form (k_match_i8_mfma + reduce), A/B
    m.frag = form == 1 && m.NDp == 128;
    launch_pack_i8<T>(ND, m.NDp, K1, m.K1p, (const T *)r1.p, center, (int8_t *)m.A.p, (int *)m.na.p, m.frag);
    launch_pack_i8<T>(ND, m.NDp, K2, m.K2p, (const T *)r2.p, center, (int8_t *)m.B.p, (int *)m.nb.p, m.frag);
    PRE3_HIP(hipGetLastError());
    PRE3_HIP(hipDeviceSynchronize());
    return PRE3_OK;
}

static int i8_run(I8Match &m, int k2_offset, hipStream_t st)
{
    if (m.frag) {
        hipLaunchKernelGGL(k_match_i8_q, dim3(m.K1p / I8Q_Q), dim3(64 * I8Q_WAVES), 0, st, m.K1, m.K2p, (const int8_t *)m.A.p, (const int8_t *)m.B.p,
                           (const int *)m.na.p, (const int *)m.nb.p, k2_offset, (double *)m.ob.p, (double *)m.os.p, (int32_t *)m.oa.p);
        PRE3_HIP(hipGetLastError());
        return PRE3_OK;
    }
    dim3 g(m.ntn, m.K1p / 128), b(256);
    hipLaunchKernelGGL(k_match_i8_mfma, g, b, 0, st, m.NDp, m.K1, m.K2, (const int8_t *)m.A.p, (const int8_t *)m.B.p, (const int *)m.na.p,
                       (const int *)m.nb.p, m.ntn, (int *)m.pb.p, (int *)m.ps.p, (int *)m.pa.p);
    hipLaunchKernelGGL(k_match_reduce_i32, dim3(ceil_div(16 * m.K1, 256)), dim3(256), 0, st, m.K1, m.K1p, m.ntn * 4, (const int *)m.pb.p, (const int *)m.ps.p,
                       (const int *)m.pa.p, k2_offset, (double *)m.ob.p, (double *)m.os.p, (int32_t *)m.oa.p);
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

template <typename T>
static int partial_i8(int ND, int K1, const T *L1, int K2, const T *L2, int center, int k2_offset, double *best, double *second, int32_t *arg)
{
    I8Match m;
    PRE3_TRY(i8_prepare(m, ND, K1, L1, K2, L2, center));
    PRE3_TRY(i8_run(m, k2_offset, 0));
    PRE3_HIP(hipMemcpy(best, m.ob.p, sizeof(double) * K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(second, m.os.p, sizeof(double) * K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(arg, m.oa.p, sizeof(int32_t) * K1, hipMemcpyDeviceToHost));
    return PRE3_OK;
}

// device-resident state of the float-class matcher (k_rank_pack / k_match_rank / the int8 route), reusable across calls
struct RankMatch {
    int cls = 0, ND = 0, K1 = 0, K2 = 0, K1p = 0, K2p = 0, route = 0;     // route 0: exact kernels (data outside the bounds), 1: int8 MFMA, 2: bf16 rank + re-evaluation
    DevBuf L1, L2, qh, ql, dh, dl, nq, nqU, nqL, nU, nL, nd, fl, pb, ps, gcnt, gcand;
    int nsl = 1;                                                          // database slices of the tiled form (k_rank_tiled)
    int nsl2 = 0;                                                         // ... of the LDS-staged form with 128 queries per workgroup (k_rank_tiled2); 0: that form does not apply
    I8Match m;                                                            // int8 operands + the outputs (ob, os, oa) of every route
};
static int float_form()
{
    const char *e = getenv("PRE3_MATCH_FLOAT_FORM");           // 0: exact VALU kernels only, 1: auto, 2: never the int8 route (A/B, tests); read per call
    return e ? atoi(e) : 1;
}
template <typename T>
static bool rank_applies(int ND, int K1, int K2) { return float_form() != 0 && ND <= 128 && (size_t)K1 * K2 >= (size_t)64 * 64 * 16; }

template <typename T>
static int rank_prepare(RankMatch &r, int ND, int K1, const T *L1, int K2, const T *L2)
{
    r.ND = ND; r.K1 = K1; r.K2 = K2; r.K1p = round_up(K1, 128); r.K2p = round_up(K2, 128);
    I8Match &m = r.m;
    m.ND = ND; m.NDp = 128; m.K1 = K1; m.K2 = K2; m.K1p = r.K1p; m.K2p = r.K2p; m.ntn = m.K2p / 128; m.frag = true;
    PRE3_TRY(r.L1.alloc(sizeof(T) * (size_t)ND * K1)); PRE3_TRY(r.L2.alloc(sizeof(T) * (size_t)ND * K2));
    PRE3_HIP(hipMemcpy(r.L1.p, L1, sizeof(T) * (size_t)ND * K1, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(r.L2.p, L2, sizeof(T) * (size_t)ND * K2, hipMemcpyHostToDevice));
    PRE3_TRY(r.qh.alloc((size_t)r.K1p * 256)); PRE3_TRY(r.ql.alloc((size_t)r.K1p * 256)); PRE3_TRY(r.dh.alloc((size_t)r.K2p * 256)); PRE3_TRY(r.dl.alloc((size_t)r.K2p * 256));
    PRE3_TRY(r.nq.alloc(sizeof(float) * r.K1p)); PRE3_TRY(r.nqU.alloc(sizeof(float) * r.K1p)); PRE3_TRY(r.nqL.alloc(sizeof(float) * r.K1p));
    PRE3_TRY(r.nU.alloc(sizeof(float) * r.K2p)); PRE3_TRY(r.nL.alloc(sizeof(float) * r.K2p)); PRE3_TRY(r.nd.alloc(sizeof(float) * r.K2p));
    PRE3_TRY(r.fl.alloc(sizeof(int) * 4));
    // tiled form: K1p / 64 query groups x nsl database slices ~ one workgroup per CU (and at least 8 blocks of 16 columns per wave-round)
    {
        static int ncu_of[64];                            // CUs per device, asked once (hipGetDeviceProperties is slow)
        int ncu = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (ncu_of[dev] == 0) { int v = 0; ncu_of[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256; }
            ncu = ncu_of[dev];
        }
        const int groups = r.K1p / RT_Q, nblk = r.K2p / 16;
        r.nsl = std::max(1, std::min(std::min(ceil_div(ncu, groups), nblk / RT_WAVES), 64));
        // (k_rank_tiled2: K1p / 128 groups, at least four steps of four blocks per slice)
        r.nsl2 = (r.K1p % R2_Q == 0 && nblk >= 4 * R2_STEP) ? std::max(1, std::min(std::min(ceil_div(ncu, r.K1p / R2_Q), nblk / (4 * R2_STEP)), 64)) : 0;
    }
    PRE3_TRY(r.pb.alloc(sizeof(float) * (size_t)std::max(r.nsl, r.nsl2) * r.K1p)); PRE3_TRY(r.ps.alloc(sizeof(float) * (size_t)std::max(r.nsl, r.nsl2) * r.K1p));
    PRE3_TRY(r.gcnt.alloc(sizeof(int) * r.K1p)); PRE3_TRY(r.gcand.alloc(sizeof(int) * (size_t)r.K1p * RK_CAP));
    PRE3_TRY(m.A.alloc((size_t)m.K1p * 128)); PRE3_TRY(m.B.alloc((size_t)m.K2p * 128));
    PRE3_TRY(m.na.alloc(sizeof(int) * m.K1p)); PRE3_TRY(m.nb.alloc(sizeof(int) * m.K2p));
    PRE3_TRY(m.ob.alloc(sizeof(double) * K1)); PRE3_TRY(m.os.alloc(sizeof(double) * K1)); PRE3_TRY(m.oa.alloc(sizeof(int32_t) * K1));
    PRE3_HIP(hipMemset(r.fl.p, 0, sizeof(int) * 4));
    hipLaunchKernelGGL((k_rank_pack<T>), dim3(r.K1p * 8 / 256), dim3(256), 0, 0, ND, K1, r.K1p, (const T *)r.L1.p, (v4i *)r.qh.p, (v4i *)r.ql.p,
                       (float *)r.nqU.p, (float *)r.nqL.p, (float *)r.nq.p, (int8_t *)m.A.p, (int *)m.na.p, (int *)r.fl.p);
    hipLaunchKernelGGL((k_rank_pack<T>), dim3(r.K2p * 8 / 256), dim3(256), 0, 0, ND, K2, r.K2p, (const T *)r.L2.p, (v4i *)r.dh.p, (v4i *)r.dl.p,
                       (float *)r.nU.p, (float *)r.nL.p, (float *)r.nd.p, (int8_t *)m.B.p, (int *)m.nb.p, (int *)r.fl.p);
    PRE3_HIP(hipGetLastError());
    int fl = 0;
    PRE3_HIP(hipMemcpy(&fl, r.fl.p, sizeof(int), hipMemcpyDeviceToHost));
    r.route = (fl & 1) ? 0 : ((fl & 2) == 0 && float_form() != 2) ? 1 : 2;
    return PRE3_OK;
}

template <typename T>
static int rank_run(RankMatch &r, int k2_offset, hipStream_t st, bool count = false)
{
    if (r.route == 1) return i8_run(r.m, k2_offset, st);
    PRE3_CHECK(r.route == 2, PRE3_E_STATE, "float-class matcher: the data is outside the ranked path's bounds");
    const char *fe = getenv("PRE3_MATCH_RANK_FORM");                      // 1 (default): tiled both ways, three launches; 0: one launch, database streamed per 16 queries
    static const int form2_default = 0;                                      // (measured, 4096 x 4096 unit-norm doubles: form 2 52.2 us per match, form 1 43.5)
    const int form = fe ? atoi(fe) : (form2_default ? 2 : 1);               // 2: 128 queries per workgroup, database staged through LDS (round 6); 1: k_rank_tiled; 0: k_match_rank
    if (form == 2 && r.nsl2 > 0) {
        static std::atomic<int> attr_ok{ 0 };
        if (attr_ok.load() == 0) {
            const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(k_rank_tiled2<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k_rank_tiled2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            attr_ok.store(e0 == hipSuccess && e1 == hipSuccess ? 1 : -1);
        }
        if (attr_ok.load() == 1) {
            dim3 g(r.K1p / R2_Q, r.nsl2), b(64 * RT_WAVES);
            hipLaunchKernelGGL((k_rank_tiled2<0>), g, b, 128 * 1024, st, r.K1, r.K2, r.K2p, r.nsl2, (const v4i *)r.qh.p, (const v4i *)r.ql.p, (const v4i *)r.dh.p, (const v4i *)r.dl.p,
                               (const float *)r.nq.p, (const float *)r.nU.p, (float *)r.pb.p, (float *)r.ps.p, r.K1p, (int *)r.gcnt.p, (int *)r.gcand.p);
            hipLaunchKernelGGL((k_rank_tiled2<1>), g, b, 128 * 1024, st, r.K1, r.K2, r.K2p, r.nsl2, (const v4i *)r.qh.p, (const v4i *)r.ql.p, (const v4i *)r.dh.p, (const v4i *)r.dl.p,
                               (const float *)r.nq.p, (const float *)r.nL.p, (float *)r.pb.p, (float *)r.ps.p, r.K1p, (int *)r.gcnt.p, (int *)r.gcand.p);
            hipLaunchKernelGGL((k_rank_tail<T>), dim3(ceil_div(r.K1, 4)), dim3(256), 0, st, r.ND, r.K1, r.K2, (const T *)r.L1.p, (const T *)r.L2.p, (const int *)r.gcnt.p,
                               (const int *)r.gcand.p, k2_offset, (double *)r.m.ob.p, (double *)r.m.os.p, (int32_t *)r.m.oa.p, count ? (int *)r.fl.p + 2 : nullptr);
            PRE3_HIP(hipGetLastError());
            return PRE3_OK;
        }
    }
    if (form != 0) {
        dim3 g(r.K1p / RT_Q, r.nsl), b(64 * RT_WAVES);
        hipLaunchKernelGGL((k_rank_tiled<0>), g, b, 0, st, r.K1, r.K2, r.K2p, r.nsl, (const v4i *)r.qh.p, (const v4i *)r.ql.p, (const v4i *)r.dh.p, (const v4i *)r.dl.p,
                           (const float *)r.nq.p, (const float *)r.nU.p, (float *)r.pb.p, (float *)r.ps.p, r.K1p, (int *)r.gcnt.p, (int *)r.gcand.p);
        hipLaunchKernelGGL((k_rank_tiled<1>), g, b, 0, st, r.K1, r.K2, r.K2p, r.nsl, (const v4i *)r.qh.p, (const v4i *)r.ql.p, (const v4i *)r.dh.p, (const v4i *)r.dl.p,
                           (const float *)r.nq.p, (const float *)r.nL.p, (float *)r.pb.p, (float *)r.ps.p, r.K1p, (int *)r.gcnt.p, (int *)r.gcand.p);
        hipLaunchKernelGGL((k_rank_tail<T>), dim3(ceil_div(r.K1, 4)), dim3(256), 0, st, r.ND, r.K1, r.K2, (const T *)r.L1.p, (const T *)r.L2.p, (const int *)r.gcnt.p,
                           (const int *)r.gcand.p, k2_offset, (double *)r.m.ob.p, (double *)r.m.os.p, (int32_t *)r.m.oa.p, count ? (int *)r.fl.p + 2 : nullptr);
        PRE3_HIP(hipGetLastError());
        return PRE3_OK;
    }
    hipLaunchKernelGGL((k_match_rank<T>), dim3(r.K1p / RK_Q), dim3(64 * RK_WAVES), 0, st, r.ND, r.K1, r.K2, r.K2p, (const v4i *)r.qh.p, (const v4i *)r.ql.p,
                       (const v4i *)r.dh.p, (const v4i *)r.dl.p, (const float *)r.nq.p, (const float *)r.nU.p, (const float *)r.nL.p, (const T *)r.L1.p,
                       (const T *)r.L2.p, k2_offset, (double *)r.m.ob.p, (double *)r.m.os.p, (int32_t *)r.m.oa.p, count ? (int *)r.fl.p + 2 : nullptr);
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// float / double classes: the ranked MFMA path when the shape and the data allow it, else the exact kernels
template <typename T>
static int partial_float(int ND, int K1, const T *L1, int K2, const T *L2, int k2_offset, double *best, double *second, int32_t *arg)
{
    if (!rank_applies<T>(ND, K1, K2)) return partial_exact<T, T>(ND, K1, L1, K2, L2, k2_offset, best, second, arg);
    RankMatch r;
    PRE3_TRY(rank_prepare<T>(r, ND, K1, L1, K2, L2));
    if (r.route == 0) return partial_exact<T, T>(ND, K1, L1, K2, L2, k2_offset, best, second, arg);
    PRE3_TRY(rank_run<T>(r, k2_offset, 0));
    PRE3_HIP(hipMemcpy(best, r.m.ob.p, sizeof(double) * K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(second, r.m.os.p, sizeof(double) * K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(arg, r.m.oa.p, sizeof(int32_t) * K1, hipMemcpyDeviceToHost));
    // (route 1, single class: int-valued sums < 2^24, so the float class's own bin-by-bin sums are the same integers)
    return PRE3_OK;
}

int match_partial(int device, int cls, int ND, int K1, const void *L1, int K2, const void *L2, int k2_offset, double *best, double *second,
                  int32_t *arg)
{
    PRE3_CHECK(ND > 0 && K1 >= 0 && K2 >= 0, PRE3_E_ARG, "siftmatch: bad sizes ND=%d K1=%d K2=%d", ND, K1, K2);
    PRE3_CHECK(K1 == 0 || (L1 && best && second && arg), PRE3_E_ARG, "siftmatch: null pointer");
    if (K1 == 0) return PRE3_OK;
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device %d", device); return PRE3_E_NODEVICE; }
    if (K2 == 0) {
        for (int i = 0; i < K1; ++i) { best[i] = (cls >= 2) ? 2147483647.0 : INFINITY; second[i] = best[i]; arg[i] = -1; }
        return PRE3_OK;
    }
    switch (cls) {
    case 0: return partial_float<double>(ND, K1, (const double *)L1, K2, (const double *)L2, k2_offset, best, second, arg);
    case 1: return partial_float<float>(ND, K1, (const float *)L1, K2, (const float *)L2, k2_offset, best, second, arg);
    case 2: return partial_i8<uint8_t>(ND, K1, (const uint8_t *)L1, K2, (const uint8_t *)L2, 128, k2_offset, best, second, arg);
    case 3: return partial_i8<int8_t>(ND, K1, (const int8_t *)L1, K2, (const int8_t *)L2, 0, k2_offset, best, second, arg);
    default: set_error("siftmatch: unsupported class %d", cls); return PRE3_E_ARG;
    }
}

// ------------------------------------------------------------------------------------------------
// IC search on the device: matching_sift_based.m:104-149 without leaving HBM.
// ------------------------------------------------------------------------------------------------
// index_in_info (matching_sift_based.m:108-114): the predicted landmarks in map order.  One wave.
__global__ void k_ic_stack(int N, const int32_t *__restrict__ has_h, int32_t *__restrict__ pred, int32_t *__restrict__ counts,
                           int32_t *__restrict__ newk2)
{
    const int lane = threadIdx.x;
    int base = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        const int f = i < N && has_h[i] != 0;
        if (i < N) newk2[i] = -1;
        const unsigned long long b = __ballot(f);
        if (f) pred[base + __popcll(b & ((1ull << lane) - 1))] = i;
        base += __popcll(b);
    }
    if (lane == 0) { counts[0] = base; counts[1] = 0; counts[2] = 0; }
}

// Lowe's test (siftmatch.c:122, in float), then the window gate of matching_sift_based.m:119-133 on the i-th match
// (quirk Q5 needs the RANK of the match: S is read from index_in_info(i)).  One wave, ordered.
__global__ __launch_bounds__(512) void k_ic_gate(const int32_t *__restrict__ pred, const double *__restrict__ best,
                          const double *__restrict__ second, const int32_t *__restrict__ arg, float thresh, int strict,
                          const double *__restrict__ pos, const double *__restrict__ h, const double *__restrict__ S,
                          const int32_t *__restrict__ has_S, double *__restrict__ z, int32_t *__restrict__ ic,
                          int32_t *__restrict__ pairs, int32_t *__restrict__ newk2, int32_t *counts,
                          int32_t *__restrict__ meas_out, double *__restrict__ z_out)
{
    // one workgroup, 512 predictions per pass (all of them at N <= 512): the rank of a match among the matches and of an accepted
    // measurement among the accepted ones come from wave ballots + a prefix over the 8 waves, so the three dependent load levels
    // (arg -> pos / pred -> S, h) are paid once per pass instead of once per 64 predictions
    __shared__ int s_ok[8], s_acc[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, npred = counts[0];
    int base = 0, m = 0;
    for (int k0 = 0; k0 < npred; k0 += 512) {
        const int k1 = k0 + tid;
        int ok = 0, k2 = -1;
        if (k1 < npred) {
            k2 = arg[k1];
            ok = k2 >= 0 && thresh * (float)best[k1] <= (float)second[k1];
        }
        const unsigned long long b = __ballot(ok);
        if (lane == 0) s_ok[wv] = __popcll(b);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const int cw = s_ok[w]; if (w < wv) off += cw; tot += cw; }
        int acc = 0;
        if (ok) {
            const int c = base + off + __popcll(b & ((1ull << lane) - 1));
            const int lm = pred[k1], slm = strict ? pred[c] : lm;
            const double half = has_S[slm] ? ceil(3 * sqrt(S[4 * slm])) : 40.0;
            const double dx = pos[4 * (size_t)k2] - h[2 * lm], dy = pos[4 * (size_t)k2 + 1] - h[2 * lm + 1];
            acc = sqrt(dx * dx + dy * dy) <= half;
            pairs[3 * c] = k1; pairs[3 * c + 1] = k2; pairs[3 * c + 2] = acc;
            if (acc) { ic[lm] = 1; z[2 * lm] = pos[4 * (size_t)k2]; z[2 * lm + 1] = pos[4 * (size_t)k2 + 1]; newk2[lm] = k2; }
        }
        base += tot;
        // accepted matches arrive in increasing landmark order (pred is sorted): compact the measurement list for the host
        const unsigned long long ab = __ballot(acc);
        if (lane == 0) s_acc[wv] = __popcll(ab);
        __syncthreads();
        int off2 = 0, tot2 = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { const int cw = s_acc[w]; if (w < wv) off2 += cw; tot2 += cw; }
        if (acc) {
            const int pm = m + off2 + __popcll(ab & ((1ull << lane) - 1));
            meas_out[pm] = pred[k1]; z_out[2 * pm] = pos[4 * (size_t)k2]; z_out[2 * pm + 1] = pos[4 * (size_t)k2 + 1];
        }
        m += tot2;
        __syncthreads();                          // s_ok / s_acc are rewritten by the next pass
    }
    if (tid == 0) { counts[1] = base; counts[2] = m; }
}

// matching_sift_based.m:135: the accepted landmark takes the scan's descriptor
__global__ void k_ic_refresh(const int32_t *__restrict__ newk2, const double *__restrict__ scan_desc, double *__restrict__ bank)
{
    const int lm = blockIdx.x, k2 = newk2[lm];
    if (k2 < 0) return;
    bank[(size_t)lm * DESC_DIM + threadIdx.x] = scan_desc[(size_t)k2 * DESC_DIM + threadIdx.x];
}

__global__ void k_bank_gather(const int32_t *__restrict__ src, const double *__restrict__ bank, double *__restrict__ out)
{
    const int s = src[blockIdx.x];
    out[(size_t)blockIdx.x * DESC_DIM + threadIdx.x] = s >= 0 ? bank[(size_t)s * DESC_DIM + threadIdx.x] : 0.0;
}

// ---- the matcher of matching_sift_based.m:118 on the matrix cores inside the IC search (round 3).  The scan's descriptors are packed into
// bf16 planes when they arrive (pre3_set_scan), the stacked query matrix des1 (matching_sift_based.m:108-114: the predicted landmarks'
// descriptors in map order) is gathered from the bank through the device-side list and packed per call; k_rank_tiled<0/1> + k_rank_tail then
// give (best, second, arg) bit-identical to the exact kernel's (tests/test_gpu_icsearch.py runs both).  Descriptors outside the ranked
// route's bounds (NaN / Inf, |x| > 2^60, 0 < |x| < 2^-40), small problems and PRE3_IC_RANK=0 keep the exact VALU kernel.
struct IcRank { RankMatch r; int K1cap = 0, K2 = -1; bool scan_ok = false, packed = false; };       // scan_ok: inside the ranked route's bounds; packed: its planes exist

// des1[q] = bank[pred[q]] (rows beyond the device-side count repeat older list entries: in range, never read back)
__global__ __launch_bounds__(DESC_DIM) void k_ic_gather_q(const int32_t *__restrict__ pred, const double *__restrict__ bank, double *__restrict__ des1)
{
    des1[(size_t)blockIdx.x * DESC_DIM + threadIdx.x] = bank[(size_t)pred[blockIdx.x] * DESC_DIM + threadIdx.x];
}

static bool ic_fused_usable(const pre3_ctx *c);
void ic_rank_free(pre3_ctx *c) { delete static_cast<IcRank *>(c->ic_rank); c->ic_rank = nullptr; }

// the scan has just been uploaded: pack it for the ranked route (or note that it cannot take it)
int ic_rank_set_scan(pre3_ctx *c, bool in_bounds)
{
    const int K2 = c->scan_K2, K1 = c->capN;
    IcRank *ic = static_cast<IcRank *>(c->ic_rank);
    if (ic) ic->scan_ok = false;
    if (K2 <= 0 || K1 <= 0) return PRE3_OK;
    if (!ic || ic->K1cap < K1 || round_up(K2, 128) > ic->r.K2p) {
        ic_rank_free(c);
        ic = new (std::nothrow) IcRank();
        PRE3_CHECK(ic != nullptr, PRE3_E_NOMEM, "out of host memory");
        c->ic_rank = ic;
        RankMatch &r = ic->r;
        r.ND = DESC_DIM; r.K1p = round_up(K1, 128); r.K2p = round_up(c->scan_cap > K2 ? c->scan_cap : K2, 128);
        ic->K1cap = K1;
        PRE3_TRY(r.L1.alloc(sizeof(double) * (size_t)DESC_DIM * r.K1p));
        PRE3_TRY(r.qh.alloc((size_t)r.K1p * 256)); PRE3_TRY(r.ql.alloc((size_t)r.K1p * 256)); PRE3_TRY(r.dh.alloc((size_t)r.K2p * 256)); PRE3_TRY(r.dl.alloc((size_t)r.K2p * 256));
        PRE3_TRY(r.nq.alloc(sizeof(float) * r.K1p)); PRE3_TRY(r.nqU.alloc(sizeof(float) * r.K1p)); PRE3_TRY(r.nqL.alloc(sizeof(float) * r.K1p));
        PRE3_TRY(r.nU.alloc(sizeof(float) * r.K2p)); PRE3_TRY(r.nL.alloc(sizeof(float) * r.K2p)); PRE3_TRY(r.nd.alloc(sizeof(float) * r.K2p));
        PRE3_TRY(r.fl.alloc(sizeof(int) * 4));
        PRE3_TRY(r.pb.alloc(sizeof(float) * (size_t)64 * r.K1p)); PRE3_TRY(r.ps.alloc(sizeof(float) * (size_t)64 * r.K1p));
        PRE3_TRY(r.gcnt.alloc(sizeof(int) * r.K1p)); PRE3_TRY(r.gcand.alloc(sizeof(int) * (size_t)r.K1p * RK_CAP));
        PRE3_TRY(r.m.A.alloc((size_t)r.K1p * 128)); PRE3_TRY(r.m.B.alloc((size_t)r.K2p * 128));
        PRE3_TRY(r.m.na.alloc(sizeof(int) * r.K1p)); PRE3_TRY(r.m.nb.alloc(sizeof(int) * r.K2p));
        PRE3_TRY(r.m.ob.alloc(sizeof(double) * r.K1p)); PRE3_TRY(r.m.os.alloc(sizeof(double) * r.K1p)); PRE3_TRY(r.m.oa.alloc(sizeof(int32_t) * r.K1p));
    }
    RankMatch &r = ic->r;
    const int K2p = round_up(K2, 128);
    ic->K2 = K2;
    ic->scan_ok = in_bounds;
    ic->packed = false;
    if (ic_fused_usable(c)) return PRE3_OK;             // the search will take the two-launch route: no planes needed (were a fill + k_rank_pack, 12 us per frame)
    // (on the context's stream, behind the upload; whether the scan is inside the ranked route's bounds is the caller's host-side check of
    // the same descriptors -- k_rank_pack's own flag word is not read back)
    PRE3_HIP(hipMemsetAsync(r.fl.p, 0, sizeof(int) * 4, c->stream));
    hipLaunchKernelGGL((k_rank_pack<double>), dim3(K2p * 8 / 256), dim3(256), 0, c->stream, DESC_DIM, K2, K2p, (const double *)c->scan_desc, (v4i *)r.dh.p, (v4i *)r.dl.p,
                       (float *)r.nU.p, (float *)r.nL.p, (float *)r.nd.p, (int8_t *)r.m.B.p, (int *)r.m.nb.p, (int *)r.fl.p);
    PRE3_HIP(hipGetLastError());
    ic->packed = true;
    return PRE3_OK;
}

static bool ic_rank_usable(const pre3_ctx *c)
{
    const char *e = getenv("PRE3_IC_RANK");                    // 0: the exact VALU kernel (A/B, tests); read per call
    if (e && atoi(e) == 0) return false;
    const IcRank *ic = static_cast<const IcRank *>(c->ic_rank);
    return ic && ic->scan_ok && ic->packed && ic->K2 == c->scan_K2 && c->bank_ok && c->N <= ic->K1cap && rank_applies<double>(DESC_DIM, c->N, c->scan_K2);
}

// ---- the fused route (round 5): at the reference's real sizes (N = 500 landmarks against a 600-keypoint scan: 3e5 pairs, 1.2e8 fp64 operations) the
// rank / tail machinery and the bookkeeping launches around it were the cost -- ten launches, 62 us of kernels.  Two launches instead:
//   k_ic_match_small  siftmatch.c:97-116 exactly (every pair's bins in order, no contraction), a 32 x 32 pair tile per workgroup so that the chip's
//                     fp64 vector pipes all have work (the 64 x 64 tiles of k_match_exact_tiled are 80 workgroups here); every landmark of the map is a
//                     query -- the stacking of matching_sift_based.m:108-114 needs a prefix over the map and is left to the gate, tiles without a
//                     predicted landmark leave at once; partial (best, second, arg) per (column tile, landmark);
//   k_ic_gate_fused   ONE workgroup: index_in_info by ballot compaction, the column tiles merged in scan order, Lowe's test, the window gate with its
//                     rank quirk (k_ic_gate's logic), the accepted landmarks' descriptor refresh (matching_sift_based.m:135), and the result block
//                     written straight into the host's mapped block and announced through the mailbox (was k_ic_stack, k_match_reduce_f / the five
//                     launches of the ranked route, k_ic_gate, k_ic_refresh, k_inbox_pull).
__global__ __launch_bounds__(256) void k_ic_match_small(IcMatchRide r)
{
    __shared__ double Qs[ICS_T][ICS_LD], Bs[ICS_T][ICS_LD];
    ic_match_tile(r, blockIdx.x, Qs, Bs);
}

constexpr int ICG_NTH = 1024, ICG_MAXN = 4096;
struct IcGateF {
    int N, ntn, capN, strict; float thresh;
    const int32_t *has_h; const double *pb, *ps; const int32_t *pa;
    const double *pos, *h, *S; const int32_t *has_S;
    double *z; int32_t *ic;
    const double *scan_desc; double *bank;
    int32_t *res;                                   // the host's mapped result block [counts 4 | meas capN | pairs 3 capN | z 2 capN doubles]
    int32_t *counts_dev;                            // the device's copy of the counts (what k_ic_gate leaves)
    int32_t *mail; int32_t seq; int slot;
};
__global__ __launch_bounds__(ICG_NTH) void k_ic_gate_fused(IcGateF a)
{
    __shared__ int s_pred[ICG_MAXN], s_lm[ICG_MAXN], s_k2[ICG_MAXN];
    __shared__ int s_cnt[ICG_NTH / 64], s_cnt2[ICG_NTH / 64];
    constexpr int NW = ICG_NTH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t *meas_out = a.res + 4, *pairs = a.res + 4 + a.capN;
    double *z_out = reinterpret_cast<double *>(a.res + 4 + 4 * (size_t)a.capN);
    // index_in_info (matching_sift_based.m:108-114): the predicted landmarks in map order
    int npred = 0;
    for (int i0 = 0; i0 < a.N; i0 += ICG_NTH) {
        const int i = i0 + tid;
        const int f = i < a.N && a.has_h[i] != 0;
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_cnt[wv] = __popcll(b);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int cw = s_cnt[w]; if (w < wv) off += cw; tot += cw; }
        if (f) s_pred[npred + off + __popcll(b & ((1ull << lane) - 1))] = i;
        npred += tot;
        __syncthreads();
    }
    // Lowe's test (siftmatch.c:122, in float) on the merged column tiles, then the window gate of matching_sift_based.m:119-133 on the i-th match
    // (quirk Q5 needs the RANK of the match: S is read from index_in_info(i))
    int base = 0, m = 0;
    for (int p0 = 0; p0 < npred; p0 += ICG_NTH) {
        const int k1 = p0 + tid;
        int ok = 0, k2 = -1, lm = 0;
        if (k1 < npred) {
            lm = s_pred[k1];
            double best = acc_max<double>(), second = acc_max<double>();
            for (int t0 = 0; t0 < a.ntn; t0 += 16) {            // sixteen tiles' partials in flight (lanes = consecutive landmarks: whole lines), merged in scan order
                double ob[16], os[16]; int oa[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const size_t o = (size_t)(t0 + u < a.ntn ? t0 + u : t0) * a.N + lm;
                    ob[u] = a.pb[o]; os[u] = a.ps[o]; oa[u] = t0 + u < a.ntn ? a.pa[o] : -1;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) merge3(best, second, k2, ob[u], os[u], oa[u]);
            }
            ok = k2 >= 0 && a.thresh * (float)best <= (float)second;
        }
        const unsigned long long b = __ballot(ok);
        if (lane == 0) s_cnt[wv] = __popcll(b);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int cw = s_cnt[w]; if (w < wv) off += cw; tot += cw; }
        int acc = 0;
        double px = 0, py = 0;
        if (ok) {
            const int c = base + off + __popcll(b & ((1ull << lane) - 1));
            const int slm = a.strict ? s_pred[c] : lm;
            const double half = a.has_S[slm] ? ceil(3 * sqrt(a.S[4 * slm])) : 40.0;
            px = a.pos[4 * (size_t)k2]; py = a.pos[4 * (size_t)k2 + 1];
            const double dx = px - a.h[2 * lm], dy = py - a.h[2 * lm + 1];
            acc = sqrt(dx * dx + dy * dy) <= half;
            pairs[3 * c] = k1; pairs[3 * c + 1] = k2; pairs[3 * c + 2] = acc;
            if (acc) { a.ic[lm] = 1; a.z[2 * lm] = px; a.z[2 * lm + 1] = py; }
        }
        base += tot;
        // accepted matches arrive in increasing landmark order (the list is sorted): the measurement list for the host, and for the refresh below
        const unsigned long long ab = __ballot(acc);
        if (lane == 0) s_cnt2[wv] = __popcll(ab);
        __syncthreads();
        int off2 = 0, tot2 = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int cw = s_cnt2[w]; if (w < wv) off2 += cw; tot2 += cw; }
        if (acc) {
            const int pm = m + off2 + __popcll(ab & ((1ull << lane) - 1));
            meas_out[pm] = lm; z_out[2 * pm] = px; z_out[2 * pm + 1] = py;
            s_lm[pm] = lm; s_k2[pm] = k2;
        }
        m += tot2;
        __syncthreads();                          // (s_cnt / s_cnt2 are rewritten by the next pass; s_lm / s_k2 are read below)
    }
    if (tid == 0) {
        a.res[0] = npred; a.res[1] = base; a.res[2] = m; a.res[3] = 0;
        a.counts_dev[0] = npred; a.counts_dev[1] = base; a.counts_dev[2] = m;
    }
    // the result block is in the host's memory: every lane's stores first, then the sequence number -- before the refresh, which the host does not
    // wait for (it draws its hypotheses meanwhile; the next launch on the stream is behind this one anyway)
    __threadfence_system();
    __syncthreads();
    if (tid == 0) { __threadfence_system(); __hip_atomic_store(a.mail + a.slot, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    // matching_sift_based.m:135: the accepted landmark takes the scan's descriptor (a descriptor per wave, eight in flight, 16 bytes per lane)
    {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        for (int j0 = 0; j0 < m; j0 += 8 * NW) {
            d2_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * NW + wv;
                if (j < m) v[u] = reinterpret_cast<const d2_t *>(a.scan_desc + (size_t)s_k2[j] * DESC_DIM)[lane];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * NW + wv;
                if (j < m) reinterpret_cast<d2_t *>(a.bank + (size_t)s_lm[j] * DESC_DIM)[lane] = v[u];
            }
        }
    }
}

static bool ic_fused_usable(const pre3_ctx *c)
{
    const char *e = getenv("PRE3_IC_FUSED");                   // 0: the ranked / exact routes (A/B, tests); read per call
    if (e && atoi(e) == 0) return false;
    const int ntn = ceil_div(c->scan_K2, ICS_T);
    return c->scan_K2 > 0 && c->N <= ICG_MAXN && ntn <= ICS_MAXT && c->ic_pb != nullptr && c->ic_pcap >= (size_t)c->capN * ICS_MAXT
           && (size_t)c->N * c->scan_K2 <= ((size_t)1 << 20);
}

// the fused route's matcher as riders of another launch (pre3_ic_search: the projection + S_i launch in front of the gate -- nothing in it reads or
// writes the descriptors, so the two run side by side and a kernel boundary goes; PRE3_IC_RIDE=0: a launch of its own, which can skip the tiles
// without a predicted landmark)
IcMatchRide ic_match_ride(const pre3_ctx *c)
{
    static const int ride_env = getenv("PRE3_IC_RIDE") ? atoi(getenv("PRE3_IC_RIDE")) : 1;
    IcMatchRide r{};
    if (!ride_env || !ic_fused_usable(c)) return r;
    r.ntn = ceil_div(c->scan_K2, ICS_T); r.N = c->N; r.K2 = c->scan_K2; r.n_blocks = r.ntn * ceil_div(c->N, ICS_T);
    r.bank = c->bank; r.scan = c->scan_desc; r.has_h = nullptr; r.pb = c->ic_pb; r.ps = c->ic_ps; r.pa = c->ic_pa;
    return r;
}

// the fused route: the matcher (unless it rode in the projection's launch) and the gate; the gate publishes mailbox word `slot` = seq once the result
// block is in c->ic_result_host
int launch_ic_search_fused(pre3_ctx *c, double thresh, int strict, int32_t seq, int slot, bool matched)
{
    const int N = c->N, K2 = c->scan_K2, ntn = ceil_div(K2, ICS_T);
    if (!matched) {
        IcMatchRide r{ ntn * ceil_div(N, ICS_T), ntn, N, K2, c->bank, c->scan_desc, c->lm.has_h, c->ic_pb, c->ic_ps, c->ic_pa };
        hipLaunchKernelGGL(k_ic_match_small, dim3(r.n_blocks), dim3(256), 0, c->stream, r);
    }
    IcGateF a{};
    a.N = N; a.ntn = ntn; a.capN = c->capN; a.strict = strict; a.thresh = (float)thresh;
    a.has_h = c->lm.has_h; a.pb = c->ic_pb; a.ps = c->ic_ps; a.pa = c->ic_pa;
    a.pos = c->scan_pos; a.h = c->lm.h; a.S = c->lm.S; a.has_S = c->lm.has_S; a.z = c->lm.z; a.ic = c->lm.ic;
    a.scan_desc = c->scan_desc; a.bank = c->bank;
    a.res = static_cast<int32_t *>(c->ic_result_host_dev); a.counts_dev = c->ic_counts; a.mail = c->mail_dev; a.seq = seq; a.slot = slot;
    hipLaunchKernelGGL(k_ic_gate_fused, dim3(1), dim3(ICG_NTH), 0, c->stream, a);
    PRE3_HIP(hipGetLastError());
    c->ic_route = 2;
    // (accepted landmarks take the scan's descriptors: a scan outside the ranked route's bounds takes the bank with it)
    const IcRank *ic = static_cast<const IcRank *>(c->ic_rank);
    if (!(ic && ic->scan_ok && ic->K2 == c->scan_K2)) c->bank_ok = false;
    return PRE3_OK;
}
bool ic_search_fused_applies(const pre3_ctx *c) { return ic_fused_usable(c); }

int launch_ic_search(pre3_ctx *c, double thresh, int strict)
{
    const int N = c->N;
    hipLaunchKernelGGL(k_ic_stack, dim3(1), dim3(64), 0, c->stream, N, c->lm.has_h, c->ic_pred, c->ic_counts, c->ic_newk2);
    if (c->scan_K2 > 0) {
        const double *best = c->ic_best, *second = c->ic_second;
        const int32_t *arg = c->ic_arg;
        c->ic_last_ranked = ic_rank_usable(c);
        c->ic_route = c->ic_last_ranked ? 1 : 0;
        if (c->ic_last_ranked) {
            RankMatch &r = static_cast<IcRank *>(c->ic_rank)->r;
            const int K2 = c->scan_K2;
            r.K1 = N; r.K2 = K2; r.K1p = round_up(N, 128); r.K2p = round_up(K2, 128); r.route = 2;
            r.m.K1 = N; r.m.K2 = K2; r.m.K1p = r.K1p; r.m.K2p = r.K2p;
            {   // the tiled form's database slices: about one workgroup per CU (as rank_prepare)
                const int groups = r.K1p / RT_Q, nblk = r.K2p / 16;
                r.nsl = std::max(1, std::min(std::min(ceil_div(c->num_cus, groups), nblk / RT_WAVES), 64));
            }
            hipLaunchKernelGGL(k_ic_gather_q, dim3(N), dim3(DESC_DIM), 0, c->stream, (const int32_t *)c->ic_pred, (const double *)c->bank, (double *)r.L1.p);
            hipLaunchKernelGGL((k_rank_pack<double>), dim3(r.K1p * 8 / 256), dim3(256), 0, c->stream, DESC_DIM, N, r.K1p, (const double *)r.L1.p, (v4i *)r.qh.p, (v4i *)r.ql.p,
                               (float *)r.nqU.p, (float *)r.nqL.p, (float *)r.nq.p, (int8_t *)r.m.A.p, (int *)r.m.na.p, (int *)r.fl.p + 1);
            // (the tail re-evaluates from the original descriptors: des1 and the scan as the caller gave them)
            const void *saveL2 = r.L2.p; r.L2.p = c->scan_desc;
            const int rc = rank_run<double>(r, 0, c->stream);
            r.L2.p = const_cast<void *>(saveL2);
            PRE3_TRY(rc);
            best = (const double *)r.m.ob.p; second = (const double *)r.m.os.p; arg = (const int32_t *)r.m.oa.p;
        } else {
        const int ntn = ceil_div(c->scan_K2, 64);
        hipLaunchKernelGGL((k_match_exact_tiled<double>), dim3(ntn, ceil_div(N, 64)), dim3(256), 0, c->stream, DESC_DIM, N, c->scan_K2,
                           (const double *)c->bank, (const double *)c->scan_desc, c->ic_pb, c->ic_ps, c->ic_pa, ntn,
                           (const int32_t *)c->ic_pred, (const int32_t *)c->ic_counts);
        // rows >= the device-side query count hold stale partials; k_ic_gate only reads the first counts[0] results
        hipLaunchKernelGGL((k_match_reduce_f<double>), dim3(ceil_div(16 * N, 256)), dim3(256), 0, c->stream, N, ntn, (const double *)c->ic_pb,
                           (const double *)c->ic_ps, (const int32_t *)c->ic_pa, 0, c->ic_best, c->ic_second, c->ic_arg);
        }
        hipLaunchKernelGGL(k_ic_gate, dim3(1), dim3(512), 0, c->stream, c->ic_pred, best, second, arg,
                           (float)thresh, strict, c->scan_pos, c->lm.h, c->lm.S, c->lm.has_S, c->lm.z, c->lm.ic, c->ic_pairs, c->ic_newk2,
                           c->ic_counts, c->ic_counts + 4, (double *)(c->ic_counts + 4 + 4 * (size_t)c->capN));
        hipLaunchKernelGGL(k_ic_refresh, dim3(N), dim3(DESC_DIM), 0, c->stream, c->ic_newk2, c->scan_desc, c->bank);
        // (accepted landmarks take the scan's descriptors: a scan outside the ranked route's bounds takes the bank with it)
        const IcRank *ic = static_cast<const IcRank *>(c->ic_rank);
        if (!(ic && ic->scan_ok && ic->K2 == c->scan_K2)) c->bank_ok = false;
    }
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}

// re-lay the descriptor bank after map management: new landmark i takes the descriptor of old landmark src[i] (-1: zeros)
int launch_bank_gather(pre3_ctx *c, int N_new, const int32_t *src_host)
{
    if (!c->bank || N_new == 0) return PRE3_OK;
    PRE3_HIP(hipMemcpyAsync(c->bank_src, src_host, sizeof(int32_t) * N_new, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_bank_gather, dim3(N_new), dim3(DESC_DIM), 0, c->stream, c->bank_src, c->bank, c->bank_alt);
    PRE3_HIP(hipGetLastError());
    PRE3_TRY(stream_drain(c, __func__));       // src_host is the caller's stack vector
    std::swap(c->bank, c->bank_alt);
    return PRE3_OK;
}

int knn_run(int device, int D, int N, const double *data, int M, const double *query, int k, double *ids, double *dist)
{
    PRE3_CHECK(D > 0 && N > 0 && M >= 0 && k >= 1 && k <= N, PRE3_E_ARG, "kNearestNeighbors: bad sizes D=%d N=%d M=%d k=%d", D, N, M, k);
    if (M == 0) return PRE3_OK;
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device %d", device); return PRE3_E_NODEVICE; }
    DevBuf dd, dq, sc, di, ds;
    PRE3_TRY(dd.alloc(sizeof(double) * (size_t)N * D)); PRE3_TRY(dq.alloc(sizeof(double) * (size_t)M * D));
    PRE3_TRY(sc.alloc(sizeof(double) * (size_t)M * N)); PRE3_TRY(di.alloc(sizeof(double) * (size_t)M * k)); PRE3_TRY(ds.alloc(sizeof(double) * (size_t)M * k));
    PRE3_HIP(hipMemcpy(dd.p, data, sizeof(double) * (size_t)N * D, hipMemcpyHostToDevice));
    PRE3_HIP(hipMemcpy(dq.p, query, sizeof(double) * (size_t)M * D, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_knn, dim3(M), dim3(256), 0, 0, D, N, M, (const double *)dd.p, (const double *)dq.p, k, (double *)sc.p, (double *)di.p, (double *)ds.p);
    PRE3_HIP(hipGetLastError());
    PRE3_HIP(hipMemcpy(ids, di.p, sizeof(double) * (size_t)M * k, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(dist, ds.p, sizeof(double) * (size_t)M * k, hipMemcpyDeviceToHost));
    return PRE3_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-resident database shard of the sharded matcher (DESIGN.md "multi-GPU", C2): the queries (replicated) and this GPU's slice of
// the database stay packed in HBM; a match is the distance kernel on the slice, an all-gather of the per-query partials (done by the
// caller on DEVICE memory: RCCL), and the merge + Lowe's test + ordered compaction on the device.  Only the final pair list crosses PCIe.
// ------------------------------------------------------------------------------------------------
struct MatchShard {
    I8Match m; RankMatch r; int cls = 2;    // cls 2: uint8 operands in m; 0 / 1: double / float descriptors in r (its outputs are r.m.ob / os / oa)
    int device = 0, k2_offset = 0;
    I8Match &out() { return cls == 2 ? m : r.m; }
    DevBuf part;            // double [3][K1]: best | second | arg (as double) -- the all-gather payload
    DevBuf res;             // double [1 + 3 K1]: count | pairs (2 K1) | scores (K1)
    std::vector<double> host_res;
    // pre3_match_shard_match: the whole match on one stream of the shard's own, result block written to pinned host memory by the merge kernel
    void *comm = nullptr;   // borrowed (pre3_match_shard_set_comm)
    hipStream_t st = nullptr;
    DevBuf raw; int raw_stride = 0;                           // [best | second | arg] of the slice, the distance kernel's outputs live here (+ the MISSING word behind arg)
    bool missing_set = false, test_fail = false;                // (test_fail: pre3_match_shard_test_stall(s, 2) makes the next match's distance kernels "fail")
    DevBuf gathered; int gathered_world = 0;
    double *res_host = nullptr, *res_host_dev = nullptr;      // [1 + 3 K1] + one int32 sequence word behind it
    int32_t seq = 0;
    ~MatchShard() { if (st) (void)hipStreamDestroy(st); if (res_host) (void)hipHostFree(res_host); }
};

__global__ void k_shard_pack(int K1, const double *__restrict__ b, const double *__restrict__ s2, const int32_t *__restrict__ a, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K1) { out[i] = b[i]; out[K1 + i] = s2[i]; out[2 * K1 + i] = (double)a[i]; }
}

// pre3_siftmatch_merge on the device for the integer classes: per query the G partials are merged (ties -> lowest global index), the ratio
// test is done in float on int-valued distances (siftmatch.c:122), and the matches are compacted in increasing k1 (one workgroup, ballots).
// raw_stride > 0 (pre3_match_shard_match): a rank's block is the distance kernel's own output, [best f64 | second f64 | arg int32] with
// `raw_stride` entries per array (no packing launch in front of the all-gather); 0: double[3][K1] with the arg as a double (pre3_match_shard_run)
__global__ __launch_bounds__(1024) void k_shard_merge(int G, int K1, const double *__restrict__ gathered, float thresh, double *__restrict__ res,
                                                      int32_t *mail = nullptr, int32_t seq = 0, int raw_stride = 0)
{
    // four blocks of 1024 queries per pass, every load of the pass issued before the first use (the loop of one block per pass was four
    // dependent rounds of load latency + three barriers each: 12 us at K1 = 4096)
    __shared__ int s_cnt[4][16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int s_missing;
    if (tid == 0) { s_base = 0; s_missing = 0; }
    __syncthreads();
    const int st = raw_stride ? raw_stride : K1;
    // raw blocks (pre3_match_shard_match): the int32 word behind a rank's arg array says that the rank's slice is MISSING (its distance kernels
    // failed before the all-gather, which it entered all the same so that nobody waits for it): every rank then fails this match
    if (raw_stride && tid < G && reinterpret_cast<const int32_t *>(gathered + (size_t)tid * 3 * st + 2 * (size_t)st)[st] != 0) atomicAdd(&s_missing, 1);
    for (int k0 = 0; k0 < K1; k0 += 4096) {
        int ok[4], Kq[4]; double Bq[4];
        unsigned long long bal[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k1 = k0 + q * 1024 + tid;
            int K = -1; double B = 0, S2 = 0;
            ok[q] = 0;
            if (k1 < K1) {
                for (int g = 0; g < G; ++g) {
                    const double *p = gathered + (size_t)g * 3 * st;
                    const double ob = p[k1], os = p[st + k1];
                    const int oa = raw_stride ? reinterpret_cast<const int32_t *>(p + 2 * (size_t)st)[k1] : (int)p[2 * K1 + k1];
                    if (oa < 0) continue;
                    if (K < 0) { B = ob; S2 = os; K = oa; continue; }
                    if (ob < B || (ob == B && oa < K)) { S2 = os < B ? os : B; B = ob; K = oa; }
                    else { S2 = ob < S2 ? ob : S2; }
                }
                if (K >= 0) ok[q] = thresh * (float)B <= (float)S2;    // siftmatch.c:122 (integer classes: the doubles hold the int distances exactly)
            }
            Kq[q] = K; Bq[q] = B;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { bal[q] = __ballot(ok[q]); if (lane == 0) s_cnt[q][wv] = __popcll(bal[q]); }
        __syncthreads();
        int run = s_base;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int off = run, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { const int c = s_cnt[q][w]; if (w < wv) off += c; tot += c; }
            if (ok[q]) {
                const int k1 = k0 + q * 1024 + tid;
                const int pos = off + __popcll(bal[q] & ((1ull << lane) - 1ull));
                if (mail) {         // pinned host memory: [pairs | scores | count], a pair is ONE 16-byte store (a wave's pairs fill whole PCIe writes)
                    typedef double d2_t __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<d2_t *>(res + 2 * (size_t)pos) = d2_t{ (double)(k1 + 1), (double)(Kq[q] + 1) };
                    res[2 * (size_t)K1 + pos] = Bq[q];
                } else { res[1 + 2 * pos] = k1 + 1; res[2 + 2 * pos] = Kq[q] + 1; res[1 + 2 * (size_t)K1 + pos] = Bq[q]; }
            }
            run += tot;
        }
        __syncthreads();
        if (tid == 0) s_base = run;
        __syncthreads();
    }
    if (mail) {             // res is pinned host memory (uncached: the stores go straight out): every wave's stores have been acknowledged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // before the word the host polls is written (a system-scope fence per wave
        __syncthreads();                                      // would also write the L2 back, once per wave)
    }
    if (tid == 0) {
        res[mail ? 3 * (size_t)K1 : 0] = (double)s_base;
        if (mail) { mail[1] = s_missing; __threadfence_system(); __hip_atomic_store(mail, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

void *match_shard_create(int device, int cls, int ND, int K1, const void *L1, int K2, const void *L2, int k2_offset)
{
    if (ND <= 0 || K1 <= 0 || K2 < 0 || !L1 || (K2 && !L2) || cls < 0 || cls > 2) { set_error("match shard: bad arguments"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("no HIP device %d", device); return nullptr; }
    MatchShard *sh = new MatchShard();
    sh->device = device; sh->k2_offset = k2_offset; sh->cls = cls;
    int rc = PRE3_OK;
    if (cls == 2) rc = i8_prepare(sh->m, ND, K1, (const uint8_t *)L1, K2, (const uint8_t *)L2, 128);
    else {
        // double / float descriptors (what matching_sift_based.m:104-118 passes): the matrix-core routes of the float classes, resident
        const bool ok = K2 > 0 && (cls == 0 ? rank_applies<double>(ND, K1, K2) : rank_applies<float>(ND, K1, K2));
        if (!ok) { set_error("match shard: this shape is not on a matrix-core path for class %d (ND <= 128, K1 * K2_local >= 65536): use the host-array form", cls); rc = PRE3_E_ARG; }
        else rc = cls == 0 ? rank_prepare<double>(sh->r, ND, K1, (const double *)L1, K2, (const double *)L2) : rank_prepare<float>(sh->r, ND, K1, (const float *)L1, K2, (const float *)L2);
        if (rc == PRE3_OK && sh->r.route == 0) { set_error("match shard: descriptors outside the ranked path's bounds (NaN / Inf / magnitude): use the host-array form"); rc = PRE3_E_ARG; }
    }
    if (rc != PRE3_OK || sh->part.alloc(sizeof(double) * 3 * (size_t)K1) != PRE3_OK || sh->res.alloc(sizeof(double) * (1 + 3 * (size_t)K1)) != PRE3_OK) { delete sh; return nullptr; }
    // the kernels' three output arrays become one allocation: it is the all-gather's payload as it stands (pre3_match_shard_match)
    sh->raw_stride = round_up(K1, 128);
    if (sh->raw.alloc(sizeof(double) * 3 * (size_t)sh->raw_stride) != PRE3_OK) { delete sh; return nullptr; }
    if (hipMemset(sh->raw.p, 0, sizeof(double) * 3 * (size_t)sh->raw_stride) != hipSuccess) { set_error("hipMemset failed"); delete sh; return nullptr; }
    I8Match &o = sh->out();
    o.ob.borrow(sh->raw.p); o.os.borrow((double *)sh->raw.p + sh->raw_stride); o.oa.borrow((double *)sh->raw.p + 2 * (size_t)sh->raw_stride);
    return sh;
}
int match_shard_run(void *h, void **partial_dev, int *n_doubles)
{
    MatchShard *sh = (MatchShard *)h;
    PRE3_HIP(hipSetDevice(sh->device));
    I8Match &o = sh->out();
    if (o.K2 > 0) {
        if (sh->cls == 2) PRE3_TRY(i8_run(sh->m, sh->k2_offset, 0));
        else if (sh->cls == 0) PRE3_TRY(rank_run<double>(sh->r, sh->k2_offset, 0));
        else PRE3_TRY(rank_run<float>(sh->r, sh->k2_offset, 0));
    } else { PRE3_HIP(hipMemsetAsync(o.oa.p, 0xff, sizeof(int32_t) * o.K1, 0)); }                                   // empty slice: arg = -1 everywhere
    hipLaunchKernelGGL(k_shard_pack, dim3(ceil_div(o.K1, 256)), dim3(256), 0, 0, o.K1, (const double *)o.ob.p, (const double *)o.os.p,
                       (const int32_t *)o.oa.p, (double *)sh->part.p);
    PRE3_HIP(hipGetLastError());
    PRE3_HIP(hipStreamSynchronize(0));              // the caller's collective runs on another stream
    if (partial_dev) *partial_dev = sh->part.p;
    if (n_doubles) *n_doubles = 3 * o.K1;
    return PRE3_OK;
}
int match_shard_merge(void *h, int G, const void *gathered_dev, double thresh, double *pairs_out, double *score_out, int *M_out)
{
    MatchShard *sh = (MatchShard *)h;
    PRE3_CHECK(G >= 1 && gathered_dev && pairs_out && M_out, PRE3_E_ARG, "match shard merge: bad arguments");
    PRE3_HIP(hipSetDevice(sh->device));
    const int K1 = sh->out().K1;
    hipLaunchKernelGGL(k_shard_merge, dim3(1), dim3(1024), 0, 0, G, K1, (const double *)gathered_dev, (float)thresh, (double *)sh->res.p);
    PRE3_HIP(hipGetLastError());
    // ONE copy of the whole result block [count | pairs (2 K1) | scores (K1)] (98 KB at K1 = 4096): three dependent copies cost three PCIe
    // round trips and the count is not known before the first
    sh->host_res.resize(1 + 3 * (size_t)K1);
    PRE3_HIP(hipMemcpy(sh->host_res.data(), sh->res.p, sizeof(double) * sh->host_res.size(), hipMemcpyDeviceToHost));
    const int M = (int)sh->host_res[0];
    if (M > 0) {
        memcpy(pairs_out, sh->host_res.data() + 1, sizeof(double) * 2 * M);
        if (score_out) memcpy(score_out, sh->host_res.data() + 1 + 2 * (size_t)K1, sizeof(double) * M);
    }
    *M_out = M;
    return PRE3_OK;
}
int match_shard_set_comm(void *h, void *comm)
{
    MatchShard *sh = (MatchShard *)h;
    PRE3_CHECK(comm == nullptr || comm_device(comm) == sh->device, PRE3_E_ARG, "match shard: the communicator lives on device %d, the shard on %d", comm ? comm_device(comm) : -1, sh->device);
    sh->comm = comm;
    if (comm) {
        // the all-gather's destination is sized here, not inside a match: an allocation that fails must not be a reason for this rank to stay out
        // of a collective its peers have entered
        int rank = 0, world = 1;
        comm_rank_world(comm, &rank, &world);
        PRE3_HIP(hipSetDevice(sh->device));
        if (sh->gathered_world != world) { sh->gathered_world = 0; PRE3_TRY(sh->gathered.alloc(sizeof(double) * 3 * (size_t)sh->raw_stride * world)); sh->gathered_world = world; }
    }
    return PRE3_OK;
}
// run + all-gather + merge on the shard's stream; the host waits once, on the word the merge kernel writes behind its result block
int match_shard_match(void *h, double thresh, double *pairs_out, double *score_out, int *M_out)
{
    MatchShard *sh = (MatchShard *)h;
    PRE3_CHECK(pairs_out && M_out, PRE3_E_ARG, "match shard: null outputs");
    PRE3_HIP(hipSetDevice(sh->device));
    I8Match &o = sh->out();
    const int K1 = o.K1;
    const size_t nres = 1 + 3 * (size_t)K1;
    if (!sh->st) PRE3_HIP(hipStreamCreateWithFlags(&sh->st, hipStreamNonBlocking));
    if (!sh->res_host) {
        PRE3_HIP(hipHostMalloc((void **)&sh->res_host, sizeof(double) * nres + 64, hipHostMallocMapped));
        memset(sh->res_host, 0, sizeof(double) * nres + 64);
        PRE3_HIP(hipHostGetDevicePointer((void **)&sh->res_host_dev, sh->res_host, 0));
    }
    int rank = 0, world = 1;
    if (sh->comm) comm_rank_world(sh->comm, &rank, &world);
    const size_t nraw = 3 * (size_t)sh->raw_stride;
    PRE3_CHECK(!sh->comm || sh->gathered_world == world, PRE3_E_STATE, "match shard: the gather buffer does not fit the communicator (pre3_match_shard_set_comm)");
    // A rank-local failure of the distance kernels must not keep this rank out of the all-gather its peers are entering: it enters all the same,
    // with its slice empty (arg = -1 everywhere) and the MISSING word behind the arg array set; the merge of every rank sees the word and every
    // rank fails this match with PRE3_E_COMM (as pre3_ransac_sharded does with its missing-slice word).
    int32_t *missing_dev = reinterpret_cast<int32_t *>((double *)sh->raw.p + 2 * (size_t)sh->raw_stride) + sh->raw_stride;
    int rc_local = PRE3_OK;
    if (sh->test_fail) { sh->test_fail = false; set_error("match shard: distance kernels failed (test hook)"); rc_local = PRE3_E_HIP; }
    else if (o.K2 > 0) {
        if (sh->cls == 2) rc_local = i8_run(sh->m, sh->k2_offset, sh->st);
        else if (sh->cls == 0) rc_local = rank_run<double>(sh->r, sh->k2_offset, sh->st);
        else rc_local = rank_run<float>(sh->r, sh->k2_offset, sh->st);
    } else if (hipMemsetAsync(o.oa.p, 0xff, sizeof(int32_t) * K1, sh->st) != hipSuccess) { set_error("match shard: hipMemsetAsync failed"); rc_local = PRE3_E_HIP; }
    if (rc_local != PRE3_OK) {
        (void)hipMemsetAsync(o.oa.p, 0xff, sizeof(int32_t) * K1, sh->st);
        (void)hipMemsetAsync(missing_dev, 1, sizeof(int32_t), sh->st);
        sh->missing_set = true;
    } else if (sh->missing_set) { (void)hipMemsetAsync(missing_dev, 0, sizeof(int32_t), sh->st); sh->missing_set = false; }
    const void *src = sh->raw.p;
    if (sh->comm) {
        const int rc_coll = comm_all_gather_f64(sh->comm, sh->raw.p, sh->gathered.p, nraw, sh->st);
        if (rc_local != PRE3_OK) return rc_local;
        PRE3_TRY(rc_coll);
        src = sh->gathered.p;
    } else if (rc_local != PRE3_OK) return rc_local;
    int32_t *mail_host = reinterpret_cast<int32_t *>(sh->res_host + nres), *mail_dev = reinterpret_cast<int32_t *>(sh->res_host_dev + nres);
    const int32_t seq = ++sh->seq;
    hipLaunchKernelGGL(k_shard_merge, dim3(1), dim3(1024), 0, sh->st, world, K1, (const double *)src, (float)thresh, sh->res_host_dev, mail_dev, seq, sh->raw_stride);
    PRE3_HIP(hipGetLastError());
    // the wait has a wall-clock deadline when a collective is in front of the merge (pre3_comm_set_timeout): a peer that stalls or never entered
    // ends in PRE3_E_COMM with the communicator aborted, never in a bare hipStreamSynchronize
    bool arrived = false;
    timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (long spin = 0; (sh->comm || spin < 2000000000L) && !arrived; ++spin) {
        if (__atomic_load_n(mail_host, __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
        if ((spin & 0x3ffff) == 0x3ffff) {
            if (sh->comm) {
                PRE3_TRY(comm_poll_error(sh->comm));
                timespec ts1; clock_gettime(CLOCK_MONOTONIC, &ts1);
                if ((ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6 > comm_timeout_ms(sh->comm)) return comm_give_up(sh->comm, "the all-gather of a sharded match");
            }
            const hipError_t q = hipStreamQuery(sh->st);
            if (q == hipSuccess) { arrived = __atomic_load_n(mail_host, __ATOMIC_ACQUIRE) == seq; break; }
            if (q != hipErrorNotReady) { set_error("match shard: stream failed: %s", hipGetErrorString(q)); return PRE3_E_HIP; }
        }
    }
    if (!arrived) { PRE3_TRY(stream_drain_on(sh->st, sh->comm, "a sharded match")); PRE3_CHECK(__atomic_load_n(mail_host, __ATOMIC_ACQUIRE) == seq, PRE3_E_HIP, "match shard: the merge kernel did not publish its result"); }
    PRE3_CHECK(mail_host[1] == 0, PRE3_E_COMM, "sharded match: %d rank(s) failed before the all-gather (their slices are missing): the match is void on every rank", mail_host[1]);
    const int M = (int)sh->res_host[3 * (size_t)K1];
    if (M > 0) {
        memcpy(pairs_out, sh->res_host, sizeof(double) * 2 * M);
        if (score_out) memcpy(score_out, sh->res_host + 2 * (size_t)K1, sizeof(double) * M);
    }
    *M_out = M;
    return PRE3_OK;
}
// test hook (pre3_match_shard_test_stall): a kernel on the shard's stream that spins on a word of the pinned result block until released
__global__ void k_shard_stall(volatile int32_t *flag)
{
    for (long spin = 0; spin < 6000000L; ++spin) {          // (~20 s: the kernel lets go by itself)
        if (__hip_atomic_load(const_cast<int32_t *>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
        __builtin_amdgcn_s_sleep(100);
    }
}
int match_shard_test_stall(void *h, int release)
{
    MatchShard *sh = (MatchShard *)h;
    if (release == 2) { sh->test_fail = true; return PRE3_OK; }
    PRE3_HIP(hipSetDevice(sh->device));
    const size_t nres = 1 + 3 * (size_t)sh->out().K1;
    if (!sh->st) PRE3_HIP(hipStreamCreateWithFlags(&sh->st, hipStreamNonBlocking));
    if (!sh->res_host) {
        PRE3_HIP(hipHostMalloc((void **)&sh->res_host, sizeof(double) * nres + 64, hipHostMallocMapped));
        memset(sh->res_host, 0, sizeof(double) * nres + 64);
        PRE3_HIP(hipHostGetDevicePointer((void **)&sh->res_host_dev, sh->res_host, 0));
    }
    int32_t *mail_host = reinterpret_cast<int32_t *>(sh->res_host + nres), *mail_dev = reinterpret_cast<int32_t *>(sh->res_host_dev + nres);
    if (release) { __atomic_store_n(mail_host + 3, 1, __ATOMIC_RELEASE); return PRE3_OK; }
    __atomic_store_n(mail_host + 3, 0, __ATOMIC_RELEASE);
    hipLaunchKernelGGL(k_shard_stall, dim3(1), dim3(1), 0, sh->st, (volatile int32_t *)(mail_dev + 3));
    PRE3_HIP(hipGetLastError());
    return PRE3_OK;
}
void match_shard_destroy(void *h)
{
    MatchShard *sh = (MatchShard *)h;
    if (!sh) return;
    // a collective of this shard may still be queued, or -- after a deadline -- the communicator's abort may still be inside RCCL: the stream is drained
    // (bounded) and the abort joined before the shard's buffers go
    if (sh->st) (void)stream_drain_on(sh->st, sh->comm, "pre3_match_shard_destroy");
    if (sh->comm && comm_abort_wait(sh->comm, comm_abort_ms(sh->comm)) && comm_broken(sh->comm) && sh->st) (void)stream_drain_on(sh->st, sh->comm, "pre3_match_shard_destroy");
    delete sh;
}

// matcher bench handle (inputs resident in HBM): cls 2 = uint8 descriptors on the int8 MFMA path; cls 0 / 1 = double / float descriptors on
// the route their data selects (info: [route, queries scanned in full, candidates re-evaluated] of the last run)
struct MatchBench { int cls = 2; I8Match m; RankMatch r; };
void *match_bench_create(int cls, int ND, int K1, const void *L1, int K2, const void *L2)
{
    MatchBench *b = new MatchBench();
    b->cls = cls;
    int rc = PRE3_E_ARG;
    if (cls == 2) rc = i8_prepare(b->m, ND, K1, (const uint8_t *)L1, K2, (const uint8_t *)L2, 128);
    else if (cls == 0 && rank_applies<double>(ND, K1, K2)) rc = rank_prepare<double>(b->r, ND, K1, (const double *)L1, K2, (const double *)L2);
    else if (cls == 1 && rank_applies<float>(ND, K1, K2)) rc = rank_prepare<float>(b->r, ND, K1, (const float *)L1, K2, (const float *)L2);
    else set_error("matcher bench: class %d / shape not on an MFMA path", cls);
    if (rc == PRE3_OK && cls != 2 && b->r.route == 0) { set_error("matcher bench: the data is outside the ranked path's bounds"); rc = PRE3_E_ARG; }
    if (rc != PRE3_OK) { delete b; return nullptr; }
    return b;
}
static int bench_once(MatchBench *b, bool count = false)
{
    if (b->cls == 2) return i8_run(b->m, 0, 0);
    if (count) PRE3_HIP(hipMemsetAsync((int *)b->r.fl.p + 2, 0, 2 * sizeof(int), 0));
    return b->cls == 0 ? rank_run<double>(b->r, 0, 0, count) : rank_run<float>(b->r, 0, 0, count);
}
int match_bench_run(void *h, int reps, double *ms_per)
{
    MatchBench *b = (MatchBench *)h;
    hipEvent_t e0, e1;
    PRE3_HIP(hipEventCreate(&e0)); PRE3_HIP(hipEventCreate(&e1));
    PRE3_TRY(bench_once(b, true));             // (the untimed warm-up run also counts the candidates for match_bench_info)
    PRE3_HIP(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) PRE3_TRY(bench_once(b));
    PRE3_HIP(hipEventRecord(e1, 0));
    PRE3_HIP(hipEventSynchronize(e1));
    float ms = 0; PRE3_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return PRE3_OK;
}
int match_bench_fetch(void *h, double *best, double *second, int32_t *arg)
{
    MatchBench *b = (MatchBench *)h;
    I8Match &m = b->cls == 2 ? b->m : b->r.m;
    PRE3_HIP(hipMemcpy(best, m.ob.p, sizeof(double) * m.K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(second, m.os.p, sizeof(double) * m.K1, hipMemcpyDeviceToHost));
    PRE3_HIP(hipMemcpy(arg, m.oa.p, sizeof(int32_t) * m.K1, hipMemcpyDeviceToHost));
    return PRE3_OK;
}
int match_bench_info(void *h, int32_t info[3])
{
    MatchBench *b = (MatchBench *)h;
    info[0] = b->cls == 2 ? 1 : b->r.route; info[1] = info[2] = 0;
    if (b->cls != 2) { int st[2] = { 0, 0 }; PRE3_HIP(hipMemcpy(st, (int *)b->r.fl.p + 2, sizeof(st), hipMemcpyDeviceToHost)); info[1] = st[0]; info[2] = st[1]; }
    return PRE3_OK;
}
void match_bench_destroy(void *h) { delete (MatchBench *)h; }

}  // namespace pre3
