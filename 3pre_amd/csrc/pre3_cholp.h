// pre3_cholp.h -- the persistent factorisation + triangular solve (pre3_cholp.hip): host entry points
#pragma once
#include "pre3_internal.h"

namespace pre3 {

constexpr int CP_NTH = 768;          // crit: ten chain waves + two side waves; rows and strips use the first four waves
constexpr int CP_MAX_NRB = 13;       // panels of 64 rows: the strips keep nrb - 1 blocks of W as bf16 planes in LDS (12 KB each) + 12 KB of scratch

size_t cholp_flag_bytes(int n_strips);
// the down-date consumers' group table for nb 64-column blocks (device records, the tiles in group order, first tile of each group)
void dd_build_groups(int nb, std::vector<int32_t> &rec, std::vector<int2> &tiles64, std::vector<int> &tile_off);
bool cholp_usable(const pre3_ctx *c, int nrb_max);
void cholp_context_count(int device, int delta);      // a context with the persistent form's buffers was created (+1) / destroyed (-1)
// S (c->Smat, stride nrb * 64) and [HP | nu] (c->W) in place -> L and W = L^-1 [HP | nu], W's bf16 planes (c->Wp) included.
// nrb < 0: the number of rows is read on the device (stats[4] measurements, as k_gather_li does); nrb_max bounds grid and LDS.
void cholp_timing_rows(pre3_ctx *c, int r);             // pre3_kernel_timing: the rows of a bracketed speculative launch have reached the host
int launch_cholp(pre3_ctx *c, int nrb, int nrb_max, int rows = -1, int which_prior = -1 /* >= 0: the launch may also compute x_k_k from that prior */);

}  // namespace pre3
