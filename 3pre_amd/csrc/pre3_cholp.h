// pre3_cholp.h -- the persistent factorisation + triangular solve (pre3_cholp.hip): host entry points
#pragma once
#include "pre3_internal.h"

namespace pre3 {

constexpr int CP_NTH = 768;          // crit: ten chain waves + two side waves; rows and strips use the first four waves
// Panels of 64 rows the one-launch form takes.  The strips keep the last 11 blocks of W as bf16 planes in LDS (a ring) and re-read older ones
// from the planes they wrote, so the form WORKS for any count up to the 64 flag slots -- config 5's 40 panels included (round 4; correct, tested
// at 15 and 16 panels) -- but it only PAYS while the strips' per-panel sum stays in the shadow of the chain: at 40 panels (N = 2000) the launch took
// 1.69-1.82 ms against 1.73 ms for the launch-per-panel form (rocprofv3, gpurun_out/r4_n2000b): the strips' block loop is latency-bound (~2 us per
// 64x32x64 block: its operands are requested one block ahead), 378 strips need two dispatch rounds on 256 CUs (the second one starts at 0.92 ms),
// and the 38 row workgroups' tile traffic delays crit's next tiles from panel 10 on.  So updates beyond 16 panels keep the launch-per-panel form.
constexpr int CP_MAX_NRB = 16;

size_t cholp_flag_bytes(int n_strips);
// the down-date consumers' group table for nb 64-column blocks (device records, the tiles in group order, first tile of each group)
void dd_build_groups(int nb, std::vector<int32_t> &rec, std::vector<int2> &tiles64, std::vector<int> &tile_off);
bool cholp_usable(const pre3_ctx *c, int nrb_max);
void cholp_context_count(int device, int delta);      // a context with the persistent form's buffers was created (+1) / destroyed (-1)
// S (c->Smat, stride nrb * 64) and [HP | nu] (c->W) in place -> L and W = L^-1 [HP | nu], W's bf16 planes (c->Wp) included.
// nrb < 0: the number of rows is read on the device (stats[4] measurements, as k_gather_li does); nrb_max bounds grid and LDS.
void cholp_timing_rows(pre3_ctx *c, int r);             // pre3_kernel_timing: the rows of a bracketed speculative launch have reached the host
// tail_req (pre3_step's LI update, row count on the device): the launch also runs the rescue stage and the HI update of up to 32 landmarks
// (mono_slam.m:184-187) when cholp_tail_usable and every tile group of P is in the launch; c->tail_launched says whether it does
struct CholpTailReq { double chi2; int32_t seq; };
bool cholp_tail_usable(const pre3_ctx *c, int nrb_max);
int launch_cholp(pre3_ctx *c, int nrb, int nrb_max, int rows = -1, int which_prior = -1 /* >= 0: the launch may also compute x_k_k from that prior */,
                 const CholpTailReq *tail_req = nullptr);

}  // namespace pre3
