/* pre3_test_hooks.h -- NOT part of the public interface (include/pre3.h): two hooks for tests/test_gpu_comm.py, exported from libpre3.so but inert
 * unless the environment has PRE3_TEST_HOOKS=1 (they return PRE3_E_STATE otherwise).
 *
 * pre3_test_stall(ctx, 0) / pre3_match_shard_test_stall(s, 0): park a one-thread kernel on the context's / the shard's stream that spins until it is
 * released (..., 1) -- it also lets go by itself after ~20 s -- so that a test can stand in for a peer that stalls inside a collective.
 * pre3_match_shard_test_stall(s, 2): the next match's distance kernels "fail" (a rank-local failure in front of the all-gather).
 *
 * Reserved mailbox words: word 15 of the context's pinned mailbox (pre3_ctx::mail_host; words 0..9 carry the step's counts and sequence numbers, 16..19 the
 * staging blocks') and word 3 of the shard's result-block mailbox (words 0..2: the match's sequence number, count and missing-slice word) belong to these
 * hooks and to nothing else. */
#pragma once
#include "../../include/pre3.h"
#ifdef __cplusplus
extern "C" {
#endif
PRE3_API int pre3_test_stall(pre3_ctx *ctx, int release);
PRE3_API int pre3_match_shard_test_stall(pre3_match_shard *s, int release);
#ifdef __cplusplus
}
#endif
