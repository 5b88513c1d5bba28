/*
 * defslam_hip_debug.h -- lab / measurement entry points of libdefslam_hip_lab.so (MI355X / gfx950).
 *
 * NOT part of the product ABI.  The product library (libdefslam_hip.so, include/defslam_hip.h) does not export these
 * symbols, reads no environment variables and contains neither the test hooks nor the A/B solver variants.  The lab
 * library is the same source built with -DDSH_LAB (make -C defslam_amd/csrc lab) and exports the product ABI plus the
 * functions below; it is what the parity tests of the normal equations, the profiling scripts under tools/ and the
 * isolated-assembly leg of bench.py load.
 */
#ifndef DEFSLAM_HIP_DEBUG_H
#define DEFSLAM_HIP_DEBUG_H

#include "defslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only this ABI is exported */
#endif

/* Solver A/B switches, applied by the next dsh_sft_batch_upload of this context:
 *   "waves"    0 = automatic (default), 4 or 8 wavefronts per problem of the persistent kernel
 *   "rounds"   1 (default) = from two problems per CU upwards the batch runs as rounds of LIN / FACTOR / TRIAL launches with one wavefront per
 *              factorisation (sft_batch.h, sft_wave.h); 0 = the persistent four-wavefront kernel for such batches as well
 *   "streams"  0 (default) = one, 1..4 = that many sub-batches of the throughput shape, each on a stream of its own (the tail of one
 *              sub-batch's FACTOR launch overlaps with the next launches of the others)
 *   "dataflow" 1 = barrier-free factor steps (default), 0 = the barrier version (only compiled into the lab library)
 *   "wide_off" 1 = half-bandwidths 128 < kd <= 256 use the row-major band solver instead of the wide tile solver
 *   "speculate"  0 = automatic, 1 = off (the one-workgroup persistent kernel), 2..4 = that many workgroups per problem try
 *                consecutive dampings of an iteration side by side (latency mode; results are bit-identical either way)
 *   "split"    2 (default) = in latency mode every band that is long enough (at least eight times its bandwidth) is factored from both
 *              ends by two workgroups with a separator of one bandwidth in between; a narrow band (kd <= 128) then runs on the
 *              left-looking wide-tile code as well (C2: 4.1 ms per frame against 4.5 on the register-window solver of one workgroup)
 *              while the launch holds at most num_cus / 20 problems (beyond that the register-window solver wins),
 *              1 = wide bands (128 < kd <= 256) only, 0 = one factorisation of the whole band (the same Cholesky in another
 *              elimination order: trajectories agree, numbers to rounding)
 *   "helpers"  -1 (default) = automatic: two or three helper workgroups per part of a two-sided factorisation of a wide band (at least 12
 *              sub-diagonal tiles) when the device holds them all (B x lanes x 2 x (1 + helpers) <= CUs); 0..3 = that many -- the results are
 *              the same bits for every value (sft_wide.h: the owner forms a far sum itself whenever a helper's is not there)
 *   "helpers_wbt"  12 (default) = the half-bandwidth in tiles from which parts get helpers (C2, 8 tiles, with helpers: 4.0 against 3.8 ms per frame)
 *   "owner_waves"  8 (default) | 16 = wavefronts of a FACTOR workgroup when there are helpers: 16 runs sft_part_factor_kernel -- one live row
 *              per wave, four wavefronts per SIMD; the same bits, 31.7 against 24.5 ms per C5 frame: a column is bound by what its waves issue
 *   "tail"     -1 (default) = automatic: the last problems of a step of the throughput shape -- from four per CU downwards, at most three quarters of
 *              the batch, a batch of two per CU or less as a whole -- are run to their end by sftb_tail_kernel (one workgroup per problem);
 *              1..8 = exactly that many problems per CU, 0 = rounds of phase kernels to the end
 *              (the eight-wavefront solver of the tail kernel rounds differently from the one-wavefront solver: same trajectories, x to 5e-13) */
int dsh_lab_set_option(dsh_ctx* ctx, const char* name, int value);
/* How problem b of the uploaded batch is solved: out[8] = {two-sided factorisation on?, first separator scalar c0, separator
 * scalars s, scalars of part 1 incl. padding, its padding, workgroups (lanes) per problem, tile mode, wavefronts per workgroup}. */
int dsh_lab_sft_solver_info(dsh_ctx* ctx, int b, int32_t* out8);

/* `launches` back-to-back runs of the uploaded batch bracketed by HIP events recorded on dsh_stream; elapsed device
 * milliseconds between the two events. */
int dsh_lab_sft_run_timed(dsh_ctx* ctx, int launches, double* total_ms);
/* Measurement aid for the Jacobian-assembly roofline: `launches` back-to-back launches in which every problem
 * of the batch does ONE linearisation (residuals + Jacobian records, DefOptimizer.cc:293-507 / g2o linearizeSystem) and one
 * normal-equation assembly at its uploaded initial state and stops; elapsed device milliseconds between two HIP events.
 * Needs a batch that has run once (H keeps the zero pattern of that run); invalidates that run's results. */
int dsh_lab_sft_assemble_timed(dsh_ctx* ctx, int launches, double* total_ms);
/* A/B of the one-wavefront factorisation (sft_wave.h) against the four-wavefront solver: the normal equations of every problem of the
 * batch at its uploaded state (like dsh_lab_sft_assemble_timed) are solved with lambda = rel * 1e-5 * max |diag H| by both; x_ref / x_new
 * (may be NULL) receive the two solutions problem after problem, Dnp + 6 doubles each (node unknowns padded to a multiple of 32, then the
 * camera), ok2[2 b + which] the "all pivots positive" flags, ms2[which] the device time of one launch (average over `launches`);
 * only = 0: both, 1: the four-wavefront solver alone, 2: the one-wavefront solver alone (with the lambda the last reference run left). */
int dsh_lab_sft_wave_check(dsh_ctx* ctx, double rel, int launches, int only, double* x_ref, double* x_new, int32_t* ok2, double* ms2);
/* One run of the uploaded batch in the throughput shape (rounds of phase kernels, sft_batch.h) with a HIP event in front of and behind
 * every launch: ms7[0..4] = total device milliseconds of the INIT, LIN, FACTOR, TRIAL and tail-kernel launches, ms7[5] = factorisations the FACTOR
 * launches performed, ms7[6] = linearisations the LIN launches performed (the last problems' are the tail kernel's); *rounds (may be NULL) = rounds of phase kernels launched.
 * What bench.py's roofline objects are computed from (sftb_factor_kernel: FP64; sftb_lin_kernel: the Jacobian assembly). */
int dsh_lab_sft_rounds_timed(dsh_ctx* ctx, double* ms7, int32_t* rounds);
/* n doubles of a workspace array of problem b: what = 0 L tiles, 1 inverse diagonal tiles, 2 border rows of L, 3 compact H blocks, 4 border rows
 * of H, 5 x, 6 corner of H, 7 debug slots, 8 / 9 the sync words of part 0 / 1 of a two-sided factorisation with helper workgroups (int32: [0] the
 * owner's progress, [1..4] owner statistics, [5..12] helper statistics; fails when the problem has none) (tuning aid; no bounds check beyond n > 0). */
int dsh_lab_sft_dump(dsh_ctx* ctx, int b, int what, int64_t n, double* out);
/* Per-phase device time of problem b in the last run, milliseconds (constant 100 MHz counter read by one lane):
 * out8[1] residuals, [2] normal-equation assembly, [3] H->L copy, [4] panel factorisation, [5] trailing update,
 * [6] back substitution, [7] state update + LM control.  DSH_ERR_STATE unless built with EXTRA=-DSFT_PHASE_TIMERS. */
int dsh_lab_sft_phase_ms(dsh_ctx* ctx, int b, double* out8);
/* Shader-clock stamps of factorisation step 40 of problem b: 8 per wavefront (out64[8 w + e]).  DSH_ERR_STATE unless built
 * with EXTRA=-DSFT_STEP_TRACE. */
int dsh_lab_sft_step_trace(dsh_ctx* ctx, int b, double* out64);
/* Run only "residuals + Jacobians + normal equations" once at the uploaded state of problem b and return the dense system
 * in the reference's index order (camera first, then active nodes ascending; column-major D x D) plus b and the robust
 * chi2.  H / bvec may be NULL.  Invalidates the results of the last run. */
int dsh_lab_sft_system(dsh_ctx* ctx, int b, int32_t D, double* H, double* bvec, double* chi2);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DEFSLAM_HIP_DEBUG_H */
