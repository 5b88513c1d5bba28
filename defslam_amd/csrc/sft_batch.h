// The throughput shape of the batched solver: the Levenberg-Marquardt loop of B independent problems as ROUNDS of phase kernels,
//   LIN     residuals + Jacobian records + normal equations        (problems that start an iteration;   512 threads)
//   FACTOR  (H + lambda I) x = b, one wavefront per problem         (every problem that is still running;  64 threads, sft_wave.h)
//   TRIAL   state update, chi2 of the trial, LM controller          (the same problems;                   256 threads)
// instead of one persistent kernel per problem (sft_lm_kernel).  Why: the factorisation wants a wave that owns a whole SIMD (512
// registers, one wave per SIMD), the assembly wants many small waves that cover each other's gather latency -- one launch shape cannot
// give both.  Between the phases a problem's state lives where it already lived (H as compact blocks, x, the node positions: HBM);
// what the persistent kernel kept in LDS between phases -- the controller -- is the SftRun record.  The kernel boundary is the only
// synchronisation.  Included by sft_kernels.hip inside its anonymous namespace; the arithmetic of LIN and TRIAL is the code of the
// persistent kernel (eval_edges, assemble, pose_oplus, classify), the controller follows sft_lm_kernel line by line
// (optimization_algorithm_levenberg.cpp:61-164, sparse_optimizer.cpp:403-475, DefOptimizer.cc:513).
#pragma once

// Wavefronts of a TRIAL workgroup -- the same as LIN's: the block-wide chi2 sums associate per thread and wavefront, and chi2 of the state LIN
// linearised at and chi2 of a trial that did not move must be the SAME number (gain ratio exactly 0, the step rejected, as in the reference);
// costs 2.4 ms per step of 16384 problems against 4 (21.7 / 19.3 ms, A/B interleaved on one box) -- paid for the property above, and
// won back by the LDS copy of the node positions in the trial's residual pass (19.5 ms; 17.2 with the trial state in LDS only).
#define SFTB_NW 8
#ifndef SFTB_LIN_NW
// Wavefronts of a LIN workgroup: 8 = one workgroup per CU with the CU's whole LDS, so every record class the assembly gathers (node matrices and
// stretching records too: placement class 2, sft_kernels.hip AsmRec) is an LDS read.  Measured on 16384 C2 problems (tools/diag/assembly_shapes.py):
// 7.22 ms per pass against 8.13-8.18 with 4 wavefronts and two workgroups per CU (class 1).
#define SFTB_LIN_NW 8
#endif
#ifndef SFTB_LIN_WAVES
#define SFTB_LIN_WAVES 2   // waves per SIMD the LIN kernel is compiled for (A/B: tools/ab_build.sh NAME "-DSFTB_LIN_WAVES=3")
#endif
#ifndef SFTB_TRIAL_WAVES
#define SFTB_TRIAL_WAVES 2
#endif

__device__ __forceinline__ void sftb_ctl_lds(char* smem, Ctl*& ctl, double*& red, double*& out, double*& panel) {
  ctl = reinterpret_cast<Ctl*>(smem);
  red = reinterpret_cast<double*>(smem + 512);
  out = red + 16 * 27 + 5;
  panel = out + 32;
}

// counters (zeroed by the host in front of INIT): [0] finished problems, [1] FACTOR's work counter, [2] entries of the LIN list, [3] LIN's work
// counter, [5] the tail kernel's work counter, [7] rounds that ran, [8] factorisations by FACTOR launches, [9] linearisations by LIN launches, [6] tail mode: few enough problems are left (B - counters[0] <= tail_below, decided by the
// first kernel of a round from the count the previous round left, so the switch does not depend on how the host groups its launches):
// the phase kernels leave at their first instruction, the tail kernel runs the rest
__global__ __launch_bounds__(64 * SFTB_NW, 2) void sftb_init_kernel(const SftDev* __restrict__ probs, SftRun* __restrict__ runs, int* __restrict__ counters, int* __restrict__ lin_list, int tail_below) {
  const SftDev& P = probs[blockIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0 && (int)gridDim.x <= tail_below) counters[6] = 1;
  init_state<64 * SFTB_NW>(P);
  if (threadIdx.x == 0) {
    SftRun& R = runs[blockIdx.x];
    R.lambda = -1.0; R.ni = 2.0; R.chi_cur = 0.0; R.chi_ini = 0.0; R.rho = 0.0; R.lambda_start = 0.0;
    R.it = 0; R.qmax = 0; R.nbad = 0; R.accepted = 0; R.all_ok = 1; R.iters = 0; R.trials = 0; R.fact_ok = 1;
    R.state = P.max_iters > 0 ? SFTB_LIN : SFTB_FINISH;
    if (P.max_iters > 0) lin_list[atomicAdd(&counters[2], 1)] = blockIdx.x;   // (the host zeroes the counters in front of this launch)
  }
}

// One linearisation (LIN, and the tail kernel below): residuals, records, normal equations; the first iteration also fixes the initial damping
template <int NW>
__device__ __forceinline__ void sftb_lin_problem(const SftDev& P, SftRun& R, Ctl* ctl, double* red, double* out, double* panel) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  auto nothing = [] {};
  const double chi0 = linearise<NW, decltype(nothing), true>(P, ctl, red, out, panel, nothing);
  double lambda = R.lambda;
  if (R.it == 0) {
    double mx = 0.0;
    for (int r = tid; r < P.Dn; r += NT) mx = fmax(mx, fabs(h_diag(P, r)));
    if (tid < 6) mx = fmax(mx, fabs(P.Hcorner[tid * 8]));
    mx = block_max(mx, red);
    lambda = 1e-5 * mx;
  }
  if (tid == 0) {
    if (R.it == 0) { R.lambda = lambda; R.ni = 2.0; R.nbad = 0; }
    R.chi_cur = chi0; R.chi_ini = chi0; R.qmax = 0; R.rho = 0.0; R.accepted = 0; R.all_ok = 1; R.lambda_start = lambda;
    R.state = SFTB_TRIAL;
  }
}

// LIN: a problem that starts an outer iteration is linearised; the first iteration also fixes the initial damping (tau = 1e-5).
// Persistent workgroups pull the problems from the list the previous TRIAL (or INIT) launch appended them to (counters[2] entries, work counter
// counters[3]; FACTOR resets both): a launch over all B problems spent 0.1-0.2 ms per round on workgroups that found nothing to do -- each of
// them holds a CU's LDS while it finds out.  The order of the list varies from run to run; the problems are independent, the results do not.
__global__ __launch_bounds__(64 * SFTB_LIN_NW, SFTB_LIN_WAVES) void sftb_lin_kernel(const SftDev* __restrict__ probs, SftRun* __restrict__ runs, int* __restrict__ counters,
                                                                                  const int* __restrict__ lin_list, int B, int tail_below) {
  constexpr int NW = SFTB_LIN_NW, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int next_b;
  Ctl* ctl; double *red, *out, *panel;
  sftb_ctl_lds(smem, ctl, red, out, panel);
  const int tid = threadIdx.x;
  // the first kernel of a round: few enough problems left for the tail kernel?  (nobody changes counters[0] during this launch: every workgroup
  // sees the same count; workgroup 0 leaves the verdict for FACTOR and TRIAL)
  if (counters[6] || B - counters[0] <= tail_below) {
    if (blockIdx.x == 0 && tid == 0) counters[6] = 1;
    return;
  }
  if (blockIdx.x == 0 && tid == 0) counters[7]++;   // rounds that ran (what the host enqueues in one go next time)
  // (the index of the problem after this one is fetched while this one is linearised: the atomic and the list entry are a round trip to memory
  // each, on every workgroup's critical path otherwise)
  auto fetch = [&]() { const int i = atomicAdd(&counters[3], 1); return i < counters[2] ? lin_list[i] : -1; };
  int nxt = -1;
  if (tid == 0) nxt = fetch();
  while (true) {
    if (tid == 0) next_b = nxt;   // (the barriers inside the previous linearisation lie between this and the last read of next_b)
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(next_b);
    if (b < 0) break;
    if (tid == 0) { nxt = fetch(); atomicAdd(&counters[9], 1); }   // in flight until the top of the loop; [9]: linearisations by LIN launches
    sftb_lin_problem<NW>(probs[b], runs[b], ctl, red, out, panel);
  }
}

// FACTOR: persistent wavefronts (one per SIMD) pull running problems from a counter; the back substitution of a wave's previous problem
// rides in the factor steps of its next one (sft_wave.h), the last one is solved right away.
__global__ __launch_bounds__(64, 1) void sftb_factor_kernel(const SftDev* __restrict__ probs, SftRun* __restrict__ runs, int* __restrict__ counters, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_double* lds = to_lds(reinterpret_cast<double*>(smem));
  const int lane = threadIdx.x;
  if (counters[6]) return;                                        // tail mode
  for (int i = lane; i < WV_LDS_DOUBLES; i += 64) lds[i] = 0.0;   // (the landing buffer is multiplied by ring zeros before its first fill)
  if (blockIdx.x == 0 && lane == 0) { counters[2] = 0; counters[3] = 0; }   // the LIN list of this round is consumed; TRIAL appends the next one
  WvPrev Q;
  Q.Lg = nullptr; Q.Linv = nullptr; Q.x = nullptr; Q.nT = 0; Q.active = 0; Q.xb = 0.0;
  int n_done = 0;
  while (true) {
    int b = 0;
    if (lane == 0) b = atomicAdd(&counters[1], 1);
    b = __builtin_amdgcn_readfirstlane(b);
    if (b >= B) break;
    if (runs[b].state != SFTB_TRIAL) continue;
    const SftDev& P = probs[b];
    const double lambda = runs[b].lambda;
    double xcam;
    const int ok = wv_factor(P, lambda, lambda, lds, Q, xcam);   // (Q's back substitution is complete when this returns)
    if (lane == 0) runs[b].fact_ok = ok;
    Q = wv_prev_of(P, ok, xcam, lane);
    n_done++;
  }
  // (the lane index formed afresh: as a value that lives from the top of the kernel to this point the compiler parked it -- and two offsets derived
  // from it -- in accumulator registers across the factorisations, i.e. on a window tile; tools/wave_audit.py rule 1)
  int lane_now;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_now));
  if (Q.active) wv_backsub_now(Q, lane_now);
  if (lane_now == 0 && n_done) atomicAdd(&counters[8], n_done);   // factorisations of this kernel over the step (the tail kernel's are not among them)
}

// One damping trial behind its factorisation (TRIAL, and the tail kernel below): push, x applied, scale, chi2 at the trial state, the
// controller's verdict; pop on rejection; at the end of an iteration the stop rules; at the end of the problem the classification.
// Returns 0: the same H is factored again with the next damping; 1: the iteration is over, the next one starts with a linearisation; 2: done.
// lin_list: where a problem that starts a new iteration is appended for the next LIN launch (nullptr: the caller linearises itself).
template <int NW>
__device__ __forceinline__ int sftb_trial_problem(const SftDev& P, SftRun& R, Ctl* ctl, double* red, double* out, double* panel, int* counters, int* lin_list, int b) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  const int Dn = P.Dn;
  const int Dnp = ((Dn + NB - 1) / NB) * NB;
  const int ok = R.fact_ok;
  const double lam = R.lambda;
  // push + update (sparse_optimizer.cpp:477-491); like g2o, x keeps its previous content when the factorisation failed.  Where the node
  // positions are staged in LDS for the residual pass (placement class >= 1) the trial state of the nodes exists THERE only: the staging adds
  // the step, an accepted trial writes the LDS copy back, a rejected one leaves memory as it was -- no backup, no restore (3 x 12 KB of
  // traffic per trial at C2); same additions, same bits.
  const bool in_lds = P.lds_class >= 1;
  if (!in_lds)
    for (int i = tid; i < 3 * P.n; i += NT) {
      const double v = P.xyz[i];
      P.xyz_bak[i] = v;
      const int a = P.act[i / 3];
      if (a >= 0) P.xyz[i] = v + P.x[3 * a + (i % 3)];
    }
  if (tid < 7) R.pose_bak[tid] = P.pose[tid];
  __syncthreads();
  if (tid == 0) pose_oplus(P.pose, P.x + Dnp);
  double sc = 0.0;
  for (int r = tid; r < Dn; r += NT) { const double xv = P.x[r]; sc += xv * (lam * xv + P.Hbord[(size_t)6 * Dnp + r]); }
  if (tid < 6) { const double xv = P.x[Dnp + tid]; sc += xv * (lam * xv + P.Hcorner[42 + tid]); }
  __syncthreads();
  block_sum<1>(&sc, red, out);
  const double scale = out[0];
  __syncthreads();
  // (the positions every residual gathers three to seven of: staged in LDS where the problem's placement class says they fit -- the launch
  // sizes the LDS for them; only the position array of class 1 is touched by a pass without Jacobians)
  const auto staged = asm_records<NW, 1>(P, panel);
  const double chi_new = in_lds ? eval_edges<false, 1>(P, ctl, red, out, staged, P.x)
                                : eval_edges<false, 0>(P, ctl, red, out, asm_records<NW, 0>(P, panel));
  if (tid == 0) {
    const double tempChi = ok ? chi_new : DBL_MAX;
    double rho = (R.chi_cur - tempChi);
    rho /= (scale + 1e-3);
    R.rho = rho;
    if (rho > 0 && isfinite(tempChi)) {
      double alpha = 1. - pow((2 * rho - 1), 3);
      alpha = fmin(alpha, 2. / 3.);
      const double sf = fmax(1. / 3., alpha);
      R.lambda = lam * sf; R.ni = 2.0; R.chi_cur = tempChi; R.accepted = 1;
      ctl->stop = 0;
    } else {
      R.lambda = lam * R.ni; R.ni *= 2.0;
      ctl->stop = 1;
    }
    R.qmax++;
    R.all_ok &= ok;
    ctl->qmax = R.qmax;
    ctl->rho = rho;
  }
  __syncthreads();
  if (ctl->stop) {  // pop
    if (!in_lds)
      for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = P.xyz_bak[i];
    if (tid < 7) P.pose[tid] = R.pose_bak[tid];
  } else if (in_lds) {   // accepted: the trial state becomes the state
    for (int i = tid; i < 3 * P.n; i += NT) P.xyz[i] = staged.xyz_l[i];
  }
  const bool again = (ctl->rho < 0) && (ctl->qmax < 10);
  if (again) return 0;   // the next round factors the same H with the new damping
  // ---- the outer iteration is over
  __syncthreads();
  if (tid == 0) {
    const int qmax = R.qmax;
    R.trials += qmax;
    R.iters++;
    if (P.trace) {
      double* t = P.trace + R.it * 8;
      t[0] = R.chi_ini; t[1] = R.lambda_start; t[2] = qmax; t[3] = R.chi_cur; t[4] = R.lambda; t[5] = R.rho; t[6] = R.accepted; t[7] = R.all_ok;
    }
    if (!R.all_ok) P.res->status |= 1;
    bool term = (qmax == 10) || (R.rho == 0);
    if (!term) {
      if ((R.chi_ini - R.chi_cur) * 1e3 < R.chi_ini) R.nbad++; else R.nbad = 0;
      term = R.nbad >= 3;
    }
    R.it++;
    if (R.it >= P.max_iters) term = true;
    ctl->nbad = term ? 1 : 0;
    ctl->it = R.iters;
    ctl->accepted = R.trials;
    R.state = term ? SFTB_DONE : SFTB_LIN;
    if (!term && lin_list) lin_list[atomicAdd(&counters[2], 1)] = b;
  }
  __syncthreads();
  if (ctl->nbad) {
    const int iters = ctl->it, trials = ctl->accepted;
    __syncthreads();
    classify<NT>(P, ctl, panel, iters, trials);
    if (tid == 0) atomicAdd(&counters[0], 1);
    return 2;
  }
  return 1;
}

// TRIAL: one workgroup per problem of the batch (a finished problem's leaves at its first instruction).
__global__ __launch_bounds__(64 * SFTB_NW, SFTB_TRIAL_WAVES) void sftb_trial_kernel(const SftDev* __restrict__ probs, SftRun* __restrict__ runs, int* __restrict__ counters, int* __restrict__ lin_list) {
  constexpr int NW = SFTB_NW, NT = 64 * NW;
  SftRun& R = runs[blockIdx.x];
  const int st = R.state;
  if (counters[6]) return;                                    // tail mode
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[1] = 0;   // the work counter of the FACTOR launch in front of this one: ready for the next round
  if (st != SFTB_TRIAL && st != SFTB_FINISH) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SftDev& P = probs[blockIdx.x];
  Ctl* ctl; double *red, *out, *panel;
  sftb_ctl_lds(smem, ctl, red, out, panel);
  const int tid = threadIdx.x;
  if (st == SFTB_FINISH) {   // max_iters == 0: nothing but the classification of the initial state; no error was ever computed (chi2 = 0, like the oracle)
    classify<NT>(P, ctl, panel, 0, 0);
    if (tid == 0) { R.state = SFTB_DONE; atomicAdd(&counters[0], 1); }
    return;
  }
  (void)sftb_trial_problem<NW>(P, R, ctl, red, out, panel, counters, lin_list, blockIdx.x);
}

// (Out of line, the LDS workspace handed over as offsets: with the generic pointers of the caller as arguments -- inlined or not -- the call of
// the factorisation from the tail kernel sends hipcc 7.2 into "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base".)
__device__ __noinline__ void sftb_solve8(const SftDev& P, unsigned ctl_off, unsigned ws_off, double lambda) {
  typedef __attribute__((address_space(3))) char lds_char;
  Ctl* ctl = reinterpret_cast<Ctl*>((char*)(lds_char*)(size_t)__builtin_amdgcn_readfirstlane(ctl_off));
  double* ws = reinterpret_cast<double*>((char*)(lds_char*)(size_t)__builtin_amdgcn_readfirstlane(ws_off));
  factor_tiles_df8<1>(P, ctl, ws, lambda, true);
  backsub_tiles<8>(P, ctl, ws);
}

// TAIL: the last rounds of a step carry a handful of problems -- fewer than there are SIMDs, so a round costs one whole one-wavefront
// factorisation (1 ms) for next to nothing.  Once few enough problems are left, every one of them gets a workgroup of its own that runs it to
// the end from its SftRun record: linearisations and trials are the code of LIN and TRIAL (same bits), the factorisation is the eight-wavefront
// register-window solver of the persistent kernel (0.35 ms per trial; its x agrees with the one-wavefront solver's to 5e-13).  Persistent
// workgroups pull problem indices from counters[5].
__global__ __launch_bounds__(64 * SFTB_NW, SFT_WAVES_PER_EU) void sftb_tail_kernel(const SftDev* __restrict__ probs, SftRun* __restrict__ runs, int* __restrict__ counters, int B, int tail_below) {
  constexpr int NW = SFTB_NW, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctl* ctl; double *red, *out, *panel;
  sftb_ctl_lds(smem, ctl, red, out, panel);
  const int tid = threadIdx.x;
  // The same test the LIN kernel of the next round would make, on the count the last TRIAL launch left: few enough problems -> this launch takes
  // them (and says so for launches enqueued behind it); otherwise the rounds go on.  While the test fails nobody changes counters[0] (every
  // workgroup of this launch leaves); once it holds the count only grows -- all workgroups decide alike.  (Until r06 only LIN raised the flag:
  // the tail launch behind exactly as many rounds as the previous step had needed left at once, and the host paid a read-back, two empty rounds
  // and a second tail launch per step.)
  if (!counters[6]) {
    if (B - counters[0] > tail_below) return;
    if (blockIdx.x == 0 && tid == 0) counters[6] = 1;
  }
  while (true) {
    __syncthreads();
    if (tid == 0) ctl->it = atomicAdd(&counters[5], 1);
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(ctl->it);
    __syncthreads();
    if (b >= B) break;
    SftRun& R = runs[b];
    const SftDev& P = probs[b];
    int st = R.state;
    if (st == SFTB_FINISH) {
      classify<NT>(P, ctl, panel, 0, 0);
      if (tid == 0) { R.state = SFTB_DONE; atomicAdd(&counters[0], 1); }
      continue;
    }
    while (st == SFTB_LIN || st == SFTB_TRIAL) {
      if (st == SFTB_LIN) {
        sftb_lin_problem<NW>(P, R, ctl, red, out, panel);
        __syncthreads();
      }
      if (tid == 0) ctl->lambda = R.lambda;
      __syncthreads();
      sftb_solve8(P, (unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)ctl, (unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)panel, R.lambda);
      if (tid == 0) R.fact_ok = ctl->fact_ok;
      __syncthreads();
      const int r = sftb_trial_problem<NW>(P, R, ctl, red, out, panel, counters, nullptr, b);
      __syncthreads();
      st = r == 0 ? SFTB_TRIAL : (r == 1 ? SFTB_LIN : SFTB_DONE);
    }
  }
}
