/*
 * defslam_hip.h -- C ABI of libdefslam_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the two DefSLAM hot paths:
 *   A. defSLAM::Optimizer::DefPoseOptimization(Frame*, Map*, RegLap, RegInex, RegTemp, layers)
 *        -- Modules/Tracking/DefOptimizer.h:51-53, DefOptimizer.cc:251-578 (g2o LM + dense LDLT)
 *   B. defSLAM::NormalEstimator::ObtainK1K2()       -- Modules/Mapping/NormalEstimator.h:46-53
 *      BBS::eval / Warps::Warp estimates            -- Thirdparty/BBS/bbs.h:52-66
 *
 * Conventions: plain pointers + sizes, caller-owned host buffers, every entry point
 * returns an int status (DSH_OK == 0); nothing throws or aborts across the ABI.  A context
 * is bound to one GPU and is not thread-safe (one context per host thread / per GPU),
 * mirroring the reference where each optimiser call runs on exactly one thread
 * (DefOptimizer.cc:287 holds MapPoint::mGlobalMutex for its whole body).
 * All floating point is FP64 unless a parameter says float (the reference's float32
 * boundaries: cv::Mat pose, keypoints, DefMapPoint world positions).
 */
#ifndef DEFSLAM_HIP_H
#define DEFSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only this ABI is exported */
#endif

#define DSH_OK 0
#define DSH_ERR_ARG 1        /* bad argument / size */
#define DSH_ERR_HIP 2        /* HIP runtime failure (see dsh_last_error) */
#define DSH_ERR_STATE 3      /* call sequence violated (no template, no batch ...) */
#define DSH_ERR_NO_DEVICE 4  /* no usable gfx950 device */

#define DSH_TRACE_STRIDE 8   /* doubles per outer LM iteration in the trace buffer:
                                chi2_start, lambda_start, trials, chi2_end, lambda_end, rho, accepted, factor_ok */
#define DSH_MAX_ITERS 64

typedef struct dsh_ctx dsh_ctx;

/* ---- context ----------------------------------------------------------------------------- */
/* device >= 0: bind to that GPU.  device == -1: host-only context (template constants, embedding and
 * problem packing work; every entry point that needs the GPU returns DSH_ERR_NO_DEVICE -- no CPU fallback). */
int dsh_create(dsh_ctx** out, int device);
int dsh_destroy(dsh_ctx* ctx);
const char* dsh_last_error(const dsh_ctx* ctx);
/* HIP stream handle (hipStream_t) the context launches on; lets a caller time with its own events. */
void* dsh_stream(dsh_ctx* ctx);
int dsh_synchronize(dsh_ctx* ctx);

/* ---- template (replaces the numbers produced by Modules/Template) ------- */
/* Derive every constant from vertices + facets the way the reference does:
 * edges/rest lengths (Facet.cc:32-56, Edge.cc:29-59), 1-ring (Node.cc:114-129), Laplacian
 * weights / boundary flags / initial mean curvature (LaplacianMesh.cc:53-162), median edge
 * (Template.cc:158-175).  Ordering: nodes by index, edges by creation order. */
int dsh_template_build(dsh_ctx* ctx, int n, const double* xyz0 /* n*3 */, int F, const int32_t* facets /* F*3 */);
/* Or hand the constants over directly (what a host shim holding the reference's Template would do). */
int dsh_template_set(dsh_ctx* ctx, int n, const double* xyz0, const uint8_t* boundary,
                     const int32_t* nbr_rowptr /* n+1 */, const int32_t* nbr_col, const double* nbr_w,
                     const double* k0 /* n */, int E, const int32_t* edge_nodes /* E*2 */, const double* edge_L0 /* E */,
                     double median_L);
/* Read back the constants of the current template (any pointer may be NULL; sizes via dsh_template_dims). */
int dsh_template_dims(const dsh_ctx* ctx, int32_t* n, int32_t* E, int32_t* nnz_nbr);
int dsh_template_get(const dsh_ctx* ctx, uint8_t* boundary, int32_t* nbr_rowptr, int32_t* nbr_col, double* nbr_w,
                     double* k0, int32_t* edge_nodes, double* edge_L0, double* median_L);
/* Barycentric embedding of P float32 points (TriangularMesh.cc:133-236): facet id (-1: none),
 * facet node ids ascending, barycentrics (float32 arithmetic as in the reference). */
int dsh_template_embed(const dsh_ctx* ctx, int P, const float* pts /* P*3 */, int32_t* facet_id, int32_t* nodes /* P*3 */,
                       float* bary /* P*3 */);

/* ---- Shape-from-Template solve ------------------------------------------------------------ */
typedef struct dsh_sft_frame {
  const float* Tcw;            /* 4x4 row-major float32 (cv::Mat pFrame->mTcw), initial camera pose */
  double K[4];                 /* fx, fy, cx, cy */
  int32_t n_frame;             /* pFrame->N: keypoints in the frame (DefOptimizer.cc:340 divides by it) */
  int32_t M;                   /* observations that enter the graph (DefOptimizer.cc:293-361) */
  const int32_t* obs_nodes;    /* M*3 node ids of the facet, ascending */
  const double* obs_bary;      /* M*3 */
  const double* obs_uv;        /* M*2 undistorted keypoints */
  const double* obs_invsig2;   /* M   mvInvLevelSigma2[octave] */
  const double* xyz;           /* n*3 current node positions */
  double reg_lap, reg_inex, reg_temp;
  int32_t neighbour_layers;    /* >=1: viewed nodes + 1-ring (the reference quirk, DefOptimizer.cc:388-406); 0: viewed only */
  int32_t max_iters;           /* 50 in the reference (DefOptimizer.cc:513); 0 = no iteration (the reference never does this): no edge error is
                                * ever computed, every observation counts as an inlier, repError is the mean reprojection error of the
                                * initial state -- what the oracle's restatement of g2o's freshly allocated edges gives */
} dsh_sft_frame;

typedef struct dsh_sft_result {
  float* Tcw;                  /* 4x4 float32 out (Converter::toCvMat) */
  double* pose7;               /* tx,ty,tz,qx,qy,qz,qw */
  double* xyz;                 /* n*3 */
  double* chi2_obs;            /* M: e^T Omega e of each observation at its last evaluation (DefOptimizer.cc:527) */
  uint8_t* outlier;            /* M: (float)chi2 > 5.991 */
  float* mappoint_xyz;         /* M*3 float32: DefMapPoint::RecalculatePosition of each observation's point */
  double rep_error;            /* mean reprojection error over inliers (pFrame->repError) */
  int32_t inliers;             /* return value of DefPoseOptimization */
  int32_t iters;               /* outer LM iterations executed */
  int32_t trials;              /* total damping trials (linear solves) */
  int32_t dim;                 /* 6 + 3*active nodes */
  int32_t half_bandwidth;      /* scalar half-bandwidth of the node block */
  int32_t status;              /* 0, or bit 0: a factorisation failed at least once */
  double* trace;               /* max_iters*DSH_TRACE_STRIDE doubles, may be NULL */
} dsh_sft_result;

/* One-shot: pack + upload + solve + download. */
int dsh_sft_solve(dsh_ctx* ctx, const dsh_sft_frame* frame, dsh_sft_result* result);

/* Batched / device-resident form (independent problems against the current template):
 *   dsh_sft_batch_upload  packs B frames on the host and starts ONE asynchronous copy to HBM on dsh_stream
 *                         (the frame buffers may be reused as soon as it returns),
 *   dsh_sft_batch_run     launches the solve on dsh_stream; may be called repeatedly -- every run restarts from the uploaded
 *                         initial state.  The launch shape is chosen at upload from the batch and the device:
 *                           - more than num_cus/2 problems, every half-bandwidth <= 128: the throughput shape.  From two problems
 *                             per compute unit upwards a step is rounds of three phase kernels over the whole batch (linearise /
 *                             factor + solve with ONE wavefront per problem / trial + controller) and, once no more than
 *                             T = min(4 num_cus, max(2 num_cus, 3 B / 4)) problems are still running, ONE launch of a tail kernel
 *                             in which every remaining problem gets a workgroup of eight wavefronts that runs it to its end (the
 *                             switch is decided on the device from the count of finished problems).  A batch of T problems or
 *                             fewer is run by the tail kernel alone.  The call enqueues launches and returns when every problem
 *                             has terminated;
 *                           - more than num_cus/2 problems with a wider band among them: one persistent kernel, one workgroup per
 *                             problem; the call returns at once;
 *                           - smaller batches (a tracked frame) run in latency mode: min(4, num_cus / B) workgroups per problem
 *                             try consecutive dampings of a Levenberg-Marquardt iteration side by side, one launch per round,
 *                             and the call returns when the problems have terminated; while the launch holds at most
 *                             num_cus/20 problems, a band of more than one tile that is long enough is cut in two parts
 *                             factored by two workgroups.
 *                         All shapes run the same Levenberg-Marquardt controller on the same normal equations; they differ
 *                         in the elimination order of the Cholesky factorisation, so the SAME frame solved in batches of
 *                         different size, or on devices with a different number of compute units, agrees to rounding
 *                         (vertices to ~1e-12 relative on the test templates), not bit for bit.  This holds INSIDE the
 *                         throughput shape as well: the tail kernel's eight-wavefront solver sums in another order than the
 *                         one-wavefront solver of the rounds (x of one damped system agrees to ~5e-13 relative), and which of
 *                         the two finishes a problem depends on B, on num_cus and on how many trials the OTHER problems of the
 *                         batch need -- the result bits of a problem depend on the batch it is solved in.  A fixed (batch,
 *                         device) reproduces itself bit for bit.  Problems that end in terminal stagnation (an iteration of >= 8
 *                         rejected dampings in a row: the steps are below one ulp of the state) may differ between shapes
 *                         in the number of rejected trials of that last iteration; the state returned agrees as above,
 *   dsh_sft_batch_download brings every result of the batch back with ONE copy (outlier classification, inlier count,
 *                         repError and the float32 map points are computed by the kernel) and waits for it.
 * To time the device work, record your own HIP events on dsh_stream() around dsh_sft_batch_run.
 * Measurement and debugging aids (per-phase timers, assembly-only launches, the dense normal equations of a problem,
 * solver A/B switches) are NOT part of this ABI: include/defslam_hip_debug.h, libdefslam_hip_lab.so. */
int dsh_sft_batch_upload(dsh_ctx* ctx, int B, const dsh_sft_frame* frames);
int dsh_sft_batch_run(dsh_ctx* ctx);
int dsh_sft_batch_download(dsh_ctx* ctx, int B, dsh_sft_result* results);
/* Totals of the last completed run (valid after a synchronise): outer iterations and trials over the batch. */
int dsh_sft_batch_counts(dsh_ctx* ctx, int64_t* iters, int64_t* trials);
/* Algorithmic bytes of one assembly pass of problem b (SURVEY 8d convention) and its edge counts
 * counts[9] = M, n_active, curvature edges (reference count), stretch edges, viewed nodes, dim,
 * half-bandwidth of the node block (scalars), wavefronts per problem of the launch shape chosen at upload (8, 4; 1 = rounds of
 * phase kernels with one wavefront per factorisation),
 * off-diagonal 3x3 blocks of H (lower triangle). */
int dsh_sft_batch_problem_info(dsh_ctx* ctx, int b, int64_t* assembly_bytes, int32_t* counts);

/* ---- shared-camera Shape-from-Template across GPUs ---------------------------------------------------------------------
 * BASELINE north star: "the path shards naturally over independent keyframes / mesh patches ... with RCCL all-reduce of the
 * shared camera-pose normal equations".  Every rank (one GPU, one context) holds one patch -- its own template, observations
 * and vertices -- and all patches are seen by ONE camera whose pose is estimated jointly (the reference's graph with a single
 * VertexSE3Expmap and the node vertices of every patch, DefOptimizer.cc:293-507).  The camera is the only coupling: each rank
 * factorises its node block and reduces to its 6x6 Schur complement of the camera; the ranks all-reduce that block, its
 * right-hand side and the scalars of the Levenberg-Marquardt control (32 doubles, three times per damping trial) and continue
 * with identical decisions.  The result equals the single-GPU solve of the union of the patches (tested).  The default for
 * independent problems stays dsh_sft_batch_*: no collective at all. */
typedef struct dsh_comm dsh_comm;
#define DSH_COMM_ID_BYTES 128
/* ncclGetUniqueId: call on one rank, hand the bytes to every rank (RCCL is loaded at the first call, not at library load). */
int dsh_comm_unique_id(void* id /* DSH_COMM_ID_BYTES */);
/* ncclCommInitRank on the context's GPU; collective over the nranks processes. */
int dsh_comm_create(dsh_ctx* ctx, int nranks, int rank, const void* id, dsh_comm** out);
int dsh_comm_destroy(dsh_comm* comm);
/* Collective: every rank passes its own patch (frame->Tcw, K, n_frame must agree); result as dsh_sft_solve, per patch, with
 * the joint pose.  The regulariser weights use the joint counts (all optimised nodes / stretch edges of all patches). */
int dsh_sft_shared_solve(dsh_ctx* ctx, dsh_comm* comm, const dsh_sft_frame* frame, dsh_sft_result* result);
/* The same protocol inside one process over G contexts (normally on one GPU), the all-reduce done by a summation kernel:
 * how the protocol is validated against the single-GPU solve where only one GPU is available. */
int dsh_sft_shared_solve_group(int G, dsh_ctx* const* ctxs, const dsh_sft_frame* frames, dsh_sft_result* results);

/* ---- one CONNECTED template across two GPUs --------------------------------------------------------------------------------
 * The shared-camera mode above joins patches that only share the camera.  A connected mesh also couples across any cut through its
 * curvature, stretching and observation edges (DefOptimizer.cc:408-507): the band ordering of the unknowns is therefore cut at a
 * SEPARATOR of one bandwidth (the 2-ring halo of the cut), rank 0 factors the part in front of it, rank 1 the part behind it (in
 * reversed order), both all-reduce their Schur contributions to the separator + camera system (one all-reduce of about
 * (kd^2 / 2 + 8 kd) doubles per damping trial), solve that reduced system redundantly, back-substitute their own part, and a second
 * all-reduce (6 + 3 n_active doubles) assembles the update.  Residuals, Jacobians and the Levenberg-Marquardt control are replicated
 * (every rank passes the SAME frame and holds the whole state), so the ranks take identical decisions without further collectives.
 * The result equals dsh_sft_solve of the same frame (the same Cholesky factorisation in another elimination order; tested against
 * the oracle).  Needs half-bandwidth <= 256 and a band long enough to cut; exactly two ranks -- a further cut along the same
 * ordering would make an inner part carry the fill of a whole separator through every column. */
int dsh_sft_connected_solve(dsh_ctx* ctx, dsh_comm* comm, const dsh_sft_frame* frame, dsh_sft_result* result);
/* The same protocol inside one process over two contexts (normally on one GPU), the all-reduces done by a summation kernel: how the
 * protocol is validated where only one GPU is available.  results[2]: one per context (identical). */
int dsh_sft_connected_solve_group(dsh_ctx* ctx0, dsh_ctx* ctx1, const dsh_sft_frame* frame, dsh_sft_result* results);

/* ---- NRSfM mapping side ----------------------------------------------------------------------- */
/* Uniform bicubic B-spline (BBS::bbs_t, Thirdparty/BBS/bbs.h:41-50). */
typedef struct dsh_bbs {
  double umin, umax;
  int32_t nptsu;
  double vmin, vmax;
  int32_t nptsv;
  int32_t valdim;
} dsh_bbs;

/* BBS::eval (Thirdparty/BBS/bbs.h:59, bbs.cc:155-195): val[valdim*k + d] = d^du d^dv spline(u_k, v_k).
 * ctrl: valdim x (nptsu*nptsv), index valdim*(iu*nptsv + iv) + d.  outside[k] (may be NULL) = 1 for a site outside the
 * definition domain (the reference reads out of bounds there; here the value is 0). */
int dsh_bbs_eval(dsh_ctx* ctx, const dsh_bbs* bbs, const double* ctrl, const double* u, const double* v, int n, int du, int dv,
                 double* val, uint8_t* outside);
/* Row view of BBS::coloc / BBS::coloc_deriv (bbs.h:61-63, bbs.cc:214-355): for site k the 16 (column, weight) pairs in
 * (iu, iv) order, cols[16k + 4iu + iv] = (iu+Iu)*nptsv + iv+Iv.  n_outside counts sites outside the domain (the reference
 * returns error code 1 for them); their columns are -1. */
int dsh_bbs_coloc(dsh_ctx* ctx, const dsh_bbs* bbs, const double* u, const double* v, int n, int du, int dv, int32_t* cols, double* w,
                  int32_t* n_outside);

/* The float32 fields of defSLAM::DiffProp the normal solve reads (Modules/Mapping/diffProp.h:52-83), in this order. */
typedef struct dsh_diffprop {
  float I1u, I1v, I2u, I2v;
  float J12a, J12b, J12c, J12d;
  float J21a, J21b, J21c, J21d;
  float H12uux, H12uuy, H12uvx, H12uvy, H12vvx, H12vvy;
} dsh_diffprop;

/* NormalEstimator::ObtainK1K2 (Modules/Mapping/NormalEstimator.h:53, NormalEstimator.cc:38-229) over the P map points that
 * have new observations.
 *   rec_ptr[P+1]            CSR: DiffProp records of point p are rec_ptr[p] .. rec_ptr[p+1]-1
 *   rec_is_ref[R]           record.KFToKF.first is the point's reference keyframe (it contributes a residual block)
 *   rec_first_normal[R*2], rec_has_first_normal[R]   (k1,k2) stored for the record's first keyframe (used by non-ref records)
 *   x0[P*2], has_x0[P]      previous normal of the reference keyframe; without it the start is (0,-0)
 *   ref_uv[P*2]             mpKeypointNorm of the point in its reference keyframe
 * Outputs: k1k2[P*2]; cov[P*4] (may be NULL); status[P]: 0 solved and written, 1 no residual block, 2 covariance failed
 * (rank-deficient Jacobian -> the reference skips the point); normal_ref[P*3] float = (k1,k2,1-k1 u-k2 v) (may be NULL);
 * normal_rec[R*3] float + rec_written[R]: normals propagated to the second keyframe of each record (may be NULL); iters[P]
 * (may be NULL). */
int dsh_normals_estimate(dsh_ctx* ctx, int P, const int32_t* rec_ptr, const dsh_diffprop* recs, const uint8_t* rec_is_ref,
                         const float* rec_first_normal, const uint8_t* rec_has_first_normal, const float* x0, const uint8_t* has_x0,
                         const float* ref_uv, double* k1k2, double* cov, int32_t* status, float* normal_ref, float* normal_rec,
                         uint8_t* rec_written, int32_t* iters);

/* Schwarzian-regularised B-spline warp between two keyframes.
 * Parameters x[2N], N = nptsu*nptsv: x[0..N) first coordinate of the control points, x[N..2N) second (index iu*nptsv+iv).
 * kp1 / kp2: P normalised key points (float32 x,y) of the reference / current keyframe; invsig[P] = sqrt(invSigma2[octave]);
 * fx_slot / fy_slot: the values the reference passes in Warp's (fx, fy) slots (SchwarpDatabase.cc:200-201 passes (fy, fx)).
 * dsh_schwarp_eval: the two Ceres cost functions evaluated once -- Warps::Warp::Evaluate (Schwarp.cc:235-303) in rows
 *   [0, 2P) and Warps::Schwarzian::Evaluate (Schwarp.cc:368-543) in rows [2P, 2P+4N); jacobian (may be NULL) is dense
 *   row-major (2P+4N) x 2N, including the reference's overwritten y-rows of the warp block. */
int dsh_schwarp_eval(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fx_slot,
                     double fy_slot, double lambda, const double* x, double* residuals, double* jacobian);
/* SchwarpDatabase::calculateSchwarps (SchwarpDatabase.cc:145-349): HuberLoss(5.77) on the warp block, Levenberg-Marquardt
 * (max_iters = 3 in the reference), then the DiffProp record of every match (diff[P], may be NULL together with drop) and
 * drop[p] = 1 when its reprojection error exceeds 10 px (fx, fy = KF->fx, KF->fy).  x is in/out.
 * info[0] = iterations, info[1] = accepted steps; costs[0] initial, costs[1] final cost (both may be NULL). */
int dsh_schwarp_fit(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, const float* invsig, double fx_slot,
                    double fy_slot, double lambda, float fx, float fy, int max_iters, double* x, dsh_diffprop* diff, uint8_t* drop,
                    int32_t* info, double* costs);

/* The same fit for B keyframe pairs at once (SchwarpDatabase::add fits one warp per anchor keyframe of the new keyframe,
 * SchwarpDatabase.cc:50-128): one copy up, a fixed sequence of launches in which all fits advance together with the
 * trust-region control on the device (no host round trip), one copy back.  A fit's result does not depend on what else is in the batch
 * (dsh_schwarp_fit is the batch of one). */
typedef struct dsh_schwarp_problem {
  dsh_bbs bbs;
  int32_t P;
  const float* kp1;            /* P x 2 */
  const float* kp2;            /* P x 2 */
  const float* invsig;         /* P */
  double fx_slot, fy_slot, lambda;
  float fx, fy;
  int32_t max_iters;
  double* x;                   /* in/out, 2 N */
  dsh_diffprop* diff;          /* P, may be NULL together with drop */
  uint8_t* drop;               /* P */
  int32_t info[2];             /* out: iterations, accepted steps */
  double costs[2];             /* out: initial, final cost */
  double init_lambda;          /* > 0: x is an output only -- the fit starts from Warps::Warp::initialize (Schwarp.cc:99-160, see
                                  dsh_warp_initialize below) with this bending weight, computed on the device as the first stage of the
                                  batch; <= 0: x holds the caller's start value */
  int32_t init_ok;             /* out: the verdict of that initialisation (1 when none was asked for) */
} dsh_schwarp_problem;
int dsh_schwarp_fit_batch(dsh_ctx* ctx, int B, dsh_schwarp_problem* problems);

/* ---- device-resident mapping chain --------------------------------------------------------------------------------------
 * defSLAM::WarpDatabase keeps the DiffProp records of every map point in a host map (WarpDatabase.h:61 mapPointsDB_): the fits write
 * them (SchwarpDatabase.cc:299-345), NormalEstimator::ObtainK1K2 reads them back (NormalEstimator.cc:50-110).  dsh_diffdb is that
 * database in HBM: dsh_schwarp_fit_batch_store appends the records of its fits on the device (in fit, match order; only the drop flags
 * travel to the host), dsh_normals_estimate_db groups the records of the requested points on the device (a point's records in their
 * insertion order, like the host vector) and solves -- key points in, normals out, no record crosses PCIe. */
typedef struct dsh_diffdb dsh_diffdb;
/* capacity_records is the initial capacity: like the reference's map the database grows on demand (appends and stores reserve their
 * worst case first, so a call stores all of its records or fails without storing any).  Lifetime: a database belongs to the context it
 * was created on; dsh_destroy of that context detaches it -- every call on it then returns DSH_ERR_ARG -- and dsh_diffdb_destroy works
 * before or after dsh_destroy. */
int dsh_diffdb_create(dsh_ctx* ctx, int64_t capacity_records, dsh_diffdb** out);
int dsh_diffdb_destroy(dsh_diffdb* db);
int dsh_diffdb_clear(dsh_diffdb* db);                 /* forget every record (WarpDatabase::clear) */
int64_t dsh_diffdb_count(const dsh_diffdb* db);       /* records stored */
/* Records a host already holds (a map loaded from elsewhere, tests): point_id[n] >= 0, tag[n] / idx2[n] may be NULL (0 / the index). */
int dsh_diffdb_append(dsh_diffdb* db, int n, const dsh_diffprop* recs, const int32_t* point_id, const int32_t* tag, const int32_t* idx2);
/* What dsh_schwarp_fit_batch_store needs per problem besides the fit itself: the map point of every match (point_id[P]; < 0: the record
 * is not stored -- the reference stores only points whose reference keyframe is the pair's first keyframe, SchwarpDatabase.cc:297), the
 * key point index of the match in the second keyframe (idx2[P], NULL = the match index) and a tag the caller chooses for the keyframe
 * pair; both come back with the propagated normals. */
typedef struct dsh_schwarp_store {
  const int32_t* point_id;
  const int32_t* idx2;
  int32_t tag;
} dsh_schwarp_store;
/* dsh_schwarp_fit_batch + storing: problems[b].diff may be NULL (no record is copied to the host), problems[b].drop receives the drop
 * flags (the host bookkeeping of SchwarpDatabase.cc:283-293 needs them).  Records of matches that are dropped or have point_id < 0 are
 * not stored.  DSH_ERR_STATE when the database is full. */
int dsh_schwarp_fit_batch_store(dsh_ctx* ctx, int B, dsh_schwarp_problem* problems, const dsh_schwarp_store* stores, dsh_diffdb* db);
/* NormalEstimator::ObtainK1K2 over the database for the P map points point_ids[P] (x0 / has_x0 / ref_uv and the per-point outputs as in
 * dsh_normals_estimate; every stored record is a residual block of its point; the ids are distinct, ids without records are
 * skipped like a point without observations).  Per-record outputs (all may be NULL), n_rec entries in
 * point order then insertion order, the buffers holding max_rec entries: rec_point (index into point_ids), rec_tag, rec_idx2, the
 * normal propagated to the second keyframe (normal_rec[3 n]) and whether the reference writes it (rec_written). */
int dsh_normals_estimate_db(dsh_ctx* ctx, dsh_diffdb* db, int P, const int32_t* point_ids, const float* x0, const uint8_t* has_x0, const float* ref_uv,
                            double* k1k2, double* cov, int32_t* status, float* normal_ref, int32_t* iters, int32_t max_rec, int32_t* n_rec,
                            int32_t* rec_point, int32_t* rec_tag, int32_t* rec_idx2, float* normal_rec, uint8_t* rec_written);

/* ---- Shape from Normals -------------------------------------------------------------------------
 * ShapeFromNormals::ShapeFromNormals + ::estimate (Modules/Mapping/ShapeFromNormals.cc:38-171, obtainM :178-260): the
 * depth B-spline (valdim 1, bbs->valdim is ignored) of a keyframe from the normals of its map points.
 *   n sites (u[n], v[n] normalised key point coordinates, normals[3n] float32 as stored by Surface::getNormalSurfacePoint;
 *   the caller filters bad map points / missing normals exactly like obtainM does); bending_weight = the constructor's
 *   bendingWeight_; mean_depth = DefKeyFrame::accMean; n_all key points (u_all, v_all) receive a surface point.
 * Least squares  min |M x|^2 + |Bend x|^2 + (sum x - N mean_depth)^2  over the N = nptsu*nptsv control points, then the
 * reference's scale: ctrl = x / float(median of float(x)) (Surface::saveArray), pts[3 n_all] = float (u d, v d, d) with
 * d = BBS eval of ctrl (Surface::set3DSurfacePoint).  ctrl_raw (may be NULL) receives x before the scaling.
 * *ok = 0 (and DSH_OK) when the reference's estimate() would return false: no key points, rank-deficient system, NaN/Inf. */
int dsh_sfn_estimate(dsh_ctx* ctx, const dsh_bbs* bbs, int n, const double* u, const double* v, const float* normals, double bending_weight,
                     double mean_depth, int n_all, const double* u_all, const double* v_all, double* ctrl_raw, double* ctrl, float* pts, int32_t* ok);
/* The same with the normals taken on the device from the last dsh_normals_estimate_db of db (they never visit the host): sel[n] >= 0 is
 * the index of a point in that call's point_ids (its normal in the reference keyframe, normal_ref), sel[n] < 0 is record -1 - sel[n] of
 * that call's per-record order (the normal propagated to the record's second keyframe, normal_rec).  The caller picks solved points /
 * written records from the status and rec_written arrays that call returned, like obtainM filters missing normals. */
int dsh_sfn_estimate_db(dsh_ctx* ctx, const dsh_bbs* bbs, const dsh_diffdb* db, int n, const int32_t* sel, const double* u, const double* v, double bending_weight,
                        double mean_depth, int n_all, const double* u_all, const double* v_all, double* ctrl_raw, double* ctrl, float* pts, int32_t* ok);
/* Warps::Warp::initialize (Modules/Mapping/Schwarp.cc:99-160): the control points of the warp kp1 -> kp2 that start the
 * Schwarzian fit, (C^T C + Bending(lambda)) X = C^T kp2 with C the colocation matrix of the P key points kp1 (float32 x,y
 * pairs, normalised coordinates).  x[2N]: first coordinate of the N control points, then the second (the layout
 * dsh_schwarp_fit takes).  *ok = 0 when the matrix is not positive definite (too few matches for this lambda). */
int dsh_warp_initialize(dsh_ctx* ctx, const dsh_bbs* bbs, int P, const float* kp1, const float* kp2, double lambda, double* x, int32_t* ok);
/* DefORBmatcher::searchBySchwarp (Modules/Matching/DefORBmatcher.cc:189-294): for each of the Q query key points of keyframe
 * 1 (the caller keeps the reference's filter :200-211: map point present, not bad, not yet in keyframe 2; kp1 = mpKeypointNorm,
 * desc1 = the 32-byte ORB descriptor rows) predict the position in keyframe 2 through the warp x[2N] (Warp::getEstimates,
 * float32 key point), convert to pixels with cam2 = {fx, fy, cx, cy}, skip predictions outside bounds2 = {mnMinX, mnMaxX,
 * mnMinY, mnMaxY} (KeyFrame::IsInImage), and among the key points of keyframe 2 (kp2 = mvKeysUn in pixels, desc2) that lie in
 * the search window of KeyFrame::GetFeaturesInArea(x, y, radius) (grid_cols x grid_rows = FRAME_GRID_COLS x FRAME_GRID_ROWS)
 * and have no map point (has_mp2[j] == 0) take the one with the smallest Hamming distance below th_low (TH_LOW = 50); among
 * equal distances the first one in the reference's visiting order (grid column, grid row, index).
 * match[q] = index in keyframe 2 or -1; *nmatches (may be NULL) = number of matches.  Bit-exact index parity. */
int dsh_search_by_schwarp(dsh_ctx* ctx, const dsh_bbs* bbs, const double* x, int Q, const float* kp1, const uint8_t* desc1, const float* cam2,
                          const float* bounds2, int grid_cols, int grid_rows, int N2, const float* kp2, const uint8_t* desc2, const uint8_t* has_mp2,
                          float radius, int th_low, int32_t* match, int32_t* nmatches);
/* BBS bending matrix (Thirdparty/BBS/bbs.cc:556-641 bending_ur, bbs_coloc.cc:406-508 BendingEigen) as a dense symmetric
 * N x N matrix, host side (the constant part of the Shape-from-Normals system). */
int dsh_bbs_bending(const dsh_bbs* bbs, double lambda, double* bending);

/* ---- surface registration ---------------------------------------------------------------------- */
/* Barycentric embedding on the device: same contract and results as dsh_template_embed (TriangularMesh.cc:133-236), one
 * wavefront per point (closest node, then the node's facets in index order).  Needs a template built from facets. */
int dsh_template_embed_device(dsh_ctx* ctx, int P, const float* pts /* P*3 */, int32_t* facet_id, int32_t* nodes /* P*3 */,
                              float* bary /* P*3 */);
/* GroundTruthTools::scaleMinMedian(PosMono, PosStereo) (Modules/GroundTruth/GroundTruthCalculator.cc:54-160).  The reference
 * draws `(double)rand() / RAND_MAX` while it runs; here the stream of those numbers is an input (u[k] = the k-th draw,
 * consumed in the reference's order: one per point i, and n-1 more for every i that was selected), so a shim that fills it
 * from rand() reproduces the reference.  *status: 0 ok, 1 stream too short (DSH_ERR_ARG is returned as well), 2 the
 * reference's early `return 0.0` (a selected point whose own selection holds fewer than two residuals; the reference reads
 * past the end of a vector when it holds none -- defined as the same early return).  *consumed (may be NULL) = draws used
 * when status is 0. */
int dsh_scale_min_median(dsh_ctx* ctx, int n, const float* pos_mono /* n*3 */, const float* pos_stereo /* n*3 */, const double* u,
                         int64_t nu, float* scale, int64_t* consumed, int32_t* status);
/* Optimizer::OptimizeHorn(pts1, pts2, g2oS12, chi, huber) (Modules/Tracking/DefOptimizer.cc:840-922): Levenberg-Marquardt on
 * one Sim(3) vertex with edges e_i = pts2_i - S.map(pts1_i), Huber(sqrt(huber) as float), numeric Jacobians (g2o's central
 * differences, delta 1e-9), optimize(50) twice.  sim3[8] = {qx, qy, qz, qw, tx, ty, tz, s}: in the initial estimate, out the
 * estimate after the FIRST optimize (what the reference reads back, :896).  *acceptable = the function's return value.
 * info[6] (may be NULL) = {plain chi2 of all edges after the second optimize, count, iterations 1st, iterations 2nd, damping
 * trials 1st, trials 2nd}. */
int dsh_optimize_horn(dsh_ctx* ctx, int n, const float* pts1 /* n*3 */, const float* pts2 /* n*3 */, double* sim3 /* 8 */, double chi,
                      double huber, int32_t* acceptable, double* info /* 6 */);
/* SurfaceRegistration::registerSurfaces (Modules/Mapping/SurfaceRegistration.cc:48-153), numeric part: cloud_surface = the
 * keyframe's surface points in world coordinates (cloud2pc), cloud_map = the map points' positions at that keyframe
 * (cloud1pc), Twc = the keyframe's inverse pose (4x4 row-major float32).  Fewer than 15 pairs -> *registered = 0.  Otherwise
 * scaleMinMedian(cloud_surface, cloud_map) initialises the scale, OptimizeHorn (chi = chi_limit^2, huber 0.01) aligns, and
 * unless (!acceptable && check_chi) the Sim(3) is composed with Twc: *s22 = recovered scale (Surface::applyScale takes it),
 * Tcw_new = the new keyframe pose (float32).  The clouds stay on the device between the two steps.
 * info[8] (may be NULL) = {initial scale, chi2, count, iterations 1st/2nd, trials 1st/2nd, acceptable}. */
int dsh_surface_register(dsh_ctx* ctx, int n, const float* cloud_surface /* n*3 */, const float* cloud_map /* n*3 */, const double* u,
                         int64_t nu, const float* Twc /* 16 */, double chi_limit, int check_chi, int32_t* registered, double* sim3 /* 8 */,
                         double* s22, float* Tcw_new /* 16 */, double* info /* 8 */);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DEFSLAM_HIP_H */
