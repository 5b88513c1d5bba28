// C ABI of libdefslam_hip.so: context, template, SfT pack / upload / run / download.
// Declared in include/defslam_hip.h.  Compiled with hipcc (host side).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/defslam_hip.h"
#include "dsh_ctx.h"
#include "dsh_template.h"
#include "sft_pack.h"
#include "sft_problem.h"

#ifdef DSH_LAB
#include "../../include/defslam_hip_debug.h"
#endif

extern "C" hipError_t sft_lm_launch(const SftDev* d_probs, int B, int max_kd, size_t jl_doubles, int nw, size_t* configured, hipStream_t stream);
extern "C" size_t sft_lm_kernel_lds_bytes(int kd, size_t jl_doubles);
extern "C" hipError_t sftb_launch(const SftDev* d_probs, SftRun* d_runs, int* d_counters, int* d_list, int B, int phase, size_t jl_doubles, size_t xyz_doubles, size_t* configured, int num_cus, int tail_below, hipStream_t stream);
extern "C" hipError_t sftb_tail_launch(const SftDev* d_probs, SftRun* d_runs, int* d_counters, int B, int max_kd, size_t jl_doubles, size_t* configured, int num_cus, int tail_below, hipStream_t stream);
extern "C" hipError_t sft_spec_launch(const SftDev* d_probs, SftSpec* d_spec, int B, int K, int phase, int nh, int owner_waves, int max_kd, size_t jl_doubles, size_t* configured, hipStream_t stream);

// The dynamic LDS size a kernel has been enabled for (hipFuncSetAttribute) is a property of the (device, kernel) pair, not of a context: two
// contexts on one GPU -- tracking and mapping, say -- must not lower each other's setting.  One high-water mark per device and kernel for the
// whole process; the launchers only ever raise it, under this lock.
struct LdsMarks { size_t lm[2] = {0, 0}, sc = 0, spec[2] = {0, 0}, cn = 0, b[2] = {0, 0}, tail = 0; };
static LdsMarks g_lds_marks[64];
static std::mutex g_lds_mu;
#define LDS_MARKS(c) (g_lds_marks[(c)->device & 63])
#define LDS_LOCK() std::lock_guard<std::mutex> lds_lock__(g_lds_mu)
extern "C" hipError_t sft_sc_launch(const SftDev* d_probs, SftSc* d_sc, int B, int phase, int max_kd, size_t jl_doubles, size_t* configured, hipStream_t stream);
extern "C" hipError_t sft_sc_local_reduce(SftSc* const* d_ptrs, int G, hipStream_t stream);
extern "C" hipError_t sft_cn_launch(const SftDev* d_probs, SftSc* d_sc, int phase, int max_kd, size_t jl_doubles, size_t* configured, hipStream_t stream);
extern "C" hipError_t sft_vec_sum2(const double* a, const double* b, double* out_a, double* out_b, int n, hipStream_t stream);
#ifdef DSH_LAB
extern "C" hipError_t sft_wave_lab_launch(const SftDev* d_probs, int B, int which, double rel, int max_kd, size_t jl_doubles, hipStream_t stream);
extern "C" hipError_t sft_assembly_launch(const SftDev* d_probs, int B, int max_kd, size_t jl_doubles, int nw, hipStream_t stream);
#endif

namespace {

constexpr int kNB = 32;  // must match NB in sft_kernels.hip
constexpr int kTS = 16, kBT = 8, kWB = 16;  // must match TS / BT in sft_kernels.hip and WB in sft_wide.h

using dsh::Tcw_from_pose7;

// ---- one packed problem on the host: the shared structure (graph) + the per-frame lists + the scalars of the device record
struct Packed {
  SftDev h{};                          // sizes + scalars (pointers filled at upload)
  dsh::SftGraph* g = nullptr;          // owned by the context's graph cache
  dsh::SftFramePack f;
};

// Host buffer of a context that the copy engine reads / writes directly: page-locked for a GPU context (hipMemcpyAsync
// from pageable memory is staged and synchronous), plain memory for a host-only one.  Grow-only.
struct HostBuf {
  char* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  hipError_t ensure(size_t bytes, bool want_pinned) {
    if (bytes <= cap) return hipSuccess;
    release();
    const size_t want = bytes + bytes / 4 + 4096;
    if (want_pinned) {
      const hipError_t e = hipHostMalloc((void**)&p, want, hipHostMallocDefault);
      if (e != hipSuccess) { p = nullptr; return e; }
      pinned = true;
    } else {
      p = static_cast<char*>(std::malloc(want));
      if (!p) return hipErrorOutOfMemory;
      pinned = false;
    }
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) { if (pinned) (void)hipHostFree(p); else std::free(p); }
    p = nullptr; cap = 0;
  }
};

struct Arena {  // byte layout of a device allocation, 256-byte aligned slices
  size_t size = 0;
  size_t take(size_t bytes) {
    const size_t off = size;
    size += (bytes + 255) & ~size_t(255);
    return off;
  }
};

}  // namespace

extern "C" hipError_t reg_embed(int, const float*, int, const double*, const int32_t*, const int32_t*, const int32_t*, int32_t*, int32_t*, float*, hipStream_t);

struct dsh_ctx : dsh_ctx_base {
  dsh::TemplateHost tmpl;
  // device copy of the template
  char* d_tmpl = nullptr;
  size_t d_tmpl_bytes = 0;
  struct {
    const double *xyz0, *nbr_w, *nbr_sumw, *k0;
    const int32_t *nbr_ptr, *nbr_idx;
  } dt{};
  // structure of the normal equations per active set of the current template (sft_pack.h), device-resident, built on first use
  std::vector<std::unique_ptr<dsh::SftGraph>> graphs;
  uint64_t upload_serial = 0;     // graphs touched by the upload in progress carry it (eviction keeps them)
  // batch
  int B = 0;
  std::vector<Packed> packed;
  char* d_batch = nullptr;
  size_t d_batch_cap = 0;
  SftDev* d_probs = nullptr;       // inside d_batch
  std::vector<SftDev> h_probs;     // host mirror with device pointers
  HostBuf stage;                   // page-locked staging of the read-only part (one hipMemcpyAsync per upload)
  hipEvent_t stage_free = nullptr; // recorded behind the upload copy: the staging buffer may be refilled once it has fired
  bool stage_busy = false;
  HostBuf results;                 // page-locked landing zone of the result region (one hipMemcpyAsync per download)
  size_t ro_bytes = 0;             // leading read-only bytes of d_batch (uploaded)
  size_t res_off = 0, res_bytes = 0;            // result region of d_batch: B headers (SftResHdr), then the bodies
  struct ResOffs { size_t xyz, chi2, trace, mp, outl; };   // offsets inside the result region
  std::vector<ResOffs> res_offs;
  int max_kd = 0;
  size_t jl_doubles = 0;
  size_t xyz_doubles = 0;   // LDS copy of the node positions in the TRIAL kernel of the phase rounds (largest problem of the batch)
  int nw = 8;        // wavefronts per problem of the persistent kernel (4: two problems share a CU)
  // latency mode: K workgroups per problem run the next K damping trials of an iteration side by side (sft_kernels.hip: sft_spec_kernel)
  int spec_k = 1;
  int spec_nh = 0;                     // helper workgroups per part of a two-sided factorisation (FACTOR launches of the latency mode, sft_wide.h)
  char* d_sync = nullptr;              // their progress words and column flags (inside the batch arena), cleared at the start of every run
  size_t sync_bytes = 0;
  int spec_hint = 12;                  // launches the previous speculative run needed (first group of the next one)
  int max_iters_batch = 0;
  SftSpec* d_spec = nullptr;           // K*B controller states, inside d_batch
  size_t spec_bytes = 0;
  HostBuf spec_done;                   // page-locked: lane 0's SftSpec of every problem (the done flag)
  SftSc* d_sc = nullptr;               // shared-camera mode: LM state between the phase kernels
  int force_waves = 0;                 // set while the shared-camera mode packs its problem (always the 8-wavefront shape)
  bool force_split = false;            // set while the connected-mesh mode packs its problem: the two-sided cut with one workgroup (rank) per part
  // throughput shape (sft_batch.h): rounds of LIN / FACTOR / TRIAL launches over the whole batch, one wavefront per factorisation
  bool rounds_mode = false;
  SftRun* d_runs = nullptr;            // B controller states + the done counter behind them, inside d_batch
  int* d_counters = nullptr;
  int* d_linlist = nullptr;            // B ints behind the counters: the problems the next LIN launch linearises (sft_batch.h)
  int rounds_hint = 24;                // rounds the previous run of this context needed
  // The batch runs as up to kMaxSub sub-batches on streams of their own: the launches of a round are enqueued sub-batch by sub-batch, so
  // the tail of one sub-batch's FACTOR launch (waves that have run out of work) overlaps with the next launches of the others.
  static constexpr int kMaxSub = 4;
  hipStream_t sub_stream[kMaxSub] = {nullptr, nullptr, nullptr, nullptr};   // [0] = stream
  hipEvent_t sub_event[kMaxSub] = {nullptr, nullptr, nullptr, nullptr};
  int n_sub = 1;
  std::vector<hipEvent_t>* phase_events = nullptr;   // lab builds (dsh_lab_sft_rounds_timed): an event in front of and behind every phase launch
  std::vector<int> phase_ids;                        // ... and which phase it was (SFTB_PH_*)
  int num_cus = 256;
  bool ran = false;
  // Solver selection.  The product library always takes the defaults; libdefslam_hip_lab.so can override them through
  // dsh_lab_set_option (include/defslam_hip_debug.h) for A/B runs.  No environment variables are read.
  struct { int waves = 0; int dataflow = 1; int wide_off = 0; int speculate = 0; int split = 2; int rounds = 1; int streams = 0; int helpers = -1; int tail = -1; int owner_waves = 8; int helpers_wbt = 12; } opt;
  bool any_split = false;              // some problem of the batch runs the two-sided factorisation (SftPart): a FACTOR launch precedes every trial launch
};

namespace {

int fail(dsh_ctx* c, int code, const std::string& m) {
  if (c) c->err = m;
  return code;
}
#define HIPCHK(c, call)                                                                              \
  do {                                                                                               \
    hipError_t e__ = (call);                                                                         \
    if (e__ != hipSuccess) return fail(c, DSH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

void drop_graphs(dsh_ctx* c) {
  if (!c->host_only && c->stream) (void)hipStreamSynchronize(c->stream);   // a batch in flight may still read them
  for (auto& g : c->graphs)
    if (g->d_base) (void)hipFree(g->d_base);
  c->graphs.clear();
  c->packed.clear();
}

int upload_template(dsh_ctx* c) {
  c->B = 0;
  c->ran = false;
  drop_graphs(c);
  if (c->host_only) return DSH_OK;
  const dsh::TemplateHost& t = c->tmpl;
  Arena a;
  const size_t o_xyz0 = a.take(sizeof(double) * 3 * t.n);
  const size_t o_ptr = a.take(sizeof(int32_t) * (t.n + 1));
  const size_t o_idx = a.take(sizeof(int32_t) * t.nbr_idx.size());
  const size_t o_w = a.take(sizeof(double) * t.nbr_w.size());
  const size_t o_sw = a.take(sizeof(double) * t.n);
  const size_t o_k0 = a.take(sizeof(double) * t.n);
  if (c->d_tmpl) { (void)hipFree(c->d_tmpl); c->d_tmpl = nullptr; }
  HIPCHK(c, hipMalloc((void**)&c->d_tmpl, a.size));
  c->d_tmpl_bytes = a.size;
  std::vector<char> st(a.size, 0);
  std::memcpy(&st[o_xyz0], t.xyz0.data(), sizeof(double) * 3 * t.n);
  std::memcpy(&st[o_ptr], t.nbr_ptr.data(), sizeof(int32_t) * (t.n + 1));
  std::memcpy(&st[o_idx], t.nbr_idx.data(), sizeof(int32_t) * t.nbr_idx.size());
  std::memcpy(&st[o_w], t.nbr_w.data(), sizeof(double) * t.nbr_w.size());
  std::memcpy(&st[o_sw], t.nbr_sumw.data(), sizeof(double) * t.n);
  std::memcpy(&st[o_k0], t.k0.data(), sizeof(double) * t.n);
  HIPCHK(c, hipMemcpy(c->d_tmpl, st.data(), a.size, hipMemcpyHostToDevice));
  c->dt.xyz0 = (const double*)(c->d_tmpl + o_xyz0);
  c->dt.nbr_ptr = (const int32_t*)(c->d_tmpl + o_ptr);
  c->dt.nbr_idx = (const int32_t*)(c->d_tmpl + o_idx);
  c->dt.nbr_w = (const double*)(c->d_tmpl + o_w);
  c->dt.nbr_sumw = (const double*)(c->d_tmpl + o_sw);
  c->dt.k0 = (const double*)(c->d_tmpl + o_k0);
  c->B = 0;
  c->ran = false;
  return DSH_OK;
}

template <class T>
size_t reserve(Arena& a, const std::vector<T>& v) { return a.take(sizeof(T) * v.size()); }
template <class T>
void put(char* st, size_t off, const std::vector<T>& v) {
  if (!v.empty()) std::memcpy(st + off, v.data(), sizeof(T) * v.size());
}

// The graph of the frame's active set: cached per template, uploaded once, shared by every problem that has it.
int graph_for(dsh_ctx* c, const std::vector<uint8_t>& opt, dsh::SftGraph** out, std::string& err) {
  uint64_t h = 1469598103934665603ull;
  for (uint8_t b : opt) { h ^= b; h *= 1099511628211ull; }
  for (auto& g : c->graphs)
    if (g->opt_hash == h && g->opt == opt) { g->last_use = c->upload_serial; *out = g.get(); return DSH_OK; }
  if (c->graphs.size() >= 64) {
    // A long sequence with an ever-changing view: drop every graph the upload in progress does not use (rebuilding one costs a slow
    // frame).  Graphs of the problems already packed by THIS upload stay -- a batch with more than 64 active sets just grows the cache.
    if (!c->host_only && c->stream) (void)hipStreamSynchronize(c->stream);   // the previous batch may still read them
    auto& gs = c->graphs;
    for (size_t i = 0; i < gs.size();) {
      if (gs[i]->last_use != c->upload_serial) {
        if (gs[i]->d_base) (void)hipFree(gs[i]->d_base);
        gs.erase(gs.begin() + i);
      } else {
        i++;
      }
    }
  }
  std::unique_ptr<dsh::SftGraph> g(new dsh::SftGraph());
  const int rc = dsh::build_graph(c->tmpl, opt, *g, err);
  if (rc != DSH_OK) return rc;
  g->last_use = c->upload_serial;
  if (!c->host_only) {
    Arena a;
    auto& o = g->o;
    o.act = reserve(a, g->act); o.actnode = reserve(a, g->actnode); o.star_node = reserve(a, g->star_node); o.star_sL = reserve(a, g->star_sL);
    o.str_nodes = reserve(a, g->str_nodes); o.str_L0 = reserve(a, g->str_L0); o.off_ptr = reserve(a, g->off_ptr); o.off_rc = reserve(a, g->off_rc);
    o.sh_ptr = reserve(a, g->sh_ptr); o.sh_rec = reserve(a, g->sh_rec); o.sh_cf = reserve(a, g->sh_cf); o.tmask = reserve(a, g->tmask);
    o.hgather = reserve(a, g->hgather);
    o.hgatherT = reserve(a, g->hgatherT);
    std::vector<char> st(a.size, 0);
    put(st.data(), o.act, g->act); put(st.data(), o.actnode, g->actnode); put(st.data(), o.star_node, g->star_node); put(st.data(), o.star_sL, g->star_sL);
    put(st.data(), o.str_nodes, g->str_nodes); put(st.data(), o.str_L0, g->str_L0); put(st.data(), o.off_ptr, g->off_ptr); put(st.data(), o.off_rc, g->off_rc);
    put(st.data(), o.sh_ptr, g->sh_ptr); put(st.data(), o.sh_rec, g->sh_rec); put(st.data(), o.sh_cf, g->sh_cf); put(st.data(), o.tmask, g->tmask);
    put(st.data(), o.hgather, g->hgather);
    put(st.data(), o.hgatherT, g->hgatherT);
    if (hipMalloc((void**)&g->d_base, a.size) != hipSuccess) { err = "out of device memory (graph)"; return DSH_ERR_HIP; }
    g->d_bytes = a.size;
    if (hipMemcpy(g->d_base, st.data(), a.size, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(g->d_base); err = "graph upload failed"; return DSH_ERR_HIP; }
  }
  *out = g.get();
  c->graphs.push_back(std::move(g));
  return DSH_OK;
}

// Build the graph of DefOptimizer.cc:293-507 as flat arrays: shared structure from the cache, per-frame lists, scalars.
int pack_problem(dsh_ctx* c, const dsh_sft_frame& f, bool wide_off, Packed& P, std::string& err) {
  const dsh::TemplateHost& t = c->tmpl;
  std::vector<uint8_t> viewed, opt;
  int rc = dsh::frame_active_set(t, f, viewed, opt, err);
  if (rc != DSH_OK) return rc;
  rc = graph_for(c, opt, &P.g, err);
  if (rc != DSH_OK) return rc;
  const dsh::SftGraph& g = *P.g;
  rc = dsh::pack_frame(t, g, f, viewed, P.f, err);
  if (rc != DSH_OK) return rc;
  SftDev& h = P.h;
  h = SftDev{};
  h.n = t.n; h.nA = g.nA; h.Dn = 3 * g.nA; h.kd = g.kd; h.ldh = h.kd + 1;
  // solver per half-bandwidth: register-window tiles (<= 128), left-looking wide tiles (<= 256; the lab option "wide_off"
  // keeps the row-major band solver for A/B runs), row-major band otherwise
  h.tile_mode = (h.kd <= kTS * kBT) ? 1 : ((h.kd <= kTS * kWB && !wide_off) ? 2 : 0);
  h.wbt = h.tile_mode == 1 ? kBT : (h.tile_mode == 2 ? (h.kd + kTS - 1) / kTS : 0);
  h.tpr = h.tile_mode ? h.wbt + 1 : 0;
  h.M = f.M; h.V = P.f.V; h.S = g.S; h.Es = g.Es; h.noff = g.noff; h.max_iters = f.max_iters; h.mode = 0;
  h.fx = f.K[0]; h.fy = f.K[1]; h.cx = f.K[2]; h.cy = f.K[3];
  h.w_ref = f.reg_temp / std::pow(t.median_L, 2);              // DefOptimizer.cc:378
  h.w_curv = f.reg_lap / (double)g.nA;                         // :458  (|OptLap|)
  h.w_str = g.Es > 0 ? f.reg_inex / (double)g.Es : 0.0;        // :497  (|medges|)
  const float deltaMono = (float)std::sqrt(5.991);             // :286
  h.hub_delta = (double)deltaMono;
  h.hub_dsqr = h.hub_delta * h.hub_delta;
  return DSH_OK;
}

// One solve of the uploaded batch.  K == 1: the persistent kernel, one launch, asynchronous.  K > 1 (latency mode): one launch
// per round of K damping trials; an iteration that accepts one of its first K trials takes one launch, so max_iters + 1 launches
// finish the typical frame; the done flags are read back behind them and further rounds are launched only while needed.
int run_rounds_enqueue(dsh_ctx* c);
// (Blocking: the rounds end with read-backs of the done counters.  On an error every sub-stream is drained before the error is returned, so that
// what the caller enqueues next on the context's stream cannot race with work left on another one.)
int run_rounds(dsh_ctx* c) {
  const int rc = run_rounds_enqueue(c);
  if (rc != DSH_OK)
    for (int s = 0; s < c->n_sub; s++) if (c->sub_stream[s]) (void)hipStreamSynchronize(c->sub_stream[s]);
  return rc;
}
int run_rounds_enqueue(dsh_ctx* c) {
  // Throughput shape: every problem of the batch advances by one damping trial per round (LIN for those that start an iteration, FACTOR,
  // TRIAL).  As many rounds as the previous run needed are enqueued in one go, then the done counters are read back and rounds are added
  // in pairs while a problem still runs (a finished problem's workgroups leave at their first instruction).
  const int B = c->B, S = c->n_sub;
  int b0[dsh_ctx::kMaxSub + 1];
  for (int s = 0; s <= S; s++) b0[s] = (int)((long long)B * s / S);
  // The last problems of a step go to the tail kernel (sft_batch.h): one workgroup runs each of them to its end.  It pays from about two
  // problems per CU downwards (a round costs one whole one-wavefront factorisation, 1 ms, however few problems it carries; a workgroup of the
  // tail kernel takes 0.5 ms per trial).  WHEN the rounds end is decided on the device, by the first kernel of a round from the count the
  // previous round left -- the results do not depend on how the launches are grouped here; a tail launch in front of the switch, like a round
  // behind it, leaves at its first instruction.
  // The threshold (A/B over batch sizes, tools/tail_ab.py): four problems per CU -- but not more than three quarters of the batch (a batch of
  // four per CU would run in the tail kernel alone: 234 against 277 k it/s), and a batch of two per CU or less does run there alone.
  int tail_below = -1;
  if (S == 1 && c->opt.tail != 0) {
    if (c->opt.tail > 0) tail_below = c->opt.tail * c->num_cus;                                                    // (lab option: exactly this many per CU)
    else tail_below = std::min(4 * c->num_cus, std::max(2 * c->num_cus, (int)((long long)3 * B / 4)));
  }
  auto launch = [&](int s, int phase) {
    const bool ev = c->phase_events && S == 1;
    if (ev) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, c->stream); c->phase_events->push_back(e); c->phase_ids.push_back(phase); } }
    hipError_t r;
    {
      LDS_LOCK();
      if (phase == SFTB_PH_TAIL) r = sftb_tail_launch(c->d_probs + b0[s], c->d_runs + b0[s], c->d_counters + 16 * s, b0[s + 1] - b0[s], c->max_kd, c->jl_doubles, &LDS_MARKS(c).tail, c->num_cus, tail_below, c->sub_stream[s]);
      else r = sftb_launch(c->d_probs + b0[s], c->d_runs + b0[s], c->d_counters + 16 * s, c->d_linlist + b0[s], b0[s + 1] - b0[s], phase, c->jl_doubles, c->xyz_doubles, LDS_MARKS(c).b,
                           c->num_cus, tail_below, c->sub_stream[s]);
    }
    if (ev) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, c->stream); c->phase_events->push_back(e); } }
    return r;
  };
  // the other streams start behind whatever the context's stream has enqueued (the upload)
  if (S > 1) {
    HIPCHK(c, hipEventRecord(c->sub_event[0], c->stream));
    for (int s = 1; s < S; s++) HIPCHK(c, hipStreamWaitEvent(c->sub_stream[s], c->sub_event[0], 0));
  }
  for (int s = 0; s < S; s++) HIPCHK(c, launch(s, SFTB_PH_INIT));
  const int worst = std::max(1, c->max_iters_batch) * 10 + 1;
  // first group: the rounds the previous run of this context needed in front of its tail kernel (or, without one, to the end)
  int rounds = 0, group = std::max(1, std::min(worst, c->rounds_hint));
  if (tail_below >= B) group = 0;   // (a batch this small: the tail kernel from the start)
  HIPCHK(c, c->spec_done.ensure(64 * dsh_ctx::kMaxSub, true));
  int rc = DSH_OK;
  while (true) {
    for (int i = 0; i < group && rounds < worst; i++, rounds++)
      for (int s = 0; s < S; s++) {
        HIPCHK(c, launch(s, SFTB_PH_LIN));
        HIPCHK(c, launch(s, SFTB_PH_FACTOR));
        HIPCHK(c, launch(s, SFTB_PH_TRIAL));
      }
    if (tail_below >= 0) HIPCHK(c, launch(0, SFTB_PH_TAIL));
    // [0] finished problems, [6] tail mode, [7] the round that switched to it
    for (int s = 0; s < S; s++) HIPCHK(c, hipMemcpyAsync(c->spec_done.p + 64 * s, c->d_counters + 16 * s, 8 * sizeof(int), hipMemcpyDeviceToHost, c->sub_stream[s]));
    for (int s = 0; s < S; s++) HIPCHK(c, hipStreamSynchronize(c->sub_stream[s]));
    int done = 0;
    for (int s = 0; s < S; s++) done += *reinterpret_cast<const int*>(c->spec_done.p + 64 * s);
    const int* c0 = reinterpret_cast<const int*>(c->spec_done.p);
    if (done >= B) { c->rounds_hint = (tail_below >= 0 && c0[6]) ? std::max(1, c0[7]) : rounds; break; }
    if (rounds >= worst) { rc = fail(c, DSH_ERR_STATE, "batched rounds: a problem did not terminate within its trial budget"); break; }
    group = 2;
  }
  return rc;   // (every stream is idle here: what the caller enqueues on the context's stream is ordered behind all of them)
}

int run_once(dsh_ctx* c) {
  if (c->rounds_mode) return run_rounds(c);
  if (c->spec_k <= 1) {
    LDS_LOCK();
    HIPCHK(c, sft_lm_launch(c->d_probs, c->B, c->max_kd, c->jl_doubles, c->nw, LDS_MARKS(c).lm, c->stream));
    return DSH_OK;
  }
  const int K = c->spec_k, B = c->B;
  HIPCHK(c, hipMemsetAsync(c->d_spec, 0, c->spec_bytes, c->stream));
  const int rounds_per_iter = (10 + K - 1) / K;
  const int worst = c->max_iters_batch * rounds_per_iter;
  // A round = a linearisation launch (verdict on the previous round; on a new iteration the lanes assemble H together) + a trial
  // launch.  First group: as many rounds as the previous run of this context needed (tracking is coherent from frame to frame:
  // usually exact), then the done flags are read back and rounds of two are added while a problem still runs.  Launches behind the
  // end of a problem cost a few microseconds each (it leaves at the first instruction); a read-back costs a stream synchronisation.
  auto launch = [&](int phase) { LDS_LOCK(); return sft_spec_launch(c->d_probs, c->d_spec, B, K, phase, c->spec_nh, c->opt.owner_waves, c->max_kd, c->jl_doubles, LDS_MARKS(c).spec, c->stream); };
  if (c->spec_nh > 0) HIPCHK(c, hipMemsetAsync(c->d_sync, 0, c->sync_bytes, c->stream));   // progress words and column flags of the helper workgroups: epochs count from here
  HIPCHK(c, launch(SFT_SPEC_INIT));
  int rounds = 0, group = std::max(2, std::min(worst, c->spec_hint));
  HIPCHK(c, c->spec_done.ensure(sizeof(SftSpec) * (size_t)B, true));
  while (true) {
    for (int i = 0; i < group && rounds < worst; i++, rounds++) {
      HIPCHK(c, launch(SFT_SPEC_LIN));
      if (c->any_split) {   // two workgroups per lane: the two parts of the two-sided factorisation, then the solve (reduced problem + own part)
        HIPCHK(c, launch(SFT_SPEC_FACTOR));
        HIPCHK(c, launch(SFT_SPEC_SOLVE));
      }
      HIPCHK(c, launch(SFT_SPEC_TRIAL));
    }
    HIPCHK(c, launch(SFT_SPEC_LIN));   // the verdict on the last round (and, unless the problem is finished, the next linearisation)
    HIPCHK(c, hipMemcpyAsync(c->spec_done.p, c->d_spec, sizeof(SftSpec) * (size_t)B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const SftSpec* sp = reinterpret_cast<const SftSpec*>(c->spec_done.p);
    bool all_done = true;
    int needed = 0;
    for (int b = 0; b < B; b++) { all_done = all_done && sp[b].done; needed = std::max(needed, sp[b].launches); }
    if (all_done) { c->spec_hint = needed; break; }
    if (rounds >= worst) return fail(c, DSH_ERR_STATE, "speculative trials: a problem did not terminate within its launch budget");
    group = 2;
  }
  return DSH_OK;
}

}  // namespace

extern "C" {

int dsh_create(dsh_ctx** out, int device) {
  if (!out) return DSH_ERR_ARG;
  *out = nullptr;
  if (device == -1) {  // host-only context: template constants + packing, every GPU entry point fails loudly
    dsh_ctx* hc = new dsh_ctx();
    hc->device = -1;
    hc->host_only = true;
    *out = hc;
    return DSH_OK;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return DSH_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return DSH_ERR_ARG;
  dsh_ctx* c = new dsh_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return DSH_ERR_HIP;
  }
  if (hipEventCreateWithFlags(&c->stage_free, hipEventDisableTiming) != hipSuccess) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return DSH_ERR_HIP;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
  // streams of the sub-batches of the throughput shape: [0] is the context's stream; the others are created when a batch is first split (the
  // lab option "streams": measured and not the default).  Not before: HIP multiplexes the streams of a process onto four hardware queues, and a
  // context that holds three idle streams pushes the streams of OTHER contexts onto shared queues -- the two contexts of a connected-mesh or
  // shared-camera group then ran their phase kernels one after the other (17.5 instead of 9.5 ms per C2 frame next to a third context).
  c->sub_stream[0] = c->stream;
  if (hipEventCreateWithFlags(&c->sub_event[0], hipEventDisableTiming) != hipSuccess) c->sub_event[0] = nullptr;
  *out = c;
  return DSH_OK;
}

int dsh_destroy(dsh_ctx* c) {
  if (!c) return DSH_ERR_ARG;
  ddb_detach_all(c);   // databases of this context stay valid objects (dsh_diffdb_destroy still frees them) but no longer name it
  if (c->host_only) { c->stage.release(); c->results.release(); delete c; return DSH_OK; }
  (void)hipSetDevice(c->device);
  drop_graphs(c);
  for (int i = 1; i < dsh_ctx::kMaxSub; i++) if (c->sub_stream[i]) { (void)hipStreamSynchronize(c->sub_stream[i]); (void)hipStreamDestroy(c->sub_stream[i]); }
  for (int i = 0; i < dsh_ctx::kMaxSub; i++) if (c->sub_event[i]) (void)hipEventDestroy(c->sub_event[i]);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (c->stage_free) (void)hipEventDestroy(c->stage_free);
  c->stage.release();
  c->results.release();
  c->spec_done.release();
  if (c->d_tmpl) (void)hipFree(c->d_tmpl);
  c->scratch.release();
  c->pin_in.release();
  c->pin_out.release();
  if (c->d_batch) (void)hipFree(c->d_batch);
  if (c->d_sc) (void)hipFree(c->d_sc);
  delete c;
  return DSH_OK;
}

const char* dsh_last_error(const dsh_ctx* c) { return c ? c->err.c_str() : "null context"; }
void* dsh_stream(dsh_ctx* c) { return c ? (void*)c->stream : nullptr; }
int dsh_synchronize(dsh_ctx* c) {
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return DSH_ERR_NO_DEVICE;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return DSH_OK;
}

int dsh_template_build(dsh_ctx* c, int n, const double* xyz0, int F, const int32_t* facets) {
  if (!c || n <= 0 || F <= 0 || !xyz0 || !facets) return fail(c, DSH_ERR_ARG, "dsh_template_build: bad argument");
  for (int i = 0; i < 3 * F; i++)
    if (facets[i] < 0 || facets[i] >= n) return fail(c, DSH_ERR_ARG, "dsh_template_build: facet index out of range");
  if (!c->host_only) (void)hipSetDevice(c->device);
  c->tmpl.build(n, xyz0, F, facets);
  return upload_template(c);
}

int dsh_template_set(dsh_ctx* c, int n, const double* xyz0, const uint8_t* boundary, const int32_t* rp, const int32_t* col, const double* w,
                     const double* k0, int E, const int32_t* en, const double* eL, double median_L) {
  if (!c || n <= 0 || E < 0 || !xyz0 || !boundary || !rp || !col || !w || !k0 || !en || !eL) return fail(c, DSH_ERR_ARG, "dsh_template_set: bad argument");
  // The arrays become indices of the packer (inc[edge_nodes[..]], opt[nbr_col[..]]): a malformed CSR or edge list must be
  // refused here, not turn into an out-of-bounds write later.
  if (rp[0] != 0) return fail(c, DSH_ERR_ARG, "dsh_template_set: nbr_rowptr[0] != 0");
  for (int i = 0; i < n; i++)
    if (rp[i + 1] < rp[i]) return fail(c, DSH_ERR_ARG, "dsh_template_set: nbr_rowptr is not non-decreasing");
  for (int p = 0; p < rp[n]; p++)
    if (col[p] < 0 || col[p] >= n) return fail(c, DSH_ERR_ARG, "dsh_template_set: neighbour index out of range");
  for (int e = 0; e < E; e++) {
    if (en[2 * e] < 0 || en[2 * e] >= n || en[2 * e + 1] < 0 || en[2 * e + 1] >= n) return fail(c, DSH_ERR_ARG, "dsh_template_set: edge node out of range");
    if (!(eL[e] > 0.0)) return fail(c, DSH_ERR_ARG, "dsh_template_set: edge rest length must be positive");
  }
  if (!(median_L > 0.0)) return fail(c, DSH_ERR_ARG, "dsh_template_set: median edge length must be positive");
  if (!c->host_only) (void)hipSetDevice(c->device);
  c->tmpl.set(n, xyz0, boundary, rp, col, w, k0, E, en, eL, median_L);
  return upload_template(c);
}

int dsh_template_dims(const dsh_ctx* c, int32_t* n, int32_t* E, int32_t* nnz) {
  if (!c || !c->tmpl.valid) return DSH_ERR_STATE;
  if (n) *n = c->tmpl.n;
  if (E) *E = c->tmpl.E;
  if (nnz) *nnz = (int32_t)c->tmpl.nbr_idx.size();
  return DSH_OK;
}

int dsh_template_get(const dsh_ctx* c, uint8_t* boundary, int32_t* rp, int32_t* col, double* w, double* k0, int32_t* en, double* eL, double* med) {
  if (!c || !c->tmpl.valid) return DSH_ERR_STATE;
  const dsh::TemplateHost& t = c->tmpl;
  if (boundary) std::memcpy(boundary, t.boundary.data(), t.n);
  if (rp) std::memcpy(rp, t.nbr_ptr.data(), sizeof(int32_t) * (t.n + 1));
  if (col) std::memcpy(col, t.nbr_idx.data(), sizeof(int32_t) * t.nbr_idx.size());
  if (w) std::memcpy(w, t.nbr_w.data(), sizeof(double) * t.nbr_w.size());
  if (k0) std::memcpy(k0, t.k0.data(), sizeof(double) * t.n);
  if (en) std::memcpy(en, t.edge_nodes.data(), sizeof(int32_t) * 2 * t.E);
  if (eL) std::memcpy(eL, t.edge_L0.data(), sizeof(double) * t.E);
  if (med) *med = t.median_L;
  return DSH_OK;
}

int dsh_template_embed(const dsh_ctx* c, int P, const float* pts, int32_t* facet_id, int32_t* nodes, float* bary) {
  if (!c || !c->tmpl.valid || c->tmpl.F <= 0) return DSH_ERR_STATE;
  if (P < 0 || !pts || !facet_id || !nodes || !bary) return DSH_ERR_ARG;
  c->tmpl.embed(P, pts, facet_id, nodes, bary);
  return DSH_OK;
}

int dsh_template_embed_device(dsh_ctx* c, int P, const float* pts, int32_t* facet_id, int32_t* nodes, float* bary) {
  if (!c) return DSH_ERR_ARG;
  if (!c->tmpl.valid || c->tmpl.F <= 0) return fail(c, DSH_ERR_STATE, "dsh_template_embed_device: needs a template built from facets");
  if (P < 0 || (P > 0 && (!pts || !facet_id || !nodes || !bary))) return fail(c, DSH_ERR_ARG, "dsh_template_embed_device: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_template_embed_device: host-only context, no GPU (dsh_template_embed is the host routine)");
  if (P == 0) return DSH_OK;
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, DSH_ERR_HIP, "dsh_template_embed_device: hipSetDevice failed");
  const dsh::TemplateHost& t = c->tmpl;
  c->scratch.reset();
  hipStream_t st = c->stream;
  struct Item { const void* src; size_t bytes; void* dev; };
  Item in[5] = {{pts, 12 * (size_t)P, nullptr}, {t.xyz0.data(), 24 * (size_t)t.n, nullptr}, {t.facets.data(), 12 * (size_t)t.F, nullptr},
                {t.nf_ptr.data(), 4 * (size_t)(t.n + 1), nullptr}, {t.nf_idx.data(), 4 * t.nf_idx.size(), nullptr}};
  for (Item& it : in) {
    if (c->scratch.take(it.bytes, &it.dev) != hipSuccess) return fail(c, DSH_ERR_HIP, "dsh_template_embed_device: out of device memory");
    if (it.bytes && hipMemcpyAsync(it.dev, it.src, it.bytes, hipMemcpyHostToDevice, st) != hipSuccess)
      return fail(c, DSH_ERR_HIP, "dsh_template_embed_device: upload failed");
  }
  void *d_fid = nullptr, *d_nodes = nullptr, *d_bary = nullptr;
  if (c->scratch.take(4 * (size_t)P, &d_fid) != hipSuccess || c->scratch.take(12 * (size_t)P, &d_nodes) != hipSuccess ||
      c->scratch.take(12 * (size_t)P, &d_bary) != hipSuccess)
    return fail(c, DSH_ERR_HIP, "dsh_template_embed_device: out of device memory");
  hipError_t e = reg_embed(P, static_cast<const float*>(in[0].dev), t.n, static_cast<const double*>(in[1].dev), static_cast<const int32_t*>(in[2].dev),
                           static_cast<const int32_t*>(in[3].dev), static_cast<const int32_t*>(in[4].dev), static_cast<int32_t*>(d_fid),
                           static_cast<int32_t*>(d_nodes), static_cast<float*>(d_bary), st);
  if (e == hipSuccess) e = hipMemcpyAsync(facet_id, d_fid, 4 * (size_t)P, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(nodes, d_nodes, 12 * (size_t)P, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(bary, d_bary, 12 * (size_t)P, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(c, DSH_ERR_HIP, std::string("dsh_template_embed_device: ") + hipGetErrorString(e));
  return DSH_OK;
}

int dsh_sft_batch_upload(dsh_ctx* c, int B, const dsh_sft_frame* frames) {
  if (!c || B <= 0 || !frames) return fail(c, DSH_ERR_ARG, "dsh_sft_batch_upload: bad argument");
  if (!c->tmpl.valid) return fail(c, DSH_ERR_STATE, "dsh_sft_batch_upload: no template");
  if (!c->host_only) (void)hipSetDevice(c->device);
  c->B = 0;
  c->ran = false;
  c->upload_serial++;
  if ((int)c->packed.size() != B) c->packed.resize(B);   // the vectors inside keep their capacity from frame to frame
  for (int b = 0; b < B; b++) {
    std::string e;
    const int rc = pack_problem(c, frames[b], c->opt.wide_off != 0, c->packed[b], e);
    if (rc != DSH_OK) return fail(c, rc, "problem " + std::to_string(b) + ": " + e);
  }
  // Launch shape: 8 wavefronts per problem give the lowest latency; with at least two problems per CU, 4 wavefronts
  // per problem (two problems resident per CU, <= 80 KB of LDS each) give the higher throughput.  Band mode needs 8.
  int nw = 8;
  {
    bool all_tiles = true;
    for (int b = 0; b < B; b++) all_tiles = all_tiles && c->packed[b].h.tile_mode == 1;
    // More problems than the latency mode takes (half a problem per CU): the throughput shape.  From two problems per CU upwards that is rounds of
    // phase kernels + the tail kernel; below, the tail threshold (run_rounds_enqueue) covers the whole batch and the step is the tail kernel
    // alone -- one persistent workgroup per CU pulling problems, with the LIN kernel's record placement: 12 against 14 ms per problem for the
    // one-workgroup-per-problem kernel that ran these sizes until r06 (tools/batch_curve.py: 256 problems 13.97 -> 11.9 ms per step)
    if (all_tiles && 2 * B > c->num_cus) nw = 4;
    if ((c->opt.waves == 4 && all_tiles) || c->opt.waves == 8) nw = c->opt.waves;   // lab builds only (dsh_lab_set_option)
    if (c->force_waves == 8) nw = 8;
    // From two problems per CU upwards the batch runs as rounds of phase kernels with one wavefront per factorisation (sft_batch.h)
    c->rounds_mode = all_tiles && nw == 4 && c->opt.waves == 0 && c->opt.rounds != 0 && !c->host_only;
    // sub-batches: each must still fill the device with factor waves (one per SIMD) several times over
    c->n_sub = 1;
    if (c->rounds_mode) {
      // (measured on MI355X, tools/streams_ab.py, 16384 C2 problems: 416 / 428 / 424 / 423 ms per step for 1 / 2 / 3 / 4 sub-batches -- what the
      // overlapped tails win, the additional launches and last-problem back substitutions lose again: one sub-batch unless asked otherwise)
      const int want = c->opt.streams > 0 ? c->opt.streams : 1;
      while (c->n_sub < want && c->n_sub < dsh_ctx::kMaxSub && B / (c->n_sub + 1) >= 16 * c->num_cus) c->n_sub++;
      if (c->opt.streams > 0) c->n_sub = std::min(std::min(c->opt.streams, (int)dsh_ctx::kMaxSub), std::max(1, B / 64));
      for (int i = 1; i < c->n_sub; i++)   // (created on first use)
        if (!c->sub_stream[i] && hipStreamCreateWithFlags(&c->sub_stream[i], hipStreamNonBlocking) != hipSuccess) c->sub_stream[i] = nullptr;
      for (int i = 0; i < c->n_sub; i++) if (!c->sub_stream[i] || !c->sub_event[0]) c->n_sub = 1;
    }
  }
  // Latency mode: while CUs would idle anyway, every problem gets K of them and tries K dampings per iteration at once.
  int K = 1;
  if (nw == 8 && !c->force_waves && !c->host_only) {
    K = (4 * B <= c->num_cus) ? 4 : ((3 * B <= c->num_cus) ? 3 : ((2 * B <= c->num_cus) ? 2 : 1));   // as many lanes as the device holds at once
    // Wide bands (two-sided factorisation with helper workgroups, sft_wide.h): a part's helpers are worth more than the third and fourth lane when
    // the device cannot hold both -- two lanes with three helpers per part against four lanes without (C5 x 16: 47.2 against 51.0 ms per step)
    // (the predicate is the one that grants helpers below: the problem IS cut in two parts -- wide tile mode, room for two parts of four tile
    // columns -- and its band has at least helpers_wbt tiles; a wide band that stays undivided keeps its four lanes)
    {
      bool wide = c->opt.split != 0 && c->opt.helpers != 0;
      for (int b = 0; b < B; b++) {
        const SftDev& hh = c->packed[b].h;
        const int sT = (hh.kd + kTS - 1) / kTS, sp = kTS * sT;
        const int c0 = ((hh.Dn - sp) / 2 / kTS) * kTS, n1 = hh.Dn - sp - c0;
        wide = wide && hh.tile_mode == 2 && sT >= c->opt.helpers_wbt && sT >= 2 && c0 >= 4 * kTS && n1 >= 4 * kTS;
      }
      if (wide && K == 4 && (long long)B * 4 * 2 * 3 > c->num_cus && (long long)B * 2 * 2 * 3 <= c->num_cus) K = 2;
    }
    if (c->opt.speculate >= 1 && c->opt.speculate <= SFT_SPEC_MAXK) K = c->opt.speculate;   // lab builds only
    for (int b = 0; b < B; b++) if (c->packed[b].f.max_iters < 1) K = 1;
  }
  // LDS of the assembly (it aliases the solver workspace): the records a gather touches most often, as far as the budget goes
  // (4 wavefronts: two problems share a CU's 160 KB)
  size_t jl_doubles = 0, xyz_doubles = 0;
  int max_kd = 0;
  // (rounds of phase kernels: the LIN kernel is the only one that stages records, eight wavefronts and one workgroup per CU -- sft_batch.h)
  const size_t lds_budget = ((((nw == 4 && !c->rounds_mode) || SFT_WAVES_PER_EU >= 4) ? 75 : 155) * 1024) / 8;   // doubles, next to ~4.3 KB of control block and reduction scratch
  for (int b = 0; b < B; b++) {
    SftDev& hh = c->packed[b].h;
    // A narrow band (kd <= 128) that is long enough for two parts also takes the two-sided factorisation in latency mode: it runs on the
    // left-looking wide-tile code (tile mode 2 works for any half-bandwidth up to 256), two workgroups per damping trial instead of one
    // (C2: 4.1 ms per frame against 4.5 on the register-window solver -- the default since the SOLVE launch split the back substitutions).
    // Only while the launch is small: measured on C2, 4 lanes (tools/latency_batch_ab.py), the two-sided path wins up to 12 problems per launch
    // (4.09 against 4.49 ms for one, 5.97 against 6.15 for twelve) and loses from 16 on (6.31 against 6.18; 48 problems: 11.0 against 7.4) --
    // a wide band gains at every size (C5: 33 against 62 ms for one problem, 89 against 116 for 64).
    if ((c->force_split && hh.tile_mode == 1) ||
        (K > 1 && hh.tile_mode == 1 && c->opt.split >= 2 && 20 * B <= c->num_cus && hh.kd > kTS && hh.Dn >= 8 * kTS * ((hh.kd + kTS - 1) / kTS))) {
      // (kd > kTS: a band of one tile has no separator of two tile columns -- it would run the wide-tile code on one workgroup for nothing)
      hh.tile_mode = 2;
      hh.wbt = (hh.kd + kTS - 1) / kTS;
      hh.tpr = hh.wbt + 1;
    }
    hh.mode = (hh.mode & ~2) | ((hh.tile_mode == 1 && c->opt.dataflow) ? 2 : 0);   // the barrier version of the factor steps exists in lab builds only
    // Two-sided factorisation (SftPart in sft_problem.h): in latency mode a wide-band problem is cut at a separator of one bandwidth and
    // its two halves are factored by two workgroups at the same time.  Needs room for two parts of at least four tile columns.
    hh.split = 0;
    if ((c->force_split || (K > 1 && c->opt.split)) && hh.tile_mode == 2) {
      const int sT = hh.wbt, sp = kTS * sT;
      const int c0 = ((hh.Dn - sp) / 2 / kTS) * kTS, n1 = hh.Dn - sp - c0;
      if (sT >= 2 && c0 >= 4 * kTS && n1 >= 4 * kTS) {
        const int n1p = ((n1 + kTS - 1) / kTS) * kTS;
        hh.split = 1; hh.sp_c0 = c0; hh.sp_s = sp; hh.sp_n1p = n1p; hh.sp_pad = n1p - n1;
        SftPart& p0 = hh.part[0]; SftPart& p1 = hh.part[1]; SftPart& p2 = hh.part[2];
        p0 = SftPart{}; p1 = SftPart{}; p2 = SftPart{};
        p0.nS = c0 / kTS; p0.nT = p0.nS + sT; p0.tpr = hh.tpr; p0.wbt = hh.wbt; p0.b_base = 0; p0.b_sign = 1; p0.b_lo = 0; p0.b_hi = c0 + sp;
        p1.nS = n1p / kTS; p1.nT = p1.nS + sT; p1.tpr = hh.tpr; p1.wbt = hh.wbt; p1.b_base = hh.Dn - 1 + hh.sp_pad; p1.b_sign = -1; p1.b_lo = hh.sp_pad; p1.b_hi = n1p;
        p2.nS = sT; p2.nT = sT; p2.wbt = sT - 1; p2.tpr = sT;
        hh.sp_xl = sT * p2.tpr * kTS * kTS + 8 * kTS * sT + 64;
        hh.part[3] = p2;   // second workspace of the reduced problem (SFT_SPEC_SOLVE: one per workgroup)
      }
    }
    size_t used = 0;
    // placement class of the records (sft_kernels.hip: AsmRec): 1 = node positions + observation weights + curvature records, 2 = + node matrices + stretch records
    const size_t need1 = ((3 * (size_t)hh.n + 1) & ~(size_t)1) + (((size_t)hh.M + 1) & ~(size_t)1) + 4 * (size_t)hh.S, need2 = need1 + 6 * (size_t)hh.nA + 4 * (size_t)hh.Es;
    const size_t need3 = need2 + 5 * (size_t)hh.M;   // + the camera records as five doubles (only the LIN kernel of the phase rounds has the code)
    hh.lds_class = (c->rounds_mode && used + need3 <= lds_budget) ? 3 : (used + need2 <= lds_budget) ? 2 : ((used + need1 <= lds_budget) ? 1 : 0);
    used += hh.lds_class == 3 ? need3 : hh.lds_class == 2 ? need2 : (hh.lds_class == 1 ? need1 : 0);
    jl_doubles = std::max(jl_doubles, used);
    if (hh.lds_class >= 1) xyz_doubles = std::max(xyz_doubles, ((3 * (size_t)hh.n + 1) & ~(size_t)1));   // (sftb_trial_kernel stages the positions of exactly these)
    max_kd = std::max(max_kd, hh.tile_mode == 2 ? std::max(hh.kd, kTS * kBT + 1) : hh.kd);   // (LDS of the wide-tile solver whenever a problem runs on it)
  }
  if (c->host_only) {  // packed on the host only; dsh_sft_batch_problem_info works, running does not
    c->h_probs.resize(B);
    for (int b = 0; b < B; b++) c->h_probs[b] = c->packed[b].h;
    c->B = B;
    c->nw = nw;
    return DSH_OK;
  }
  // ---- layout: [SftDev table][per-frame read-only arrays of every problem] | [result region: B headers, bodies] | [workspace]
  Arena a;
  const size_t o_tab = a.take(sizeof(SftDev) * B * K);   // lane-major: lane 0 of every problem first
  struct Offs { size_t obs_nodes, obs_bary, obs_uv, obs_w, ob_ptr, ob_m, ob_c, viewed, xyz_init, pose_init; };
  std::vector<Offs> ro(B);
  for (int b = 0; b < B; b++) {
    dsh::SftFramePack& F = c->packed[b].f;
    Offs& o = ro[b];
    o.obs_nodes = reserve(a, F.obs_nodes); o.obs_bary = reserve(a, F.obs_bary); o.obs_uv = reserve(a, F.obs_uv); o.obs_w = reserve(a, F.obs_w);
    o.ob_ptr = reserve(a, F.ob_ptr); o.ob_m = reserve(a, F.ob_m); o.ob_c = reserve(a, F.ob_c); o.viewed = reserve(a, F.viewed);
    o.xyz_init = reserve(a, F.xyz_init);
    o.pose_init = a.take(8 * 8);
  }
  c->ro_bytes = a.size;
  // result region: every header first (dsh_sft_batch_counts reads only them), then the bodies
  c->res_off = a.size;
  (void)a.take(sizeof(SftResHdr) * (size_t)B);
  c->res_offs.resize(B);
  for (int b = 0; b < B; b++) {
    const SftDev& h = c->packed[b].h;
    dsh_ctx::ResOffs& r = c->res_offs[b];
    r.xyz = a.take(8 * 3 * (size_t)h.n) - c->res_off; r.chi2 = a.take(8 * (size_t)h.M) - c->res_off;
    r.trace = a.take(8 * DSH_TRACE_STRIDE * DSH_MAX_ITERS) - c->res_off;
    r.mp = a.take(4 * 3 * (size_t)h.M) - c->res_off; r.outl = a.take((size_t)h.M) - c->res_off;
  }
  c->res_bytes = a.size - c->res_off;
  struct POffs { size_t Hb, Lb, Lt, LbT, Lbord, Linv, x, xchg, Pf, PfB, sync; };
  struct WOffs { size_t bak, camrec, wtv, Anode, Jstar, Jstr, Hc, Hb, Hbord, Hcn, Lb, Lbord, Lc, Linv, Lt, LbT, x, dbg, sx0, sx1, shadow_xyz, shadow_chi2, shadow_hdr; POffs part[4]; };
  std::vector<WOffs> wo((size_t)B * K);
  const size_t ws_off = a.size;
  const size_t o_spec = a.take(sizeof(SftSpec) * (size_t)B * K);
  const size_t o_runs = a.take(c->rounds_mode ? sizeof(SftRun) * (size_t)B + 64 * dsh_ctx::kMaxSub + sizeof(int) * (size_t)B : 0);
  // Helper workgroups of the two-sided factorisation: while CUs idle anyway, every part gets nh more of them for the far products of its block
  // columns (sft_wide.h).  Everything has to be resident at once for that to pay, so nh is what the device holds: B * K * 2 * (1 + nh) <= CUs.
  int nh = 0;
  {
    // (only where the far products are most of a block column: bands of at least 12 tiles; a narrow band through this path gains nothing --
    // C2, 8 tiles: 4.4 against 4.0 ms per frame with helpers)
    bool any = false;
    for (int b = 0; b < B; b++) any = any || (c->packed[b].h.split != 0 && c->packed[b].h.wbt >= c->opt.helpers_wbt);
    for (int b = 0; b < B; b++)   // (the owner's progress word keeps the finished block columns in 16 bits)
      if (c->packed[b].h.split && std::max(c->packed[b].h.part[0].nT, c->packed[b].h.part[1].nT) >= 60000) any = false;
    if (any && K > 1 && !c->force_split) {
      while (nh < 3 && (long long)B * K * 2 * (2 + nh) <= c->num_cus) nh++;
      if (nh == 1) nh = 0;                            // (one helper cannot feed its owner: 8.5 against 7 us per block column -- measured slower than none)
      if (c->opt.helpers >= 0) nh = c->opt.helpers;   // lab builds only
    }
  }
  // (one block: a run clears it with one memset)
  size_t sync_total = 0;
  if (nh > 0)
    for (int e = 0; e < B * K; e++) {
      const SftDev& h = c->packed[e % B].h;
      if (h.split) for (int g = 0; g < 2; g++) sync_total += ((size_t)4 * (16 + h.part[g].nT) + 255) & ~(size_t)255;
    }
  const size_t o_sync = a.take(sync_total);
  size_t sync_used = 0;
  for (int e = 0; e < B * K; e++) {
    const int b = e % B, lane = e / B;
    const SftDev& h = c->packed[b].h;
    const size_t Dnp = (size_t)((h.Dn + kNB - 1) / kNB) * kNB;
    WOffs& w = wo[e];
    w.sx0 = a.take(K > 1 ? 8 * 3 * (size_t)h.n : 0); w.sx1 = a.take(K > 1 ? 8 * 3 * (size_t)h.n : 0);
    // lanes > 0 keep their state, errors and pose in the workspace: only lane 0 owns a slot of the result region
    w.shadow_xyz = a.take(lane ? 8 * 3 * (size_t)h.n : 0); w.shadow_chi2 = a.take(lane ? 8 * (size_t)h.M : 0); w.shadow_hdr = a.take(lane ? sizeof(SftResHdr) : 0);
    w.bak = a.take(8 * 3 * (size_t)h.n);
    w.camrec = a.take(8 * (size_t)h.M * SFT_CAM_STRIDE);
    w.wtv = a.take(h.lds_class >= 1 ? 0 : 8 * ((size_t)h.M + 1)); w.Jstar = a.take(h.lds_class >= 1 ? 0 : 8 * 4 * (size_t)h.S);
    w.Anode = a.take(h.lds_class >= 2 ? 0 : 8 * 6 * (size_t)h.nA); w.Jstr = a.take(h.lds_class >= 2 ? 0 : 8 * 4 * (size_t)h.Es);
    // tile mode: BT+1 zero tile rows below the matrix and an 8th (zero) border row + one window of columns let the
    // factorisation load every tile of its sliding window unconditionally (SFT_H_PAD_* in sft_problem.h)
    const size_t band_elems = h.tile_mode ? (Dnp / kTS + SFT_H_PAD_TILE_ROWS) * (size_t)h.tpr * kTS * kTS : Dnp * (size_t)h.ldh;
    const size_t bord_elems = (SFT_BORDER + 1) * Dnp + SFT_H_PAD_BORDER;
    w.Hc = a.take(h.tile_mode == 1 ? 8 * c->packed[b].g->hc_elems() : 0);
    w.Hb = a.take(h.tile_mode == 1 ? 0 : 8 * band_elems); w.Hbord = a.take(8 * bord_elems); w.Hcn = a.take(8 * 56);
    w.Lb = a.take(8 * band_elems); w.Lbord = a.take(8 * bord_elems); w.Lc = a.take(8 * 56);
    w.Linv = a.take(8 * (Dnp / kTS) * (size_t)kTS * kTS);
    w.Lt = a.take(h.tile_mode == 2 ? 8 * band_elems : 0); w.LbT = a.take(h.tile_mode == 2 ? 8 * (Dnp / kTS) * (size_t)kTS * kTS : 0);
    w.x = a.take(8 * (Dnp + 8)); w.dbg = a.take(1024);
    if (h.split)
      for (int g = 0; g < 4; g++) {   // the band matrices of the two parts and of the reduced problem, twice (H of a part: one copy, lane 0's, shared by the lanes)
        const SftPart& q = h.part[g];
        const size_t tiles = 8 * (size_t)q.nT * q.tpr * kTS * kTS, col = 8 * (size_t)q.nT * kTS * kTS;
        POffs& po = w.part[g];
        po.Hb = a.take(g < 2 && lane == 0 ? tiles : 0);
        po.Lb = a.take(tiles); po.Lt = a.take(tiles); po.LbT = a.take(col); po.Linv = a.take(col);
        po.Lbord = a.take(8 * 8 * (size_t)kTS * q.nT); po.x = a.take(8 * ((size_t)kTS * q.nT + 8)); po.xchg = a.take(8 * (size_t)h.sp_xl);
        const bool helped = nh > 0 && g < 2;
        po.Pf = a.take(helped ? tiles : 0); po.PfB = a.take(helped ? col : 0);
        po.sync = o_sync + sync_used;
        if (helped) sync_used += ((size_t)4 * (16 + q.nT) + 255) & ~(size_t)255;
      }
  }
  if (sft_lm_kernel_lds_bytes(max_kd, jl_doubles) > 160 * 1024 || max_kd + kNB + SFT_BORDER > SFT_NT)
    return fail(c, DSH_ERR_ARG, "half-bandwidth too large for the LDS panel / workgroup");
  bool fresh_arena = false;
  if (a.size > c->d_batch_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_batch) { (void)hipFree(c->d_batch); c->d_batch = nullptr; c->d_batch_cap = 0; }
    HIPCHK(c, hipMalloc((void**)&c->d_batch, a.size));
    c->d_batch_cap = a.size;
    fresh_arena = true;
  }
  // the staging buffer of the previous upload may still be read by its copy
  if (c->stage_busy) { HIPCHK(c, hipEventSynchronize(c->stage_free)); c->stage_busy = false; }
  HIPCHK(c, c->stage.ensure(c->ro_bytes, true));
  char* st = c->stage.p;
  char* base = c->d_batch;
  c->h_probs.resize((size_t)B * K);
  SftResHdr* d_hdr = (SftResHdr*)(base + c->res_off);
  int max_iters = 0;
  for (int e = 0; e < B * K; e++) {
    const int b = e % B, lane = e / B;
    Packed& P = c->packed[b];
    const dsh::SftFramePack& F = P.f;
    const dsh::SftGraph& g = *P.g;
    const Offs& o = ro[b];
    if (lane == 0) {
      put(st, o.obs_nodes, F.obs_nodes); put(st, o.obs_bary, F.obs_bary); put(st, o.obs_uv, F.obs_uv); put(st, o.obs_w, F.obs_w);
      put(st, o.ob_ptr, F.ob_ptr); put(st, o.ob_m, F.ob_m); put(st, o.ob_c, F.ob_c); put(st, o.viewed, F.viewed); put(st, o.xyz_init, F.xyz_init);
      std::memcpy(st + o.pose_init, F.pose_init, 7 * sizeof(double));
    }
    SftDev h = P.h;
    const WOffs& w = wo[e];
    const dsh_ctx::ResOffs& r = c->res_offs[b];
    max_iters = std::max(max_iters, F.max_iters);
    char* rbase = base + c->res_off;
    const char* gb = g.d_base;
    h.xyz0 = c->dt.xyz0; h.nbr_ptr = c->dt.nbr_ptr; h.nbr_idx = c->dt.nbr_idx; h.nbr_w = c->dt.nbr_w; h.nbr_sumw = c->dt.nbr_sumw; h.k0 = c->dt.k0;
    h.act = (const int32_t*)(gb + g.o.act); h.actnode = (const int32_t*)(gb + g.o.actnode); h.star_node = (const int32_t*)(gb + g.o.star_node);
    h.star_sL = (const double*)(gb + g.o.star_sL); h.str_nodes = (const int32_t*)(gb + g.o.str_nodes); h.str_L0 = (const double*)(gb + g.o.str_L0);
    h.off_ptr = (const int32_t*)(gb + g.o.off_ptr); h.off_rc = (const int32_t*)(gb + g.o.off_rc); h.sh_ptr = (const int32_t*)(gb + g.o.sh_ptr);
    h.sh_rec = (const uint32_t*)(gb + g.o.sh_rec); h.sh_cf = (const double*)(gb + g.o.sh_cf); h.tmask = (const int32_t*)(gb + g.o.tmask);
    h.hgather = (const uint32_t*)(gb + g.o.hgather);
    h.hgatherT = (const uint32_t*)(gb + g.o.hgatherT);
    h.obs_nodes = (const int32_t*)(base + o.obs_nodes); h.obs_bary = (const double*)(base + o.obs_bary);
    h.obs_uv = (const double*)(base + o.obs_uv); h.obs_w = (const double*)(base + o.obs_w);
    h.ob_ptr = (const int32_t*)(base + o.ob_ptr); h.ob_m = (const int32_t*)(base + o.ob_m); h.ob_c = (const double*)(base + o.ob_c);
    h.viewed = (const uint8_t*)(base + o.viewed);
    h.xyz_init = (const double*)(base + o.xyz_init); h.pose_init = (const double*)(base + o.pose_init);
    h.res = d_hdr + b; h.pose = d_hdr[b].pose;   // address arithmetic on a device pointer: nothing is dereferenced on the host
    h.xyz = (double*)(rbase + r.xyz); h.chi2_obs = (double*)(rbase + r.chi2); h.trace = (double*)(rbase + r.trace);
    h.mappoint = (float*)(rbase + r.mp); h.outlier = (uint8_t*)(rbase + r.outl);
    h.xyz_bak = (double*)(base + w.bak);
    h.camrec = (double*)(base + w.camrec); h.wtv = (double*)(base + w.wtv); h.Anode = (double*)(base + w.Anode);
    h.Jstar = (double*)(base + w.Jstar); h.Jstr = (double*)(base + w.Jstr);
    h.Hc = (double*)(base + w.Hc); h.Hb = (double*)(base + w.Hb); h.Hbord = (double*)(base + w.Hbord); h.Hcorner = (double*)(base + w.Hcn);
    h.Lb = (double*)(base + w.Lb); h.Lbord = (double*)(base + w.Lbord); h.Lcorner = (double*)(base + w.Lc); h.Linv = (double*)(base + w.Linv);
    h.Lt = (double*)(base + w.Lt); h.LbT = (double*)(base + w.LbT);
    h.x = (double*)(base + w.x); h.dbg = (double*)(base + w.dbg);
    h.spec_xyz[0] = (double*)(base + w.sx0); h.spec_xyz[1] = (double*)(base + w.sx1);
    if (h.split)
      for (int g = 0; g < 4; g++) {
        const POffs& po = w.part[g];
        SftPart& q = h.part[g];
        q.Hb = g < 2 ? (double*)(base + (lane ? wo[b].part[g].Hb : po.Hb)) : (double*)(base + po.xchg);   // reduced problem: H = the summed exchange buffer
        q.Lb = (double*)(base + po.Lb); q.Lt = (double*)(base + po.Lt); q.LbT = (double*)(base + po.LbT); q.Linv = (double*)(base + po.Linv);
        q.Lbord = (double*)(base + po.Lbord); q.x = (double*)(base + po.x); q.xchg = (double*)(base + po.xchg);
        const bool helped = nh > 0 && g < 2;
        q.Pf = helped ? (double*)(base + po.Pf) : nullptr; q.PfB = helped ? (double*)(base + po.PfB) : nullptr;
        q.sync = helped ? (int32_t*)(base + po.sync) : nullptr;
      }
    if (lane) {
      h.xyz = (double*)(base + w.shadow_xyz); h.chi2_obs = (double*)(base + w.shadow_chi2);
      h.res = (SftResHdr*)(base + w.shadow_hdr); h.pose = ((SftResHdr*)(base + w.shadow_hdr))->pose;
      h.trace = nullptr; h.mappoint = nullptr; h.outlier = nullptr;
      if (h.tile_mode == 1 || h.split) {   // the lanes of a problem assemble one H together (each its share of the block rows) and all factor from it
        const SftDev& h0 = c->h_probs[b];
        h.Hc = h0.Hc; h.Hbord = h0.Hbord; h.Hcorner = h0.Hcorner;
      }
    }
    c->h_probs[e] = h;
  }
  std::memcpy(st + o_tab, c->h_probs.data(), sizeof(SftDev) * B * K);
  HIPCHK(c, hipMemcpyAsync(base, st, c->ro_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->stage_free, c->stream));
  c->stage_busy = true;
  // The kernel initialises everything it reads (state, H with its zero padding, border, x, counters).  A fresh allocation
  // is cleared once so that no tile of L / Lt that a masked lane may touch holds a NaN pattern; after that the workspace
  // only ever holds finite numbers this library wrote.  The result region is always cleared (a caller that downloads
  // without running gets zeros, not the previous batch).
  if (fresh_arena) HIPCHK(c, hipMemsetAsync(base + ws_off, 0, a.size - ws_off, c->stream));
  HIPCHK(c, hipMemsetAsync(base + c->res_off, 0, c->res_bytes, c->stream));
  c->d_probs = (SftDev*)(base + o_tab);
  c->d_spec = (SftSpec*)(base + o_spec);
  c->d_runs = (SftRun*)(base + o_runs);
  c->d_counters = (int*)(base + o_runs + sizeof(SftRun) * (size_t)B);
  c->d_linlist = c->d_counters + 16 * dsh_ctx::kMaxSub;
  c->spec_bytes = sizeof(SftSpec) * (size_t)B * K;
  c->spec_k = K;
  c->spec_nh = nh;
  c->d_sync = base + o_sync;
  c->sync_bytes = sync_total;
  c->any_split = false;
  for (int b = 0; b < B; b++) c->any_split = c->any_split || c->packed[b].h.split != 0;
  c->max_iters_batch = max_iters;
  c->B = B;
  c->max_kd = max_kd;
  c->jl_doubles = jl_doubles;
  c->xyz_doubles = xyz_doubles;
  c->nw = nw;
  return DSH_OK;   // asynchronous: the launch of dsh_sft_batch_run is ordered behind the copy on the same stream (dsh_sft_batch_run itself BLOCKS in the
                   // latency mode and in the rounds of phase kernels: it reads done flags / counters back between groups of launches)
}

int dsh_sft_batch_run(dsh_ctx* c) {
  if (!c) return DSH_ERR_ARG;
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_sft_batch_run: host-only context, no GPU (there is no CPU fallback)");
  if (c->B <= 0) return fail(c, DSH_ERR_STATE, "dsh_sft_batch_run: nothing uploaded");
  (void)hipSetDevice(c->device);
  const int rc = run_once(c);
  if (rc != DSH_OK) return rc;
  c->ran = true;
  return DSH_OK;
}

int dsh_sft_batch_counts(dsh_ctx* c, int64_t* iters, int64_t* trials) {
  if (!c || c->B <= 0 || !c->ran) return DSH_ERR_STATE;
  if (c->host_only) return DSH_ERR_NO_DEVICE;
  (void)hipSetDevice(c->device);
  // one copy of the B result headers (they are contiguous), ordered behind the run on the context's stream
  const size_t bytes = sizeof(SftResHdr) * (size_t)c->B;
  HIPCHK(c, c->results.ensure(std::max(bytes, c->results.cap), true));
  HIPCHK(c, hipMemcpyAsync(c->results.p, c->d_batch + c->res_off, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const SftResHdr* hd = reinterpret_cast<const SftResHdr*>(c->results.p);
  int64_t it = 0, tr = 0;
  for (int b = 0; b < c->B; b++) { it += hd[b].iters; tr += hd[b].trials; }
  if (iters) *iters = it;
  if (trials) *trials = tr;
  return DSH_OK;
}

int dsh_sft_batch_problem_info(dsh_ctx* c, int b, int64_t* bytes, int32_t* counts) {
  if (!c || b < 0 || b >= c->B) return DSH_ERR_ARG;
  const Packed& P = c->packed[b];
  const SftDev& h = P.h;
  // materialised-Jacobian convention, reference edge counts (curvature unfused)
  const int64_t M = h.M, n = h.n, C = P.g->n_curv_ref, E = h.Es, V = h.V;
  const int64_t reads = 60 * M + 24 * n + 88 + 92 * C + 16 * E + 28 * V;
  const int64_t writes = 8 * (30 * M + 21 * C + 6 * E + 9 * V) + 8 * (2 * M + C + E + 3 * V) + 8 * M;
  if (bytes) *bytes = reads + writes;
  if (counts) { counts[0] = h.M; counts[1] = h.nA; counts[2] = P.g->n_curv_ref; counts[3] = h.Es; counts[4] = h.V; counts[5] = 6 + h.Dn; counts[6] = h.kd; counts[7] = c->rounds_mode ? 1 : c->nw; counts[8] = P.g->noff; }
  return DSH_OK;
}

int dsh_sft_batch_download(dsh_ctx* c, int B, dsh_sft_result* res) {
  if (!c || !res || B != c->B) return fail(c, DSH_ERR_ARG, "dsh_sft_batch_download: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_sft_batch_download: host-only context");
  if (!c->ran) return fail(c, DSH_ERR_STATE, "dsh_sft_batch_download: no run");
  (void)hipSetDevice(c->device);
  // the whole result region (headers + bodies of every problem) in ONE copy into page-locked memory
  HIPCHK(c, c->results.ensure(c->res_bytes, true));
  HIPCHK(c, hipMemcpyAsync(c->results.p, c->d_batch + c->res_off, c->res_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const char* rb = c->results.p;
  const SftResHdr* hd = reinterpret_cast<const SftResHdr*>(rb);
  for (int b = 0; b < B; b++) {
    const SftDev& h = c->h_probs[b];
    const Packed& P = c->packed[b];
    const dsh_ctx::ResOffs& o = c->res_offs[b];
    dsh_sft_result& r = res[b];
    r.rep_error = hd[b].rep_error;
    r.inliers = hd[b].inliers;
    r.iters = hd[b].iters;
    r.trials = hd[b].trials;
    r.status = hd[b].status;
    r.dim = 6 + h.Dn;
    r.half_bandwidth = h.kd;
    if (r.chi2_obs) std::memcpy(r.chi2_obs, rb + o.chi2, 8 * (size_t)h.M);
    if (r.outlier) std::memcpy(r.outlier, rb + o.outl, (size_t)h.M);
    if (r.xyz) std::memcpy(r.xyz, rb + o.xyz, 8 * 3 * (size_t)h.n);
    if (r.pose7) std::memcpy(r.pose7, hd[b].pose, 8 * 7);
    if (r.Tcw) Tcw_from_pose7(hd[b].pose, r.Tcw);
    if (r.mappoint_xyz) std::memcpy(r.mappoint_xyz, rb + o.mp, 4 * 3 * (size_t)h.M);
    if (r.trace) {   // rows of the executed iterations, zeros behind them
      const size_t rows = (size_t)std::max(P.f.max_iters, 0), done = std::min(rows, (size_t)std::max(hd[b].iters, 0));
      std::memcpy(r.trace, rb + o.trace, 8 * DSH_TRACE_STRIDE * done);
      std::memset(r.trace + DSH_TRACE_STRIDE * done, 0, 8 * DSH_TRACE_STRIDE * (rows - done));
    }
  }
  return DSH_OK;
}

int dsh_sft_solve(dsh_ctx* c, const dsh_sft_frame* frame, dsh_sft_result* result) {
  if (!c || !frame || !result) return fail(c, DSH_ERR_ARG, "dsh_sft_solve: bad argument");
  int rc = dsh_sft_batch_upload(c, 1, frame);
  if (rc != DSH_OK) return rc;
  rc = dsh_sft_batch_run(c);
  if (rc != DSH_OK) return rc;
  return dsh_sft_batch_download(c, 1, result);
}

// ---- shared-camera mode across GPUs -------------------------------------------------------------------------------------
// RCCL is bound at run time (dlopen): a process that never creates a communicator does not load it, and a host process that
// already carries an RCCL (PyTorch) keeps a single copy.
namespace {
struct RcclUniqueId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load(std::string& err) {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("RCCL not found: ") + dlerror(); return false; }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { err = "RCCL symbols missing"; lib = nullptr; return false; }
    return true;
  }
};
Rccl g_rccl;
constexpr int kNcclDouble = 8, kNcclSum = 0;   // ncclFloat64, ncclSum (rccl.h)
}  // namespace

struct dsh_comm {
  int nranks = 1, rank = 0;
  void* comm = nullptr;      // ncclComm_t
  dsh_ctx* ctx = nullptr;
};

namespace {

// One rank of a shared-camera solve as the driver sees it.
struct ScRank { dsh_ctx* c; };

// The all-reduce of the exchange vectors: RCCL between processes (one local rank), or a summation kernel between the
// contexts of an in-process group.
struct ScReducer {
  dsh_comm* comm = nullptr;            // RCCL
  SftSc** d_ptrs = nullptr;            // in-process group: device array of the ranks' state pointers
  int reduce(std::vector<ScRank>& R, std::string& err) {
    if (comm) {
      dsh_ctx* c = R[0].c;
      const int rc = g_rccl.AllReduce(c->d_sc->send, c->d_sc->recv, SFT_SC_XCHG, kNcclDouble, kNcclSum, comm->comm, c->stream);
      if (rc != 0) { err = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return DSH_ERR_HIP; }
      return DSH_OK;
    }
    for (auto& r : R)
      if (hipStreamSynchronize(r.c->stream) != hipSuccess) { err = "stream synchronise failed"; return DSH_ERR_HIP; }
    if (sft_sc_local_reduce(d_ptrs, (int)R.size(), R[0].c->stream) != hipSuccess || hipStreamSynchronize(R[0].c->stream) != hipSuccess) { err = "local reduce failed"; return DSH_ERR_HIP; }
    return DSH_OK;
  }
};

int sc_phase(std::vector<ScRank>& R, int phase, std::string& err) {
  for (auto& r : R) {
    dsh_ctx* c = r.c;
    (void)hipSetDevice(c->device);
    LDS_LOCK();
    if (sft_sc_launch(c->d_probs, c->d_sc, 1, phase, c->max_kd, c->jl_doubles, &LDS_MARKS(c).sc, c->stream) != hipSuccess) { err = "phase kernel launch failed"; return DSH_ERR_HIP; }
  }
  return DSH_OK;
}

// The Levenberg-Marquardt loop of the shared-camera mode: four phase kernels per damping trial, an all-reduce of SFT_SC_XCHG
// doubles behind LIN, FAC and SOL (sft_kernels.hip: sft_sc_kernel).  Every rank reads the same all-reduced numbers and takes
// the same decisions; the host only reads "again" / "done" of its first local rank.
int sc_solve(std::vector<ScRank>& R, ScReducer& red, int rank0, int nranks, const dsh_sft_frame* frames, dsh_sft_result* results, std::string& err) {
  const int G = (int)R.size();
  if (nranks > SFT_SC_XCHG - 13) { err = "too many ranks for the exchange vector"; return DSH_ERR_ARG; }   // (the same verdict on every rank)
  // A failure that only THIS rank sees (a bad frame, a template the mode cannot take, no memory) must not leave the other ranks inside a
  // collective: it is carried through the first all-reduce (slot 2 of the exchange vector) and every rank returns together.
  int local_rc = DSH_OK;
  std::string local_err;
  for (int g = 0; g < G; g++) {
    dsh_ctx* c = R[g].c;
    if (c->host_only) { err = "host-only context, no GPU (there is no CPU fallback)"; return DSH_ERR_NO_DEVICE; }
    (void)hipSetDevice(c->device);
    if (!c->d_sc && hipMalloc((void**)&c->d_sc, sizeof(SftSc)) != hipSuccess) { err = "out of device memory"; return DSH_ERR_HIP; }   // (nothing to exchange with)
    SftSc init{};
    init.rank = rank0 + g;
    init.nranks = nranks;
    int rc = frames[g].max_iters < 1 ? DSH_ERR_ARG : DSH_OK;
    if (rc != DSH_OK && local_rc == DSH_OK) { local_rc = rc; local_err = "max_iters must be >= 1"; }
    if (rc == DSH_OK) {
      c->force_waves = 8;
      rc = dsh_sft_batch_upload(c, 1, &frames[g]);
      c->force_waves = 0;
      if (rc != DSH_OK && local_rc == DSH_OK) { local_rc = rc; local_err = c->err; }
    }
    if (rc == DSH_OK && c->packed[0].h.tile_mode != 1) {
      rc = DSH_ERR_ARG;
      if (local_rc == DSH_OK) { local_rc = rc; local_err = "the shared-camera mode needs a template with half-bandwidth <= 128 (register-window solver); dsh_sft_connected_solve takes wider ones"; }
    }
    if (rc == DSH_OK) {
      init.send[0] = (double)c->packed[0].h.nA;     // the regulariser weights divide by the JOINT counts (DefOptimizer.cc:458,497)
      init.send[1] = (double)c->packed[0].h.Es;
    }
    init.send[2] = rc == DSH_OK ? 0.0 : 1.0;
    if (hipMemcpyAsync(c->d_sc, &init, sizeof(SftSc), hipMemcpyHostToDevice, c->stream) != hipSuccess) { err = "state upload failed"; return DSH_ERR_HIP; }
  }
  int rc = red.reduce(R, err);
  if (rc != DSH_OK) return rc;
  {
    double tot3[3];
    dsh_ctx* c = R[0].c;
    if (hipMemcpyAsync(tot3, c->d_sc->recv, sizeof(tot3), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { err = "read-back failed"; return DSH_ERR_HIP; }
    if (tot3[2] != 0.0) {   // some rank could not set its patch up: every rank leaves here
      if (local_rc != DSH_OK) { err = local_err; return local_rc; }
      err = "another rank failed to set its patch up";
      return DSH_ERR_STATE;
    }
  }
  for (int g = 0; g < G; g++) {   // joint counts -> weights of every rank's problem record
    dsh_ctx* c = R[g].c;
    double tot[2];
    if (hipMemcpyAsync(tot, c->d_sc->recv, sizeof(tot), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { err = "read-back failed"; return DSH_ERR_HIP; }
    SftDev& h = c->h_probs[0];
    h.w_curv = frames[g].reg_lap / tot[0];
    h.w_str = tot[1] > 0 ? frames[g].reg_inex / tot[1] : 0.0;
    if (hipMemcpyAsync(c->d_probs, &h, sizeof(SftDev), hipMemcpyHostToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { err = "problem record upload failed"; return DSH_ERR_HIP; }
  }
  for (int guard = 0; guard < DSH_MAX_ITERS + 1; guard++) {
    if ((rc = sc_phase(R, SFT_SC_LIN, err)) != DSH_OK || (rc = red.reduce(R, err)) != DSH_OK) return rc;
    int again = 0, done = 0;
    do {
      if ((rc = sc_phase(R, SFT_SC_FAC, err)) != DSH_OK || (rc = red.reduce(R, err)) != DSH_OK) return rc;
      if ((rc = sc_phase(R, SFT_SC_SOL, err)) != DSH_OK || (rc = red.reduce(R, err)) != DSH_OK) return rc;
      if ((rc = sc_phase(R, SFT_SC_CTL, err)) != DSH_OK) return rc;
      int32_t flags[2];
      dsh_ctx* c = R[0].c;
      if (hipMemcpyAsync(flags, &c->d_sc->again, sizeof(flags), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { err = "flag read-back failed"; return DSH_ERR_HIP; }
      again = flags[0];
      done = flags[1];
    } while (again);
    if (done) break;
  }
  for (int g = 0; g < G; g++) {
    R[g].c->ran = true;
    rc = dsh_sft_batch_download(R[g].c, 1, &results[g]);
    if (rc != DSH_OK) { err = R[g].c->err; return rc; }
  }
  return DSH_OK;
}

}  // namespace

int dsh_comm_unique_id(void* id) {
  std::string err;
  if (!id || !g_rccl.load(err)) return DSH_ERR_HIP;
  return g_rccl.GetUniqueId(static_cast<RcclUniqueId*>(id)) == 0 ? DSH_OK : DSH_ERR_HIP;
}

int dsh_comm_create(dsh_ctx* c, int nranks, int rank, const void* id, dsh_comm** out) {
  if (!c || !out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, DSH_ERR_ARG, "dsh_comm_create: bad argument");
  *out = nullptr;
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_comm_create: host-only context");
  std::string err;
  if (!g_rccl.load(err)) return fail(c, DSH_ERR_HIP, "dsh_comm_create: " + err);
  (void)hipSetDevice(c->device);
  RcclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  std::unique_ptr<dsh_comm> cm(new dsh_comm());
  cm->nranks = nranks; cm->rank = rank; cm->ctx = c;
  const int rc = g_rccl.CommInitRank(&cm->comm, nranks, uid, rank);
  if (rc != 0) return fail(c, DSH_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
  *out = cm.release();
  return DSH_OK;
}

int dsh_comm_destroy(dsh_comm* cm) {
  if (!cm) return DSH_ERR_ARG;
  if (cm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(cm->comm);
  delete cm;
  return DSH_OK;
}

int dsh_sft_shared_solve(dsh_ctx* c, dsh_comm* cm, const dsh_sft_frame* frame, dsh_sft_result* result) {
  if (!c || !cm || !frame || !result || cm->ctx != c) return fail(c, DSH_ERR_ARG, "dsh_sft_shared_solve: bad argument");
  std::vector<ScRank> R{ScRank{c}};
  ScReducer red;
  red.comm = cm;
  std::string err;
  const int rc = sc_solve(R, red, cm->rank, cm->nranks, frame, result, err);
  return rc == DSH_OK ? rc : fail(c, rc, "dsh_sft_shared_solve: " + err);
}

int dsh_sft_shared_solve_group(int G, dsh_ctx* const* ctxs, const dsh_sft_frame* frames, dsh_sft_result* results) {
  if (G < 1 || !ctxs || !frames || !results) return DSH_ERR_ARG;
  for (int g = 0; g < G; g++)
    if (!ctxs[g]) return DSH_ERR_ARG;
  dsh_ctx* c0 = ctxs[0];
  std::vector<ScRank> R;
  for (int g = 0; g < G; g++) R.push_back(ScRank{ctxs[g]});
  for (int g = 0; g < G; g++) {   // the state blocks must exist before their addresses are collected
    if (ctxs[g]->host_only) return fail(c0, DSH_ERR_NO_DEVICE, "dsh_sft_shared_solve_group: host-only context, no GPU (there is no CPU fallback)");
    (void)hipSetDevice(ctxs[g]->device);
    if (!ctxs[g]->d_sc && hipMalloc((void**)&ctxs[g]->d_sc, sizeof(SftSc)) != hipSuccess) return fail(c0, DSH_ERR_HIP, "dsh_sft_shared_solve_group: out of device memory");
  }
  std::vector<SftSc*> ptrs;
  for (int g = 0; g < G; g++) ptrs.push_back(ctxs[g]->d_sc);
  ScReducer red;
  (void)hipSetDevice(c0->device);
  if (hipMalloc((void**)&red.d_ptrs, sizeof(SftSc*) * G) != hipSuccess) return fail(c0, DSH_ERR_HIP, "dsh_sft_shared_solve_group: out of device memory");
  int rc = DSH_OK;
  std::string err;
  if (hipMemcpy(red.d_ptrs, ptrs.data(), sizeof(SftSc*) * G, hipMemcpyHostToDevice) != hipSuccess) { rc = DSH_ERR_HIP; err = "pointer table upload failed"; }
  if (rc == DSH_OK) rc = sc_solve(R, red, 0, G, frames, results, err);
  (void)hipFree(red.d_ptrs);
  return rc == DSH_OK ? rc : fail(c0, rc, "dsh_sft_shared_solve_group: " + err);
}


// ---- connected-mesh mode: one problem, one connected template, the factorisation cut in two (sft_kernels.hip: sft_cn_kernel) ----------
namespace {

// all-reduce (sum) of n doubles: send -> recv on every rank.  RCCL between two processes, a summation kernel between two contexts of one process.
int cn_allreduce(std::vector<ScRank>& R, dsh_comm* comm, double* const* send, double* const* recv, int n, std::string& err) {
  if (comm) {
    dsh_ctx* c = R[0].c;
    const int rc = g_rccl.AllReduce(send[0], recv[0], (size_t)n, kNcclDouble, kNcclSum, comm->comm, c->stream);
    if (rc != 0) { err = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return DSH_ERR_HIP; }
    return DSH_OK;
  }
  for (auto& r : R)
    if (hipStreamSynchronize(r.c->stream) != hipSuccess) { err = "stream synchronise failed"; return DSH_ERR_HIP; }
  if (sft_vec_sum2(send[0], send[1], recv[0], recv[1], n, R[0].c->stream) != hipSuccess || hipStreamSynchronize(R[0].c->stream) != hipSuccess) { err = "local reduce failed"; return DSH_ERR_HIP; }
  return DSH_OK;
}

int cn_phase(std::vector<ScRank>& R, int phase, std::string& err) {
  for (auto& r : R) {
    dsh_ctx* c = r.c;
    (void)hipSetDevice(c->device);
    LDS_LOCK();
    if (sft_cn_launch(c->d_probs, c->d_sc, phase, c->max_kd, c->jl_doubles, &LDS_MARKS(c).cn, c->stream) != hipSuccess) { err = "phase kernel launch failed"; return DSH_ERR_HIP; }
  }
  return DSH_OK;
}

// R: the local ranks (one with RCCL, two in the in-process group); every rank packs the SAME frame.
int cn_solve(std::vector<ScRank>& R, dsh_comm* comm, int rank0, const dsh_sft_frame* frame, dsh_sft_result* results, std::string& err) {
  const int G = (int)R.size();
  int local_rc = DSH_OK;
  std::string local_err;
  for (int g = 0; g < G; g++) {
    dsh_ctx* c = R[g].c;
    if (c->host_only) { err = "host-only context, no GPU (there is no CPU fallback)"; return DSH_ERR_NO_DEVICE; }
    (void)hipSetDevice(c->device);
    if (!c->d_sc && hipMalloc((void**)&c->d_sc, sizeof(SftSc)) != hipSuccess) { err = "out of device memory"; return DSH_ERR_HIP; }
    int rc = frame->max_iters < 1 ? DSH_ERR_ARG : DSH_OK;
    if (rc != DSH_OK && local_rc == DSH_OK) { local_rc = rc; local_err = "max_iters must be >= 1"; }
    if (rc == DSH_OK) {
      c->force_waves = 8;
      c->force_split = true;
      rc = dsh_sft_batch_upload(c, 1, frame);
      c->force_waves = 0;
      c->force_split = false;
      if (rc != DSH_OK && local_rc == DSH_OK) { local_rc = rc; local_err = c->err; }
    }
    if (rc == DSH_OK && !c->packed[0].h.split) {
      rc = DSH_ERR_ARG;
      if (local_rc == DSH_OK) { local_rc = rc; local_err = "the connected-mesh mode needs a band of at most 256 that is long enough to cut (two parts of four tile columns next to a separator of one bandwidth)"; }
    }
    SftSc init{};
    init.rank = rank0 + g;
    init.nranks = 2;
    init.send[0] = rc == DSH_OK ? 0.0 : 1.0;
    if (hipMemcpyAsync(c->d_sc, &init, sizeof(SftSc), hipMemcpyHostToDevice, c->stream) != hipSuccess) { err = "state upload failed"; return DSH_ERR_HIP; }
  }
  // rank-local failures are agreed on before the first phase (nobody is left inside a collective)
  {
    double* snd[2]; double* rcv[2];
    for (int g = 0; g < G; g++) { snd[g] = R[g].c->d_sc->send; rcv[g] = R[g].c->d_sc->recv; }
    int rc = cn_allreduce(R, comm, snd, rcv, 4, err);
    if (rc != DSH_OK) return rc;
    double bad = 0.0;
    dsh_ctx* c = R[0].c;
    if (hipMemcpyAsync(&bad, c->d_sc->recv, sizeof(bad), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { err = "read-back failed"; return DSH_ERR_HIP; }
    if (bad != 0.0) {
      if (local_rc != DSH_OK) { err = local_err; return local_rc; }
      err = "the other rank failed to set the problem up";
      return DSH_ERR_STATE;
    }
  }
  int rc;
  double* xs[2]; double* xr[2]; double* vx[2];
  int xl = 0, nx = 0;
  for (int g = 0; g < G; g++) {
    const SftDev& h = R[g].c->h_probs[0];
    xs[g] = h.part[rank0 + g].xchg; xr[g] = h.part[2].xchg; vx[g] = h.x;
    xl = h.sp_xl;
    nx = ((h.Dn + kNB - 1) / kNB) * kNB + 6;
  }
  for (int guard = 0; guard < DSH_MAX_ITERS + 1; guard++) {
    if ((rc = cn_phase(R, SFT_CN_LIN, err)) != DSH_OK) return rc;
    int again = 0, done = 0;
    do {
      if ((rc = cn_phase(R, SFT_CN_FAC, err)) != DSH_OK || (rc = cn_allreduce(R, comm, xs, xr, xl, err)) != DSH_OK) return rc;
      if ((rc = cn_phase(R, SFT_CN_SOL, err)) != DSH_OK || (rc = cn_allreduce(R, comm, vx, vx, nx, err)) != DSH_OK) return rc;
      if ((rc = cn_phase(R, SFT_CN_CTL, err)) != DSH_OK) return rc;
      int32_t flags[2];
      dsh_ctx* c = R[0].c;
      if (hipMemcpyAsync(flags, &c->d_sc->again, sizeof(flags), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { err = "flag read-back failed"; return DSH_ERR_HIP; }
      again = flags[0];
      done = flags[1];
    } while (again);
    if (done) break;
  }
  for (int g = 0; g < G; g++) {
    R[g].c->ran = true;
    rc = dsh_sft_batch_download(R[g].c, 1, &results[g]);
    if (rc != DSH_OK) { err = R[g].c->err; return rc; }
  }
  return DSH_OK;
}

}  // namespace

int dsh_sft_connected_solve(dsh_ctx* c, dsh_comm* cm, const dsh_sft_frame* frame, dsh_sft_result* result) {
  if (!c || !cm || !frame || !result || cm->ctx != c) return fail(c, DSH_ERR_ARG, "dsh_sft_connected_solve: bad argument");
  if (cm->nranks != 2) return fail(c, DSH_ERR_ARG, "dsh_sft_connected_solve: the cut has two parts: the communicator must have exactly two ranks");
  std::vector<ScRank> R{ScRank{c}};
  std::string err;
  const int rc = cn_solve(R, cm, cm->rank, frame, result, err);
  return rc == DSH_OK ? rc : fail(c, rc, "dsh_sft_connected_solve: " + err);
}

int dsh_sft_connected_solve_group(dsh_ctx* c0, dsh_ctx* c1, const dsh_sft_frame* frame, dsh_sft_result* results) {
  if (!c0 || !c1 || c0 == c1 || !frame || !results) return DSH_ERR_ARG;
  std::vector<ScRank> R{ScRank{c0}, ScRank{c1}};
  std::string err;
  const int rc = cn_solve(R, nullptr, 0, frame, results, err);
  return rc == DSH_OK ? rc : fail(c0, rc, "dsh_sft_connected_solve_group: " + err);
}

#ifdef DSH_LAB
// ---- lab entry points (include/defslam_hip_debug.h): libdefslam_hip_lab.so only ----------------------------------------
namespace {
struct EventPair {   // destroyed on every path
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t create() { hipError_t e = hipEventCreate(&e0); return e != hipSuccess ? e : hipEventCreate(&e1); }
  ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};
}  // namespace

int dsh_lab_set_option(dsh_ctx* c, const char* name, int value) {
  if (!c || !name) return DSH_ERR_ARG;
  const std::string k(name);
  if (k == "waves") { if (value != 0 && value != 4 && value != 8) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: waves is 0 (automatic), 4 or 8"); c->opt.waves = value; }
  else if (k == "dataflow") c->opt.dataflow = value != 0;
  else if (k == "wide_off") c->opt.wide_off = value != 0;
  else if (k == "rounds") c->opt.rounds = value != 0;
  else if (k == "streams") { if (value < 0 || value > dsh_ctx::kMaxSub) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: streams is 0 (automatic) or 1..4 sub-batches"); c->opt.streams = value; }
  else if (k == "split") { if (value < 0 || value > 2) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: split is 0 (off), 1 (wide bands only) or 2 (every band long enough)"); c->opt.split = value; }
  else if (k == "helpers_wbt") { if (value < 1 || value > 16) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: helpers_wbt is 1..16 (tiles of half-bandwidth from which parts get helper workgroups)"); c->opt.helpers_wbt = value; }
  else if (k == "owner_waves") { if (value != 8 && value != 16) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: owner_waves is 8 or 16 (wavefronts of a FACTOR workgroup with helpers)"); c->opt.owner_waves = value; }
  else if (k == "helpers") { if (value < -1 || value > 3) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: helpers is -1 (automatic) or 0..3 workgroups per part"); c->opt.helpers = value; }
  else if (k == "tail") { if (value < -1 || value > 8) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: tail is -1 (automatic), 0 (rounds to the end) or the number of problems per CU from which downwards the last problems go to the tail kernel"); c->opt.tail = value; }
  else if (k == "speculate") { if (value < 0 || value > SFT_SPEC_MAXK) return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: speculate is 0 (automatic) or 1..4 lanes"); c->opt.speculate = value; }
  else return fail(c, DSH_ERR_ARG, "dsh_lab_set_option: unknown option " + k);
  return DSH_OK;
}

int dsh_lab_sft_solver_info(dsh_ctx* c, int b, int32_t* out8) {
  if (!c || !out8 || b < 0 || b >= c->B) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_solver_info: bad argument");
  const SftDev& h = c->h_probs[b];
  out8[0] = h.split; out8[1] = h.sp_c0; out8[2] = h.sp_s; out8[3] = h.sp_n1p; out8[4] = h.sp_pad; out8[5] = c->spec_k; out8[6] = h.tile_mode; out8[7] = c->nw;
  return DSH_OK;
}

int dsh_lab_sft_run_timed(dsh_ctx* c, int launches, double* total_ms) {
  if (!c || launches <= 0 || !total_ms) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_run_timed: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_run_timed: host-only context, no GPU (there is no CPU fallback)");
  if (c->B <= 0) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_run_timed: nothing uploaded");
  (void)hipSetDevice(c->device);
  EventPair ev;
  HIPCHK(c, ev.create());
  HIPCHK(c, hipEventRecord(ev.e0, c->stream));
  for (int i = 0; i < launches; i++) { const int rc = run_once(c); if (rc != DSH_OK) return rc; }
  HIPCHK(c, hipEventRecord(ev.e1, c->stream));
  HIPCHK(c, hipEventSynchronize(ev.e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *total_ms = (double)ms;
  c->ran = true;
  return DSH_OK;
}

int dsh_lab_sft_assemble_timed(dsh_ctx* c, int launches, double* total_ms) {
  if (!c || launches <= 0 || !total_ms) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_assemble_timed: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_assemble_timed: host-only context, no GPU (there is no CPU fallback)");
  if (c->B <= 0 || !c->ran) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_assemble_timed: needs an uploaded batch that has run once");
  (void)hipSetDevice(c->device);
  EventPair ev;
  HIPCHK(c, ev.create());
  HIPCHK(c, hipEventRecord(ev.e0, c->stream));
  for (int i = 0; i < launches; i++) HIPCHK(c, sft_assembly_launch(c->d_probs, c->B, c->max_kd, c->jl_doubles, c->rounds_mode ? 8 : c->nw, c->stream));
  HIPCHK(c, hipEventRecord(ev.e1, c->stream));
  HIPCHK(c, hipEventSynchronize(ev.e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *total_ms = ms;
  c->ran = false;   // results of the last full run are gone (state reset, H reassembled at the initial state)
  return DSH_OK;
}

int dsh_lab_sft_wave_check(dsh_ctx* c, double rel, int launches, int only, double* x_ref, double* x_new, int32_t* ok2, double* ms2) {
  if (!c || launches <= 0 || !ms2) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_wave_check: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_wave_check: host-only context, no GPU (there is no CPU fallback)");
  if (c->B <= 0 || !c->ran) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_wave_check: needs an uploaded batch that has run once");
  for (int b = 0; b < c->B; b++)
    if (c->h_probs[b].tile_mode != 1) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_wave_check: register-window problems (half-bandwidth <= 128) only");
  (void)hipSetDevice(c->device);
  HIPCHK(c, sft_assembly_launch(c->d_probs, c->B, c->max_kd, c->jl_doubles, c->rounds_mode ? 8 : c->nw, c->stream));   // H of the initial state
  EventPair ev;
  HIPCHK(c, ev.create());
  for (int which = 0; which < 2; which++) {
    if (only == 2 - which) continue;   // only = 1: the four-wavefront solver alone, 2: the one-wavefront solver alone (lambda of the last reference run)
    HIPCHK(c, sft_wave_lab_launch(c->d_probs, c->B, which, rel, c->max_kd, c->jl_doubles, c->stream));   // (also the warm-up)
    HIPCHK(c, hipEventRecord(ev.e0, c->stream));
    for (int i = 0; i < launches; i++) HIPCHK(c, sft_wave_lab_launch(c->d_probs, c->B, which, rel, c->max_kd, c->jl_doubles, c->stream));
    HIPCHK(c, hipEventRecord(ev.e1, c->stream));
    HIPCHK(c, hipEventSynchronize(ev.e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    ms2[which] = (double)ms / launches;
    double* dst = which == 0 ? x_ref : x_new;
    size_t off = 0;
    for (int b = 0; b < c->B; b++) {
      const SftDev& h = c->h_probs[b];
      const size_t Dnp = (size_t)((h.Dn + kNB - 1) / kNB) * kNB;
      if (dst) HIPCHK(c, hipMemcpy(dst + off, h.x, 8 * (Dnp + 6), hipMemcpyDeviceToHost));
      off += Dnp + 6;
      double flag = 0.0;
      if (ok2) { HIPCHK(c, hipMemcpy(&flag, h.dbg + 2, 8, hipMemcpyDeviceToHost)); ok2[2 * b + which] = (int32_t)flag; }
    }
  }
  return DSH_OK;   // (H stays assembled at the initial state: the check can be repeated; the results of the last full run are stale)
}

int dsh_lab_sft_rounds_timed(dsh_ctx* c, double* ms4, int32_t* rounds) {
  if (!c || !ms4) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_rounds_timed: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_rounds_timed: host-only context, no GPU (there is no CPU fallback)");
  if (c->B <= 0 || !c->rounds_mode) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_rounds_timed: needs an uploaded batch that runs as rounds of phase kernels");
  (void)hipSetDevice(c->device);
  std::vector<hipEvent_t> ev;
  c->phase_events = &ev;
  c->phase_ids.clear();
  const int rc = run_rounds(c);
  c->phase_events = nullptr;
  (void)hipStreamSynchronize(c->stream);
  // launches by phase: ms[0] INIT, [1] LIN, [2] FACTOR, [3] TRIAL, [4] the tail kernel; [5] = factorisations done by the FACTOR launches, [6] = linearisations done by the LIN launches
  for (int i = 0; i < 7; i++) ms4[i] = 0.0;
  { int32_t cnt[16] = {0}; if (hipMemcpy(cnt, c->d_counters, sizeof(cnt), hipMemcpyDeviceToHost) == hipSuccess) { ms4[5] = (double)cnt[8]; ms4[6] = (double)cnt[9]; } }
  int n_trial = 0;
  for (size_t i = 0; i + 1 < ev.size(); i += 2) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
    const int ph = c->phase_ids[i / 2];
    ms4[ph == SFTB_PH_INIT ? 0 : ph == SFTB_PH_LIN ? 1 : ph == SFTB_PH_FACTOR ? 2 : ph == SFTB_PH_TRIAL ? 3 : 4] += ms;
    n_trial += ph == SFTB_PH_TRIAL;
  }
  if (rounds) *rounds = n_trial;
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  if (rc == DSH_OK) c->ran = true;
  return rc;
}

int dsh_lab_sft_dump(dsh_ctx* c, int b, int what, int64_t n, double* out) {
  if (!c || !out || b < 0 || b >= c->B || n <= 0) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_dump: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_dump: host-only context");
  const SftDev& h = c->h_probs[b];
  const double* src = what == 0 ? h.Lb : what == 1 ? h.Linv : what == 2 ? h.Lbord : what == 3 ? h.Hc : what == 4 ? h.Hbord : what == 5 ? h.x : what == 6 ? h.Hcorner :
                      (what == 8 || what == 9) ? (const double*)h.part[what - 8].sync : h.dbg;   // 8, 9: the helper statistics of part 0 / 1 (int32 words)
  if (!src) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_dump: the problem has no such buffer");
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, src, 8 * (size_t)n, hipMemcpyDeviceToHost));
  return DSH_OK;
}

int dsh_lab_sft_phase_ms(dsh_ctx* c, int b, double* out8) {
  if (!c || !out8 || b < 0 || b >= c->B) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_phase_ms: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_phase_ms: host-only context");
  if (!c->ran) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_phase_ms: no run");
#ifndef SFT_PHASE_TIMERS
  return fail(c, DSH_ERR_STATE, "dsh_lab_sft_phase_ms: this build has no phase timers (make lab EXTRA=-DSFT_PHASE_TIMERS)");
#else
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out8, c->h_probs[b].dbg, 8 * sizeof(double), hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; i++) out8[i] *= 1e-5;  // 100 MHz ticks -> ms
  {   // sections of the assembly (shader-clock cycles of wave 0), printed for tuning runs
    double as[6];
    HIPCHK(c, hipMemcpy(as, c->h_probs[b].dbg + 32, sizeof(as), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "[lab] assembly sections of problem %d, kcycles of wave 0: corner %.0f, diagonal gather %.0f, butterfly+finish %.0f, off-diagonal %.0f, round overhead %.0f; rounds %.0f\n",
                 b, as[0] * 1e-3, as[1] * 1e-3, as[2] * 1e-3, as[3] * 1e-3, as[4] * 1e-3, as[5]);
  }
  return DSH_OK;
#endif
}

int dsh_lab_sft_step_trace(dsh_ctx* c, int b, double* out64) {
  if (!c || !out64 || b < 0 || b >= c->B) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_step_trace: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_step_trace: host-only context");
  if (!c->ran) return fail(c, DSH_ERR_STATE, "dsh_lab_sft_step_trace: no run");
#ifndef SFT_STEP_TRACE
  return fail(c, DSH_ERR_STATE, "dsh_lab_sft_step_trace: this build has no step trace (make lab EXTRA=-DSFT_STEP_TRACE)");
#else
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out64, c->h_probs[b].dbg + 16, 64 * sizeof(double), hipMemcpyDeviceToHost));
  return DSH_OK;
#endif
}

int dsh_lab_sft_system(dsh_ctx* c, int b, int32_t D, double* H, double* bvec, double* chi2) {
  if (!c || b < 0 || b >= c->B) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_system: bad argument");
  if (c->host_only) return fail(c, DSH_ERR_NO_DEVICE, "dsh_lab_sft_system: host-only context");
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));   // an asynchronous run may still be using the problem table
  SftDev h = c->h_probs[b];
  if (D != 6 + h.Dn) return fail(c, DSH_ERR_ARG, "dsh_lab_sft_system: D mismatch");
  // flip the mode of this one problem, run it alone, restore
  const int32_t mode_saved = h.mode, split_saved = h.split;
  h.mode = 1;
  h.split = 0;   // the one-workgroup kernel assembles into the undivided band matrix
  HIPCHK(c, hipMemcpy(c->d_probs + b, &h, sizeof(SftDev), hipMemcpyHostToDevice));
  { LDS_LOCK(); HIPCHK(c, sft_lm_launch(c->d_probs + b, 1, c->max_kd, c->jl_doubles, c->nw, LDS_MARKS(c).lm, c->stream)); }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->ran = false;   // the state of problem b was reset: a download would not return the results of the last run
  h.mode = mode_saved;
  h.split = split_saved;
  HIPCHK(c, hipMemcpy(c->d_probs + b, &h, sizeof(SftDev), hipMemcpyHostToDevice));
  const size_t Dnp = (size_t)((h.Dn + kNB - 1) / kNB) * kNB;
  const size_t band_elems = h.tile_mode ? (Dnp / kTS) * (size_t)h.tpr * kTS * kTS : Dnp * (size_t)h.ldh;
  std::vector<double> Hb(h.tile_mode == 1 ? 0 : band_elems), Hbord(SFT_BORDER * Dnp), Hc(56);
  auto hidx = [&](int r, int cc) -> size_t {
    const size_t tb = ((size_t)(r >> 4) * h.tpr + ((r >> 4) - (cc >> 4))) * (kTS * kTS);
    if (h.tile_mode == 1) return tb + ((((r & 15) & 3) << 4) + (cc & 15)) * 4 + ((r & 15) >> 2);
    if (h.tile_mode == 2) return tb + ((((cc & 15) & 3) << 4) + (r & 15)) * 4 + ((cc & 15) >> 2);   // wide mode keeps the tiles transposed
    return (size_t)r * h.ldh + (cc - r + h.kd);
  };
  // tile mode 1: H is kept as compact 3x3 blocks; it is read here the way the factorisation reads it, through the gather lists
  const dsh::SftGraph& g = *c->packed[b].g;
  std::vector<double> Hcomp;
  if (h.tile_mode == 1) {
    Hcomp.resize(g.hc_elems());
    HIPCHK(c, hipMemcpy(Hcomp.data(), h.Hc, 8 * Hcomp.size(), hipMemcpyDeviceToHost));
  } else {
    HIPCHK(c, hipMemcpy(Hb.data(), h.Hb, 8 * Hb.size(), hipMemcpyDeviceToHost));
  }
  auto hval = [&](int r, int cc) -> double {
    if (h.tile_mode != 1) return Hb[hidx(r, cc)];
    const int I = r >> 4, d = I - (cc >> 4);
    if (d > kBT) return 0.0;
    const int lane = (((r & 15) & 3) << 4) + (cc & 15), q = (r & 15) >> 2;
    return Hcomp[g.hgather[(((size_t)I * (kBT + 1) + d) * 64 + lane) * 4 + q] / 8];
  };
  HIPCHK(c, hipMemcpy(Hbord.data(), h.Hbord, 8 * Hbord.size(), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(Hc.data(), h.Hcorner, 8 * 49, hipMemcpyDeviceToHost));
  if (chi2) HIPCHK(c, hipMemcpy(chi2, h.dbg, 8, hipMemcpyDeviceToHost));
  // reference index order: camera 0..5, node a -> 6+3a
  if (H) {
    std::fill(H, H + (size_t)D * D, 0.0);
    for (int r = 0; r < h.Dn; r++)
      for (int k = 0; k <= h.kd; k++) {
        const int cidx = r - h.kd + k;
        if (cidx < 0) continue;
        const double v = hval(r, cidx);
        H[(size_t)(6 + r) + (size_t)(6 + cidx) * D] = v;
        H[(size_t)(6 + cidx) + (size_t)(6 + r) * D] = v;
      }
    for (int k = 0; k < 6; k++)
      for (int r = 0; r < h.Dn; r++) {
        const double v = Hbord[(size_t)k * Dnp + r];
        H[(size_t)k + (size_t)(6 + r) * D] = v;
        H[(size_t)(6 + r) + (size_t)k * D] = v;
      }
    for (int r = 0; r < 6; r++)
      for (int k = 0; k <= r; k++) {
        H[(size_t)r + (size_t)k * D] = Hc[r * 7 + k];
        H[(size_t)k + (size_t)r * D] = Hc[r * 7 + k];
      }
  }
  if (bvec) {
    for (int r = 0; r < 6; r++) bvec[r] = Hc[42 + r];
    for (int r = 0; r < h.Dn; r++) bvec[6 + r] = Hbord[(size_t)6 * Dnp + r];
  }
  return DSH_OK;
}
#endif  // DSH_LAB

}  // extern "C"
