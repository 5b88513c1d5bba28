// Mapping side of the host integration shim: the reference's warp database and normal estimator on top of
// the C ABI of libdefslam_hip.so.
//
//   SchwarpDatabaseHIP<...> : WarpDatabase     drop-in for defSLAM::SchwarpDatabase (Modules/Mapping/SchwarpDatabase.h/.cc), the
//        plugin DefLocalMapping installs with `warpDB_ = new SchwarpDatabase(reg_)` (DefLocalMapping.cc:65): add(KeyFrame*) finds
//        the anchor keyframes, initialises the warp, searches more matches through it and fits the Schwarzian warp -- every
//        numeric step on the GPU (dsh_warp_initialize, dsh_schwarp_eval, dsh_search_by_schwarp, dsh_schwarp_fit), the object
//        bookkeeping (map point observations, the DiffProp database) exactly where the reference does it.
//   ObtainK1K2HIP(ctx, warpDB)                 drop-in for NormalEstimator(warpDB).ObtainK1K2() (NormalEstimator.cc:38-229).
//   enable_device_records(capacity) + ObtainK1K2DeviceHIP(ctx, warpDB): the same two steps with the DiffProp records resident in HBM
//        (dsh_diffdb): the fit appends them on the device, the normal solve groups them there -- mapPointsDB_ stays empty, only key
//        points go up and normals come down.
//
// Templates over the reference's own classes (members cited at each use); the repository's CI instantiates them with the stand-ins
// of integration/standin_mapping_types.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <set>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../include/defslam_hip.h"

namespace defslam_hip {

// Base = defSLAM::WarpDatabase (WarpDatabase.h:39-72: virtual add / erase / clear, mapPointsDB_, newInformation_)
template <class Base, class KeyFrameT, class DefKeyFrameT, class MapPointT, class DiffPropT>
class SchwarpDatabaseHIP : public Base {
 public:
  typedef std::vector<std::pair<size_t, size_t>> Matches;
  SchwarpDatabaseHIP(dsh_ctx* ctx, double lambda) : ctx_(ctx), lambda_(lambda) {}

  // SchwarpDatabase::add (SchwarpDatabase.cc:50-128)
  void add(KeyFrameT* mpCurrentKeyFrame) override {
    if (mpkeyframes.count(mpCurrentKeyFrame)) return;                    // CHECK(count == 0) in the reference
    if (mpkeyframes.empty()) { mpkeyframes.insert(mpCurrentKeyFrame); return; }
    const std::vector<MapPointT*> vpMapPointMatches = mpCurrentKeyFrame->GetMapPointMatches();
    std::unordered_map<KeyFrameT*, int> countKFMatches;                  // reference keyframes of the matched map points
    std::vector<KeyFrameT*> order;                                       // (first-seen order: deterministic, unlike the hash map's)
    for (MapPointT* mapPoint : vpMapPointMatches) {
      if (!mapPoint || mapPoint->isBad()) continue;
      KeyFrameT* refkf = mapPoint->GetReferenceKeyFrame();
      if (!countKFMatches.count(refkf)) { countKFMatches[refkf] = 0; order.push_back(refkf); }
      countKFMatches[refkf]++;
    }
    for (KeyFrameT* refkf : order) {
      Matches vMatchedIndices;
      for (MapPointT* mapPoint : vpMapPointMatches) {
        if (!mapPoint || mapPoint->isBad()) continue;
        if (mapPoint->IsInKeyFrame(mpCurrentKeyFrame) && mapPoint->IsInKeyFrame(refkf))
          vMatchedIndices.push_back({(size_t)mapPoint->GetIndexInKeyFrame(refkf), (size_t)mapPoint->GetIndexInKeyFrame(mpCurrentKeyFrame)});
      }
      if (vMatchedIndices.size() < 20) continue;
      DefKeyFrameT* KF = static_cast<DefKeyFrameT*>(refkf);
      std::vector<double> x(2 * (size_t)KF->NCu * KF->NCv, 0.0);
      findbyWarp(refkf, mpCurrentKeyFrame, vMatchedIndices, x, lambda_);
      if (vMatchedIndices.size() < 20) continue;
      calculateSchwarps(refkf, mpCurrentKeyFrame, vMatchedIndices, x, lambda_);
      static_cast<DefKeyFrameT*>(mpCurrentKeyFrame)->KeyframesRelated++;
    }
    mpkeyframes.insert(mpCurrentKeyFrame);
  }
  void erase(KeyFrameT* kf) override { mpkeyframes.erase(kf); }
  void clear() override {
    mpkeyframes.clear();
    if (devdb_) { dsh_diffdb_clear(devdb_); point_id_.clear(); points_.clear(); anchor_kf_.clear(); tag_kf_.clear(); }
  }
  int last_status() const { return status_; }
  ~SchwarpDatabaseHIP() { if (devdb_) dsh_diffdb_destroy(devdb_); }

  // Device-resident records (include/defslam_hip.h: dsh_diffdb): from now on calculateSchwarps leaves the DiffProp records in HBM.
  bool enable_device_records(int64_t capacity_records) { return devdb_ || dsh_diffdb_create(ctx_, capacity_records, &devdb_) == DSH_OK; }
  dsh_diffdb* device_records() const { return devdb_; }
  const std::vector<MapPointT*>& device_points() const { return points_; }       // map point of a point id
  const std::vector<KeyFrameT*>& device_tags() const { return tag_kf_; }         // second keyframe of a record tag
  int32_t device_point_id(MapPointT* mp) const { auto it = point_id_.find(mp); return it == point_id_.end() ? -1 : it->second; }
  // the keyframe the point's records are anchored in: its reference keyframe at the time its id was created (every record of the id is stored
  // under `GetReferenceKeyFrame() == KFi`, so a point that is re-anchored later simply stops receiving records under the old anchor)
  KeyFrameT* device_point_anchor(int32_t id) const { return anchor_kf_[id]; }

 protected:
  static dsh_bbs bbs_of(DefKeyFrameT* KF, int valdim) { return dsh_bbs{KF->umin, KF->umax, KF->NCu, KF->vmin, KF->vmax, KF->NCv, valdim}; }

  // matched key points as the flat arrays of the ABI: normalised key points of both keyframes, sqrt(invSigma2[octave]) of keyframe 1
  static void gather(DefKeyFrameT* KF, DefKeyFrameT* KF2, const Matches& m, std::vector<float>& k1, std::vector<float>& k2, std::vector<float>& isg) {
    k1.clear(); k2.clear(); isg.clear();
    for (const auto& pr : m) {
      k1.push_back(KF->mpKeypointNorm[pr.first].pt.x); k1.push_back(KF->mpKeypointNorm[pr.first].pt.y);
      k2.push_back(KF2->mpKeypointNorm[pr.second].pt.x); k2.push_back(KF2->mpKeypointNorm[pr.second].pt.y);
      isg.push_back(std::sqrt(KF->mvInvLevelSigma2[KF->mvKeysUn[pr.first].octave]));
    }
  }

  // DefORBmatcher::findbyWarp = CalculateInitialSchwarp + searchBySchwarp + registration of the new matches (DefORBmatcher.cc:47-71)
  void findbyWarp(KeyFrameT* Kf1, KeyFrameT* Kf2, Matches& vMatchedIndices, std::vector<double>& x, double lambda) {
    DefKeyFrameT* KF = static_cast<DefKeyFrameT*>(Kf1);
    DefKeyFrameT* KF2 = static_cast<DefKeyFrameT*>(Kf2);
    const dsh_bbs bbs = bbs_of(KF, KF->valdim);
    // ---- CalculateInitialSchwarp (DefORBmatcher.cc:111-187): Warp::initialize, NaN control points -> 0, then the matches whose
    // squared (loss-corrected, see below) reprojection residual of the Warp cost function with (fx, fy) exceeds 20 are removed
    std::vector<float> k1, k2, isg;
    gather(KF, KF2, vMatchedIndices, k1, k2, isg);
    const int P = (int)vMatchedIndices.size();
    int32_t ok = 0;
    status_ = dsh_warp_initialize(ctx_, &bbs, P, k1.data(), k2.data(), lambda, x.data(), &ok);
    if (status_ != DSH_OK) return;
    // (the reference clears NaN control points among the first 2 NCu NCu entries only: its loops run to
    // _NumberOfControlPointsU * _NumberOfControlPointsU * 2, DefORBmatcher.cc:139-152)
    for (size_t i = 0; i < x.size() && i < 2 * (size_t)KF->NCu * KF->NCu; i++)
      if (std::isnan(x[i])) x[i] = 0.0;
    (void)ok;
    const int Nc = bbs.nptsu * bbs.nptsv;
    std::vector<double> res(2 * (size_t)P + 4 * (size_t)Nc);
    status_ = dsh_schwarp_eval(ctx_, &bbs, P, k1.data(), k2.data(), isg.data(), (double)KF->fx, (double)KF->fy, 0.0, x.data(), res.data(), nullptr);
    if (status_ != DSH_OK) return;
    // The reference reads the residuals from ceres::Problem::Evaluate with default EvaluateOptions (DefORBmatcher.cc:155-166):
    // apply_loss_function is true, so the ONE residual block of 2P entries comes back loss-corrected by HuberLoss(5.77) -- Ceres'
    // Corrector with rho'' <= 0 scales the block by sqrt(rho'(s)), s = |r|^2: 1 for s <= 5.77^2, sqrt(5.77 / |r|) beyond.
    {
      double s = 0.0;
      for (int i = 0; i < 2 * P; i++) s += res[i] * res[i];
      if (s > 5.77 * 5.77) {
        const double sc = std::sqrt(5.77 / std::sqrt(s));
        for (int i = 0; i < 2 * P; i++) res[i] *= sc;
      }
    }
    // Warp::Evaluate lays the residuals out as [x_0 .. x_{P-1}, y_0 .. y_{P-1}] (Schwarp.cc:277-282) and the reference tests
    // residuals[2 i]^2 + residuals[2 i + 1]^2 for match i (DefORBmatcher.cc:167-175): entries 2i and 2i+1 of that array, i.e. two
    // neighbouring x- (or y-) residuals, not the two components of match i.  Which matches go is observable behaviour: reproduced.
    Matches kept;
    for (int i = 0; i < P; i++) {
      const double error = res[2 * i] * res[2 * i] + res[2 * i + 1] * res[2 * i + 1];
      if (error > 20) KF2->EraseMapPointMatch(vMatchedIndices[i].second);
      else kept.push_back(vMatchedIndices[i]);
    }
    vMatchedIndices.swap(kept);
    // ---- searchBySchwarp (DefORBmatcher.cc:189-294): candidates = map points of keyframe 1 that are not yet in keyframe 2
    std::vector<int> listMapPoints;
    std::vector<float> q1;
    std::vector<uint8_t> d1;
    for (size_t i = 0; i < KF->mpKeypointNorm.size(); i++) {
      MapPointT* pMP = KF->GetMapPoint(i);
      if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(KF2)) continue;
      listMapPoints.push_back((int)i);
      q1.push_back(KF->mpKeypointNorm[i].pt.x); q1.push_back(KF->mpKeypointNorm[i].pt.y);
      const uint8_t* d = KF->descriptor(i);                                // mDescriptors.row(i): 32 bytes
      d1.insert(d1.end(), d, d + 32);
    }
    Matches vMatchedIndices2;
    if (!listMapPoints.empty()) {
      const int N2 = KF2->N;
      std::vector<float> kp2(2 * (size_t)N2);
      std::vector<uint8_t> has(N2), d2(32 * (size_t)N2);
      for (int j = 0; j < N2; j++) {
        kp2[2 * j] = KF2->mvKeysUn[j].pt.x; kp2[2 * j + 1] = KF2->mvKeysUn[j].pt.y;
        has[j] = KF2->GetMapPoint(j) != nullptr;
        std::copy(KF2->descriptor(j), KF2->descriptor(j) + 32, d2.begin() + 32 * (size_t)j);
      }
      const float cam[4] = {KF2->fx, KF2->fy, KF2->cx, KF2->cy}, bnd[4] = {KF2->mnMinX, KF2->mnMaxX, KF2->mnMinY, KF2->mnMaxY};
      std::vector<int32_t> m(listMapPoints.size());
      int32_t nm = 0;
      status_ = dsh_search_by_schwarp(ctx_, &bbs, x.data(), (int)listMapPoints.size(), q1.data(), d1.data(), cam, bnd, KF2->gridCols(), KF2->gridRows(), N2,
                                      kp2.data(), d2.data(), has.data(), /*radius th=*/2.f, /*TH_LOW=*/50, m.data(), &nm);
      if (status_ != DSH_OK) return;
      for (size_t i = 0; i < m.size(); i++)
        if (m[i] >= 0) vMatchedIndices2.push_back({(size_t)listMapPoints[i], (size_t)m[i]});
    }
    for (const auto& pr : vMatchedIndices2) {                              // DefORBmatcher.cc:58-66
      MapPointT* pMP = Kf1->GetMapPoint(pr.first);
      if (pMP) { pMP->AddObservation(Kf2, pr.second); Kf2->addMapPoint(pMP, pr.second); }
    }
    vMatchedIndices.insert(vMatchedIndices.end(), vMatchedIndices2.begin(), vMatchedIndices2.end());
  }

  // SchwarpDatabase::calculateSchwarps (SchwarpDatabase.cc:145-349)
  void calculateSchwarps(KeyFrameT* KFi, KeyFrameT* KF2i, Matches& vMatchedIndices, std::vector<double>& x, double lambda) {
    DefKeyFrameT* KF = static_cast<DefKeyFrameT*>(KFi);
    DefKeyFrameT* KF2 = static_cast<DefKeyFrameT*>(KF2i);
    const dsh_bbs bbs = bbs_of(KF, KF->valdim);
    std::vector<float> k1, k2, isg;
    gather(KF, KF2, vMatchedIndices, k1, k2, isg);
    const int P = (int)vMatchedIndices.size();
    std::vector<dsh_diffprop> dp(P);
    std::vector<uint8_t> drop(P);
    int32_t info[2];
    double costs[2];
    // the reference hands (fy, fx) to Warp's (fx, fy) slots (:199-201); 3 LM iterations (:213)
    if (devdb_) {
      // which matches the reference would store (:268-297, the drop test aside -- the library applies it): both map points alive and the
      // point anchored in the estimated keyframe; their records go into the device database under the point's id
      std::vector<int32_t> pid(P, -1), i2(P);
      for (int ikp = 0; ikp < P; ikp++) {
        i2[ikp] = (int32_t)vMatchedIndices[ikp].second;
        MapPointT* mapPoint = KF->GetMapPoint(vMatchedIndices[ikp].first);
        MapPointT* mapPoint2 = KF2->GetMapPoint(vMatchedIndices[ikp].second);
        if (!mapPoint || !mapPoint2 || mapPoint->isBad() || mapPoint2->isBad() || mapPoint->GetReferenceKeyFrame() != KFi) continue;
        auto it = point_id_.find(mapPoint);
        if (it == point_id_.end()) { it = point_id_.emplace(mapPoint, (int32_t)points_.size()).first; points_.push_back(mapPoint); anchor_kf_.push_back(KFi); }
        else if (anchor_kf_[it->second] != KFi) continue;   // re-anchored since its first record: its old records belong to another keyframe
        pid[ikp] = it->second;
      }
      dsh_schwarp_problem q{};
      q.bbs = bbs; q.P = P; q.kp1 = k1.data(); q.kp2 = k2.data(); q.invsig = isg.data(); q.fx_slot = (double)KF->fy; q.fy_slot = (double)KF->fx;
      q.lambda = lambda; q.fx = KF->fx; q.fy = KF->fy; q.max_iters = 3; q.x = x.data(); q.diff = nullptr; q.drop = drop.data();
      const dsh_schwarp_store st{pid.data(), i2.data(), (int32_t)tag_kf_.size()};
      tag_kf_.push_back(KF2i);
      status_ = dsh_schwarp_fit_batch_store(ctx_, 1, &q, &st, devdb_);
      if (status_ != DSH_OK) return;
      for (int ikp = 0; ikp < P; ikp++) {
        MapPointT* mapPoint = KF->GetMapPoint(vMatchedIndices[ikp].first);
        MapPointT* mapPoint2 = KF2->GetMapPoint(vMatchedIndices[ikp].second);
        if (!mapPoint || !mapPoint2 || mapPoint->isBad() || mapPoint2->isBad()) continue;
        if (drop[ikp]) { mapPoint2->EraseObservation(KF2); KF2->EraseMapPointMatch(vMatchedIndices[ikp].second); continue; }
        if (pid[ikp] >= 0) this->newInformation_[mapPoint] = true;
      }
      return;
    }
    status_ = dsh_schwarp_fit(ctx_, &bbs, P, k1.data(), k2.data(), isg.data(), (double)KF->fy, (double)KF->fx, lambda, KF->fx, KF->fy, 3, x.data(), dp.data(),
                              drop.data(), info, costs);
    if (status_ != DSH_OK) return;
    for (int ikp = 0; ikp < P; ikp++) {                                    // :268-345
      const size_t idx1 = vMatchedIndices[ikp].first, idx2 = vMatchedIndices[ikp].second;
      MapPointT* mapPoint = KF->GetMapPoint(idx1);
      MapPointT* mapPoint2 = KF2->GetMapPoint(idx2);
      if (!mapPoint || !mapPoint2) continue;
      if (mapPoint->isBad() || mapPoint2->isBad()) continue;
      if (drop[ikp]) {                                                     // reprojection error of the fitted warp > 10 px (:283-293)
        mapPoint2->EraseObservation(KF2);
        KF2->EraseMapPointMatch(idx2);
        continue;
      }
      if (mapPoint->GetReferenceKeyFrame() != KFi) continue;               // only points anchored in the estimated keyframe are saved
      this->mapPointsDB_[mapPoint].push_back(std::shared_ptr<DiffPropT>(new DiffPropT()));
      std::shared_ptr<DiffPropT> d = this->mapPointsDB_[mapPoint].back();
      d->KFToKF = std::pair<KeyFrameT*, KeyFrameT*>(KFi, KF2i);
      d->idx1 = idx1; d->idx2 = idx2;
      const dsh_diffprop& q = dp[ikp];
      d->I1u = q.I1u; d->I1v = q.I1v; d->I2u = q.I2u; d->I2v = q.I2v;
      d->J12a = q.J12a; d->J12b = q.J12b; d->J12c = q.J12c; d->J12d = q.J12d;
      d->J21a = q.J21a; d->J21b = q.J21b; d->J21c = q.J21c; d->J21d = q.J21d;
      d->H12uux = q.H12uux; d->H12uuy = q.H12uuy; d->H12uvx = q.H12uvx; d->H12uvy = q.H12uvy; d->H12vvx = q.H12vvx; d->H12vvy = q.H12vvy;
      this->newInformation_[mapPoint] = true;
    }
  }

  dsh_ctx* ctx_;
  double lambda_;
  std::set<KeyFrameT*> mpkeyframes;
  int status_ = DSH_OK;
  dsh_diffdb* devdb_ = nullptr;
  std::unordered_map<MapPointT*, int32_t> point_id_;
  std::vector<MapPointT*> points_;
  std::vector<KeyFrameT*> anchor_kf_;   // per point id
  std::vector<KeyFrameT*> tag_kf_;
};

// NormalEstimator::ObtainK1K2 (NormalEstimator.cc:38-229) over the warp database: the points with new information, their DiffProp
// records, start values from the surfaces; results written where the reference writes them.  Returns the number of points solved.
template <class WarpDBT, class KeyFrameT, class DefKeyFrameT, class MapPointT>
int ObtainK1K2HIP(dsh_ctx* ctx, WarpDBT* warpDB) {
  auto& toProcess = warpDB->getToProccess();                               // WarpDatabase.h:65
  auto& diffDB = warpDB->getDiffDatabase();                                // WarpDatabase.h:61
  std::vector<int32_t> rec_ptr{0};
  std::vector<dsh_diffprop> recs;
  std::vector<uint8_t> is_ref, has_fn, has_x0;
  std::vector<float> first_n, x0, ref_uv;
  std::vector<MapPointT*> pts;
  std::vector<typename std::remove_reference<decltype(*diffDB.begin()->second.begin()->get())>::type*> flat;
  for (auto& pr : toProcess) {                                             // NormalEstimator.cc:50-64
    if (!pr.second) continue;
    pr.second = false;
    MapPointT* mp = pr.first;
    if (!mp || mp->isBad()) continue;
    auto& v = diffDB[mp];
    if (v.empty()) continue;
    KeyFrameT* refKF = mp->GetReferenceKeyFrame();
    for (auto& d : v) {
      recs.push_back(dsh_diffprop{d->I1u, d->I1v, d->I2u, d->I2v, d->J12a, d->J12b, d->J12c, d->J12d, d->J21a, d->J21b, d->J21c, d->J21d,
                                  d->H12uux, d->H12uuy, d->H12uvx, d->H12uvy, d->H12vvx, d->H12vvy});
      flat.push_back(d.get());
      const bool ref = refKF == d->KFToKF.first;
      is_ref.push_back(ref);
      float Ni[3] = {0, 0, 0};
      const bool h = !ref && static_cast<DefKeyFrameT*>(d->KFToKF.first)->surface->getNormalSurfacePoint(d->idx1, Ni);
      has_fn.push_back(h);
      first_n.push_back(h ? Ni[0] : 0.f); first_n.push_back(h ? Ni[1] : 0.f);
    }
    rec_ptr.push_back((int32_t)recs.size());
    pts.push_back(mp);
    const size_t idx = mp->GetIndexInKeyFrame(refKF);
    float Ni[3] = {0, 0, 0};
    const bool h = static_cast<DefKeyFrameT*>(refKF)->surface->getNormalSurfacePoint(idx, Ni);   // :125-135
    has_x0.push_back(h);
    x0.push_back(h ? Ni[0] : 0.f); x0.push_back(h ? Ni[1] : 0.f);
    const auto& kpn = static_cast<DefKeyFrameT*>(refKF)->mpKeypointNorm[idx].pt;
    ref_uv.push_back(kpn.x); ref_uv.push_back(kpn.y);
  }
  const int P = (int)pts.size();
  if (P == 0) return 0;
  std::vector<double> k(2 * (size_t)P), cov(4 * (size_t)P);
  std::vector<int32_t> st(P);
  std::vector<float> nref(3 * (size_t)P), nrec(3 * recs.size());
  std::vector<uint8_t> wr(recs.size());
  if (dsh_normals_estimate(ctx, P, rec_ptr.data(), recs.data(), is_ref.data(), first_n.data(), has_fn.data(), x0.data(), has_x0.data(), ref_uv.data(), k.data(),
                           cov.data(), st.data(), nref.data(), nrec.data(), wr.data(), nullptr) != DSH_OK)
    return -1;
  int solved = 0;
  for (int p = 0; p < P; p++) {
    MapPointT* mp = pts[p];
    if (st[p] == 0) {                                                      // solved and covariance available
      solved++;
      KeyFrameT* refKF = mp->GetReferenceKeyFrame();
      for (int k = 0; k < 4; k++) mp->covNorm[k] = cov[4 * (size_t)p + k];   // NormalEstimator.cc:153-159
      static_cast<DefKeyFrameT*>(refKF)->surface->setNormalSurfacePoint(mp->GetIndexInKeyFrame(refKF), &nref[3 * (size_t)p]);   // :160-170
    }
    // propagation to the second keyframe of every record (:173-224); the library reports which records the reference writes
    // (none after a covariance failure: the reference `continue`s before this loop)
    for (int r = rec_ptr[p]; r < rec_ptr[p + 1]; r++)
      if (wr[r]) static_cast<DefKeyFrameT*>(flat[r]->KFToKF.second)->surface->setNormalSurfacePoint(flat[r]->idx2, &nrec[3 * (size_t)r]);
  }
  return solved;
}

// The same over the device-resident records (SchwarpDatabaseHIP::enable_device_records): key points in, normals out.  Every stored record
// is anchored in the keyframe that was its point's reference keyframe when the point got its id (the fit stores no other), so there are no
// first-keyframe normals to look up.  A point whose reference keyframe changed since (MapPoint::EraseObservation can reassign mpRefKF) is
// SKIPPED: the reference would treat its records as non-reference ones (NormalEstimator.cc:80) and that needs the host route's first-keyframe
// normals; *skipped_reanchored counts them (they stay unsolved, their flag is cleared like the reference clears it).
template <class WarpDBT, class KeyFrameT, class DefKeyFrameT, class MapPointT>
int ObtainK1K2DeviceHIP(dsh_ctx* ctx, WarpDBT* warpDB, int* skipped_reanchored = nullptr) {
  if (skipped_reanchored) *skipped_reanchored = 0;
  auto& toProcess = warpDB->getToProccess();
  std::vector<int32_t> ids;
  std::vector<uint8_t> has_x0;
  std::vector<float> x0, ref_uv;
  std::vector<MapPointT*> pts;
  for (auto& pr : toProcess) {
    if (!pr.second) continue;
    pr.second = false;
    MapPointT* mp = pr.first;
    if (!mp || mp->isBad()) continue;
    const int32_t id = warpDB->device_point_id(mp);
    if (id < 0) continue;
    KeyFrameT* refKF = mp->GetReferenceKeyFrame();
    if (refKF != warpDB->device_point_anchor(id)) { if (skipped_reanchored) ++*skipped_reanchored; continue; }
    const size_t idx = mp->GetIndexInKeyFrame(refKF);
    float Ni[3] = {0, 0, 0};
    const bool h = static_cast<DefKeyFrameT*>(refKF)->surface->getNormalSurfacePoint(idx, Ni);
    ids.push_back(id); pts.push_back(mp);
    has_x0.push_back(h);
    x0.push_back(h ? Ni[0] : 0.f); x0.push_back(h ? Ni[1] : 0.f);
    const auto& kpn = static_cast<DefKeyFrameT*>(refKF)->mpKeypointNorm[idx].pt;
    ref_uv.push_back(kpn.x); ref_uv.push_back(kpn.y);
  }
  const int P = (int)pts.size();
  if (P == 0) return 0;
  const int32_t max_rec = (int32_t)dsh_diffdb_count(warpDB->device_records());
  std::vector<double> k(2 * (size_t)P), cov(4 * (size_t)P);
  std::vector<int32_t> st(P), rp(max_rec), rt(max_rec), ri(max_rec);
  std::vector<float> nref(3 * (size_t)P), nrec(3 * (size_t)max_rec);
  std::vector<uint8_t> wr(max_rec);
  int32_t n_rec = 0;
  if (dsh_normals_estimate_db(ctx, warpDB->device_records(), P, ids.data(), x0.data(), has_x0.data(), ref_uv.data(), k.data(), cov.data(), st.data(), nref.data(),
                              nullptr, max_rec, &n_rec, rp.data(), rt.data(), ri.data(), nrec.data(), wr.data()) != DSH_OK)
    return -1;
  int solved = 0;
  for (int p = 0; p < P; p++) {
    if (st[p] != 0) continue;
    solved++;
    MapPointT* mp = pts[p];
    KeyFrameT* refKF = mp->GetReferenceKeyFrame();
    for (int c4 = 0; c4 < 4; c4++) mp->covNorm[c4] = cov[4 * (size_t)p + c4];
    static_cast<DefKeyFrameT*>(refKF)->surface->setNormalSurfacePoint(mp->GetIndexInKeyFrame(refKF), &nref[3 * (size_t)p]);
  }
  for (int r = 0; r < n_rec; r++)
    if (wr[r]) static_cast<DefKeyFrameT*>(warpDB->device_tags()[rt[r]])->surface->setNormalSurfacePoint(ri[r], &nrec[3 * (size_t)r]);
  return solved;
}

}  // namespace defslam_hip
