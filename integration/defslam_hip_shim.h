// Host integration shim: the reference's call sites on top of the C ABI of libdefslam_hip.so.
//
//   DefPoseOptimizationHIP(ctx, pFrame, mMap, RegLap, RegInex, RegTemp, NeighboursLayers)
//       drop-in for defSLAM::Optimizer::DefPoseOptimization (Modules/Tracking/DefOptimizer.h:51-53, DefOptimizer.cc:251-578):
//       same arguments, same return value (nInitialCorrespondences - nBad), same in-place mutations of Frame / Node / DefMapPoint.
//
// The functions are templates over the reference's own classes: they only use members the reference declares (cited at each
// use), so inside DefSLAM they are instantiated with ORB_SLAM2::Frame / defSLAM::DefMap etc. unchanged, and the repository's CI
// instantiates them with the stand-ins of integration/standin_types.h (OpenCV / Eigen / Pangolin are not in the build image).
// The only type-specific pieces are the two pose accessors of `ShimPose<FrameT>`: cv::Mat in DefSLAM (opencv_pose.h), a
// plain float[16] in the stand-ins.
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "../include/defslam_hip.h"

namespace defslam_hip {

// pFrame->mTcw (CV_32F 4x4) <-> row-major float[16]; specialise for the frame type (opencv_pose.h does it for cv::Mat).
template <class FrameT>
struct ShimPose {
  static void get(const FrameT& f, float* T16) { for (int i = 0; i < 16; i++) T16[i] = f.mTcw[i]; }
  static void set(FrameT& f, const float* T16) { f.SetPose(T16); }
};

// The template of the map on the device: rest positions + facets in the library's numbering (nodes in the iteration order of
// Template::getNodes(), a std::set<Node*>, exactly the order setMeshNodes walks -- DefOptimizer.cc:926-952).  Rebuilt when the map
// hands out another Template object (DefLocalMapping creates a new one per template keyframe).
template <class TemplateT, class NodeT>
class TemplateBinding {
 public:
  // returns DSH_OK, or the library's status (dsh_last_error(ctx) has the text)
  int sync(dsh_ctx* ctx, TemplateT* tmpl) {
    if (tmpl == bound_ && ctx == ctx_) return DSH_OK;
    nodes_.clear();
    index_.clear();
    facet_ids_.clear();
    for (NodeT* n : tmpl->getNodes()) {                        // Template.h:87
      index_[n] = (int32_t)nodes_.size();
      nodes_.push_back(n);
    }
    std::vector<double> xyz0(3 * nodes_.size());
    for (size_t i = 0; i < nodes_.size(); i++) nodes_[i]->getInitialPose(xyz0[3 * i], xyz0[3 * i + 1], xyz0[3 * i + 2]);   // Node.h:129
    std::vector<int32_t> facets;
    for (auto* f : tmpl->getFacets())                          // Template.h:90
      for (NodeT* n : f->getNodes()) facets.push_back(index_.at(n));   // Facet.h:65 (std::set<Node*>: three nodes)
    const int rc = dsh_template_build(ctx, (int)nodes_.size(), xyz0.data(), (int)(facets.size() / 3), facets.data());
    if (rc == DSH_OK) { bound_ = tmpl; ctx_ = ctx; }
    return rc;
  }
  const std::vector<NodeT*>& nodes() const { return nodes_; }
  int32_t index_of(NodeT* n) const { return index_.at(n); }
  // node ids of a facet in the order its std::set<Node*> iterates (the order the barycentrics b1..b3 refer to, DefOptimizer.cc:315-333),
  // looked up once per facet and template instead of three std::map searches per observation and frame
  template <class FacetT>
  const int32_t* facet_nodes(FacetT* f) {
    auto it = facet_ids_.find(f);
    if (it == facet_ids_.end()) {
      FacetIds ids{{0, 0, 0}};
      int k = 0;
      for (NodeT* nd : f->getNodes()) ids.v[k++] = index_.at(nd);
      it = facet_ids_.emplace(f, ids).first;
    }
    return it->second.v;
  }

 private:
  struct FacetIds { int32_t v[3]; };
  TemplateT* bound_ = nullptr;
  dsh_ctx* ctx_ = nullptr;
  std::vector<NodeT*> nodes_;
  std::map<NodeT*, int32_t> index_;
  std::map<const void*, FacetIds> facet_ids_;
};

// Shape-from-template with camera motion estimation for one frame.  `binding` lives as long as the tracker (one per map).
template <class FrameT, class MapT, class TemplateT, class NodeT, class DefMapPointT>
int DefPoseOptimizationHIP(dsh_ctx* ctx, TemplateBinding<TemplateT, NodeT>& binding, FrameT* pFrame, MapT* mMap, double RegLap = 5000,
                           double RegInex = 5000, double RegTemp = 0, unsigned int NeighboursLayers = 1) {
  TemplateT* tmpl = mMap->GetTemplate();                       // DefMap.h:69
  if (!tmpl || binding.sync(ctx, tmpl) != DSH_OK) return 0;
  const std::vector<NodeT*>& nodes = binding.nodes();
  const int n = (int)nodes.size();
  // setMeshNodes (DefOptimizer.cc:926-952): vertex ids 1..n in set order; our node id is the vertex id minus one
  for (int i = 0; i < n; i++) nodes[i]->setIndex((unsigned)(i + 1));   // Node.h:68
  // The reference holds MapPoint::mGlobalMutex from here to the end of the function (DefOptimizer.cc:287): map points are read
  // (facets, barycentrics) and moved (RecalculatePosition) below while the mapping thread may be creating or culling them.
  std::unique_lock<decltype(DefMapPointT::mGlobalMutex)> lock(DefMapPointT::mGlobalMutex);
  // ---- observations (DefOptimizer.cc:293-361): key points that are not flagged, with a map point that is not bad and lies on a facet
  const int N = pFrame->N;
  std::vector<int32_t> obs_nodes;
  std::vector<double> obs_bary, obs_uv, obs_isig2;
  std::vector<size_t> vnIndexEdgeMono;
  std::set<NodeT*> ViewedNodes;
  int nInitialCorrespondences = 0;
  for (int i = 0; i < N; i++) {
    if (pFrame->mvbOutlier[i]) continue;
    auto* pMP = pFrame->mvpMapPoints[i];
    if (!pMP || pMP->isBad()) continue;
    DefMapPointT* dMP = static_cast<DefMapPointT*>(pMP);
    if (!dMP->getFacet()) continue;                            // DefMapPoint.h:76
    nInitialCorrespondences++;
    pFrame->mvbOutlier[i] = false;
    const double bary[3] = {dMP->b1, dMP->b2, dMP->b3};        // DefMapPoint.h:96, in the order the facet's node set iterates (:315-333)
    const int32_t* ids = binding.facet_nodes(dMP->getFacet());
    for (int k = 0; k < 3; k++) {
      NodeT* nd = nodes[ids[k]];
      obs_nodes.push_back(ids[k]);
      obs_bary.push_back(bary[k]);
      ViewedNodes.insert(nd);
      nd->setViewed();                                         // Node.h:102 (DefOptimizer.cc:332)
    }
    const auto& kpUn = pFrame->mvKeysUn[i];
    obs_uv.push_back((double)kpUn.pt.x);
    obs_uv.push_back((double)kpUn.pt.y);
    obs_isig2.push_back((double)pFrame->mvInvLevelSigma2[kpUn.octave]);   // DefOptimizer.cc:339
    vnIndexEdgeMono.push_back((size_t)i);
  }
  const int M = (int)vnIndexEdgeMono.size();
  if (M == 0) return 0;                                        // empty graph: optimize() fails, nothing changes, no inliers
  // the optimised zone gets the LOCAL role (DefOptimizer.cc:388-432): viewed nodes and, for any NeighboursLayers >= 1, their 1-ring
  if (NeighboursLayers >= 1)
    for (NodeT* v : ViewedNodes)
      for (NodeT* nb : v->GetNeighbours()) nb->setLocal();     // Node.h:99,105 (a VIEWED node keeps its role)
  for (NodeT* v : ViewedNodes) v->setLocal();
  std::vector<double> xyz(3 * (size_t)n);
  for (int i = 0; i < n; i++) nodes[i]->getXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);   // Node.h:126
  float Tcw[16];
  ShimPose<FrameT>::get(*pFrame, Tcw);
  dsh_sft_frame f{};
  f.Tcw = Tcw;
  f.K[0] = pFrame->fx; f.K[1] = pFrame->fy; f.K[2] = pFrame->cx; f.K[3] = pFrame->cy;
  f.n_frame = N;
  f.M = M;
  f.obs_nodes = obs_nodes.data(); f.obs_bary = obs_bary.data(); f.obs_uv = obs_uv.data(); f.obs_invsig2 = obs_isig2.data();
  f.xyz = xyz.data();
  f.reg_lap = RegLap; f.reg_inex = RegInex; f.reg_temp = RegTemp;
  f.neighbour_layers = (int32_t)NeighboursLayers;
  f.max_iters = 50;                                            // optimizer.optimize(50), DefOptimizer.cc:513
  std::vector<double> xyz_out(3 * (size_t)n);
  std::vector<uint8_t> outlier(M);
  float Tcw_out[16];
  dsh_sft_result r{};
  r.Tcw = Tcw_out; r.xyz = xyz_out.data(); r.outlier = outlier.data();
  if (dsh_sft_solve(ctx, &f, &r) != DSH_OK) return 0;
  // ---- write-back (DefOptimizer.cc:515-577)
  for (int e = 0; e < M; e++) pFrame->mvbOutlier[vnIndexEdgeMono[e]] = outlier[e] != 0;   // :515-537
  pFrame->repError = (float)r.rep_error;                        // :559 (Frame::repError is a float)
  ShimPose<FrameT>::set(*pFrame, Tcw_out);                      // :561-565
  for (int i = 0; i < n; i++) {                                 // updateNodes, :954-968
    nodes[i]->update();
    nodes[i]->resetRole();
    nodes[i]->setXYZ(xyz_out[3 * i], xyz_out[3 * i + 1], xyz_out[3 * i + 2]);
  }
  for (auto* pMP : mMap->GetAllMapPoints())                     // :567-575
    if (static_cast<DefMapPointT*>(pMP)->getFacet()) static_cast<DefMapPointT*>(pMP)->RecalculatePosition();
  return r.inliers;                                             // nInitialCorrespondences - nBad
}

}  // namespace defslam_hip
