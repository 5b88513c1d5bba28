// CI driver of the host integration shim: builds stand-in Frame / DefMap / Template objects from a text file, calls
// DefPoseOptimizationHIP (integration/defslam_hip_shim.h) through the C ABI of libdefslam_hip.so, writes every mutation the
// reference's call makes (SURVEY.md 8b "Ownership") to a text file, and appends the frame's row to Matches.txt.
//   usage: shim_test <input.txt> <output.txt> <Matches.txt> [device]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <iomanip>
#include <memory>

#include "defslam_hip_shim.h"
#include "result_writers.h"
#include "standin_types.h"

using namespace standin;
CountingMutex standin::MapPoint::mGlobalMutex;

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s in out matches [device]\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  if (!in.good()) return 2;
  int n, F;
  in >> n >> F;
  // nodes and facets in contiguous storage: std::set<Node*> then iterates in index order (any order works, this one is reproducible)
  std::vector<Node> nodes;
  nodes.reserve(n);
  for (int i = 0; i < n; i++) { double x, y, z; in >> x >> y >> z; nodes.emplace_back(x, y, z); }
  for (int i = 0; i < n; i++) { double x, y, z; in >> x >> y >> z; nodes[i].setXYZ(x, y, z); }
  std::vector<Facet> facets;
  facets.reserve(F);
  for (int f = 0; f < F; f++) {
    int a, b, c;
    in >> a >> b >> c;
    facets.emplace_back(&nodes[a], &nodes[b], &nodes[c]);
    const int v[3] = {a, b, c};
    for (int p = 0; p < 3; p++)
      for (int q = 0; q < 3; q++)
        if (p != q) nodes[v[p]].neighbours.insert(&nodes[v[q]]);
  }
  Template tmpl;
  for (auto& nd : nodes) tmpl.nodes.insert(&nd);
  for (auto& fc : facets) tmpl.facets.insert(&fc);
  Frame frame;
  in >> frame.fx >> frame.fy >> frame.cx >> frame.cy;
  for (int i = 0; i < 16; i++) in >> frame.mTcw[i];
  in >> frame.N >> frame.mTimeStamp;
  int levels;
  in >> levels;
  frame.mvInvLevelSigma2.resize(levels);
  for (auto& v : frame.mvInvLevelSigma2) in >> v;
  DefMap map;
  map.tmpl = &tmpl;
  std::vector<std::unique_ptr<DefMapPoint>> owned;
  frame.mvKeysUn.resize(frame.N);
  frame.mvpMapPoints.assign(frame.N, nullptr);
  frame.mvbOutlier.assign(frame.N, false);
  for (int i = 0; i < frame.N; i++) {
    int kind, facet;
    double b1, b2, b3;
    in >> frame.mvKeysUn[i].pt.x >> frame.mvKeysUn[i].pt.y >> frame.mvKeysUn[i].octave >> kind >> facet >> b1 >> b2 >> b3;
    // kind: 0 no map point, 1 map point on a facet, 2 bad map point, 3 map point without facet, 4 as 1 but flagged outlier by the caller
    if (kind == 0) continue;
    owned.emplace_back(new DefMapPoint());
    DefMapPoint* mp = owned.back().get();
    mp->bad = kind == 2;
    if (kind != 3) { mp->facet = &facets[facet]; mp->b1 = b1; mp->b2 = b2; mp->b3 = b3; }
    frame.mvpMapPoints[i] = mp;
    frame.mvbOutlier[i] = kind == 4;
    map.points.push_back(mp);
  }
  double RegLap, RegInex, RegTemp;
  unsigned layers;
  in >> RegLap >> RegInex >> RegTemp >> layers;
  int numberLocalMapPoints;
  in >> numberLocalMapPoints;
  if (!in.good()) { std::fprintf(stderr, "short input\n"); return 2; }

  dsh_ctx* ctx = nullptr;
  if (dsh_create(&ctx, argc > 4 ? std::atoi(argv[4]) : 0) != DSH_OK) { std::fprintf(stderr, "no device\n"); return 3; }
  defslam_hip::TemplateBinding<Template, Node> binding;
  // What one call leaves behind, in the layout the tests parse.
  auto dump = [&](const std::string& path, int inliers) {
    std::ofstream out(path);
    out << std::setprecision(17);
    int under_lock = 0;
    for (auto* p : map.points) under_lock += static_cast<DefMapPoint*>(p)->recalculated_under_lock;
    out << inliers << " " << frame.repError << " " << frame.pose_sets << " " << MapPoint::mGlobalMutex.locks << " " << (int)MapPoint::mGlobalMutex.held << " " << under_lock << "\n";
    for (int i = 0; i < 16; i++) out << frame.mTcw[i] << (i == 15 ? "\n" : " ");
    for (int i = 0; i < n; i++) {
      double x, y, z;
      nodes[i].getXYZ(x, y, z);
      out << x << " " << y << " " << z << " " << nodes[i].getIndex() << " " << (int)nodes[i].viewed << " " << (int)nodes[i].local << " " << (int)nodes[i].role << "\n";
    }
    for (int i = 0; i < frame.N; i++) out << (int)frame.mvbOutlier[i] << (i + 1 == frame.N ? "\n" : " ");
    for (auto* p : map.points) {
      auto* mp = static_cast<DefMapPoint*>(p);
      out << mp->mWorldPos[0] << " " << mp->mWorldPos[1] << " " << mp->mWorldPos[2] << " " << mp->recalculated << "\n";
    }
  };
  const bool switch_frame = argc > 5 && std::string(argv[5]) == "switch";
  int inliers;
  if (switch_frame) {
    // The frame right behind a template switch (DefTracking.cc:109-123): DefPoseOptimization with RegTemp = 0, then -- TrackLocalMap,
    // :244-247 -- the regular call on the SAME frame: it starts from the pose and the nodes the first call wrote and skips the key points
    // the first call flagged (pFrame->mvbOutlier, DefOptimizer.cc:295).
    inliers = defslam_hip::DefPoseOptimizationHIP<Frame, DefMap, Template, Node, DefMapPoint>(ctx, binding, &frame, &map, RegLap, RegInex, 0.0, layers);
    dump(std::string(argv[2]) + ".first", inliers);
  }
  inliers = defslam_hip::DefPoseOptimizationHIP<Frame, DefMap, Template, Node, DefMapPoint>(ctx, binding, &frame, &map, RegLap, RegInex, RegTemp, layers);
  dump(argv[2], inliers);
  defslam_hip::MatchesWriter mw(argv[3]);
  mw.add_frame(frame, numberLocalMapPoints);
  std::fprintf(stderr, "%s\n", dsh_last_error(ctx));
  dsh_destroy(ctx);
  return 0;
}
