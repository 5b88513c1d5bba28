// Stand-ins of the reference classes the shim touches, for the repository's CI (the image has no OpenCV / Eigen / Pangolin, so
// ORB_SLAM2::Frame etc. cannot be compiled here).  Each class declares the members the shim uses with the reference's names and
// semantics (file:line of the reference declaration behind each); nothing else.  Inside DefSLAM these headers are not used.
#pragma once
#include <array>
#include <cmath>
#include <mutex>
#include <set>
#include <vector>

namespace standin {

class Node {                                   // Modules/Template/Node.h
 public:
  enum Role { NONOBS, LOCAL, VIEWED };
  Node(double x, double y, double z) : x(x), y(y), z(z), xO(x), yO(y), zO(z) {}
  unsigned getIndex() { return indx; }                                   // :65
  void setIndex(unsigned i) { indx = i; }                                // :68
  std::set<Node*> GetNeighbours() { return neighbours; }                 // :99
  void setViewed() { role = VIEWED; }                                    // :102, Node.cc:132-136
  void setLocal() { if (role != VIEWED) role = LOCAL; }                  // :105, Node.cc:138-143
  void update() {                                                        // :108, Node.cc:83-101
    if (role == VIEWED) { local = false; viewed = true; }
    else if (role == LOCAL) { viewed = false; local = true; }
    else { viewed = false; local = false; }
  }
  void resetRole() { role = NONOBS; }                                    // :117
  void setXYZ(double xx, double yy, double zz) { x = xx; y = yy; z = zz; }   // :123
  void getXYZ(double& xx, double& yy, double& zz) { xx = x; yy = y; zz = z; }   // :126
  void getInitialPose(double& xx, double& yy, double& zz) { xx = xO; yy = yO; zz = zO; }   // :129
  std::set<Node*> neighbours;
  double x, y, z, xO, yO, zO;
  unsigned indx = 0;
  Role role = NONOBS;
  bool viewed = false, local = false;
};

class Facet {                                  // Modules/Template/Facet.h
 public:
  Facet(Node* a, Node* b, Node* c) : nodes{a, b, c} {}
  std::set<Node*> getNodes() { return nodes; }                           // :65
  std::set<Node*> nodes;
};

class Template {                               // Modules/Template/Template.h
 public:
  const std::set<Node*> getNodes() { return nodes; }                     // :87
  const std::set<Facet*> getFacets() { return facets; }                  // :90
  std::set<Node*> nodes;
  std::set<Facet*> facets;
};

// std::mutex that counts its lock() calls and can tell whether it is held: lets the CI check that the shim takes
// MapPoint::mGlobalMutex once per call and holds it while map points are moved (DefOptimizer.cc:287)
class CountingMutex {
 public:
  void lock() { m_.lock(); locks++; held = true; }
  void unlock() { held = false; m_.unlock(); }
  bool try_lock() { if (!m_.try_lock()) return false; locks++; held = true; return true; }
  int locks = 0;
  bool held = false;
 private:
  std::mutex m_;
};

class MapPoint {                               // ORB_SLAM2 MapPoint
 public:
  virtual ~MapPoint() = default;
  bool isBad() { return bad; }
  bool bad = false;
  static CountingMutex mGlobalMutex;           // MapPoint.h:115 (static std::mutex mGlobalMutex)
};

class DefMapPoint : public MapPoint {          // Modules/Common/DefMapPoint.h
 public:
  Facet* getFacet() { return facet; }                                    // :76
  void RecalculatePosition() {                                           // :86, DefMapPoint.cc:129-147: float32 world position from the facet nodes
    int k = 0;
    const double b[3] = {b1, b2, b3};
    double p[3] = {0, 0, 0};
    for (Node* n : facet->getNodes()) {
      double x, y, z;
      n->getXYZ(x, y, z);
      p[0] += b[k] * x; p[1] += b[k] * y; p[2] += b[k] * z;
      k++;
    }
    for (int c = 0; c < 3; c++) mWorldPos[c] = (float)p[c];
    recalculated++;
    if (mGlobalMutex.held) recalculated_under_lock++;
  }
  Facet* facet = nullptr;
  double b1 = 0, b2 = 0, b3 = 0;                                         // :96
  float mWorldPos[3] = {0, 0, 0};
  int recalculated = 0, recalculated_under_lock = 0;
};

struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; int octave; };  // cv::KeyPoint

class Frame {                                  // ORB_SLAM2 Frame (+ DefSLAM's repError)
 public:
  void SetPose(const float* T16) { for (int i = 0; i < 16; i++) mTcw[i] = T16[i]; pose_sets++; }
  int N = 0;
  double mTimeStamp = 0;
  std::vector<KeyPoint> mvKeysUn;
  std::vector<float> mvInvLevelSigma2;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  float mTcw[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // cv::Mat CV_32F in the reference
  float repError = 0;
  int pose_sets = 0;
};

class DefMap {                                 // Modules/Common/DefMap.h
 public:
  Template* GetTemplate() { return tmpl; }                               // :69
  std::vector<MapPoint*> GetAllMapPoints() { return points; }
  Template* tmpl = nullptr;
  std::vector<MapPoint*> points;
};

}  // namespace standin
