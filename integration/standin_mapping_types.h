// Stand-ins of the reference's mapping classes for the CI of integration/schwarp_database_hip.h (see standin_types.h): only the
// members the shim uses, with the reference's names.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <utility>
#include <vector>

#include "standin_types.h"

namespace standin {

class KeyFrame;

class Surface {                                // Modules/Mapping/Surface.h: per key point normal storage
 public:
  explicit Surface(size_t n) : normals(3 * n, 0.f), has(n, 0) {}
  bool getNormalSurfacePoint(size_t i, float* N) { if (!has[i]) return false; for (int k = 0; k < 3; k++) N[k] = normals[3 * i + k]; return true; }
  void setNormalSurfacePoint(size_t i, const float* N) { for (int k = 0; k < 3; k++) normals[3 * i + k] = N[k]; has[i] = 1; writes++; }
  std::vector<float> normals;
  std::vector<uint8_t> has;
  int writes = 0;
};

class MapPointM {                              // ORB_SLAM2::MapPoint as the mapping thread sees it
 public:
  bool isBad() { return bad; }
  KeyFrame* GetReferenceKeyFrame() { return refKF; }
  bool IsInKeyFrame(KeyFrame* kf) { return obs.count(kf) != 0; }
  int GetIndexInKeyFrame(KeyFrame* kf) { auto it = obs.find(kf); return it == obs.end() ? -1 : (int)it->second; }
  void AddObservation(KeyFrame* kf, size_t idx) { obs[kf] = idx; }
  void EraseObservation(KeyFrame* kf) { obs.erase(kf); }
  bool bad = false;
  KeyFrame* refKF = nullptr;
  std::map<KeyFrame*, size_t> obs;
  double covNorm[4] = {0, 0, 0, 0};            // MapPoint::covNorm (NormalEstimator.cc:159)
  int id = -1;
};

class KeyFrame {                               // ORB_SLAM2::KeyFrame + defSLAM::DefKeyFrame members (one class is enough for the stand-in)
 public:
  KeyFrame(int N_, int NCu_, int NCv_) : N(N_), NCu(NCu_), NCv(NCv_), mvpMapPoints(N_, nullptr), surface(new Surface(N_)) {}
  std::vector<MapPointM*> GetMapPointMatches() { return mvpMapPoints; }
  MapPointM* GetMapPoint(size_t i) { return mvpMapPoints[i]; }
  void addMapPoint(MapPointM* mp, size_t i) { mvpMapPoints[i] = mp; }
  void EraseMapPointMatch(size_t i) { mvpMapPoints[i] = nullptr; }
  const uint8_t* descriptor(size_t i) const { return &desc[32 * i]; }        // mDescriptors.ptr(i)
  int gridCols() const { return 64; }                                        // FRAME_GRID_COLS
  int gridRows() const { return 48; }                                        // FRAME_GRID_ROWS
  int N, NCu, NCv, valdim = 2, KeyframesRelated = 0;
  double umin = 0, umax = 0, vmin = 0, vmax = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
  std::vector<KeyPoint> mvKeysUn, mpKeypointNorm;
  std::vector<float> mvInvLevelSigma2;
  std::vector<uint8_t> desc;
  std::vector<MapPointM*> mvpMapPoints;
  std::unique_ptr<Surface> surface;
};

struct DiffProp {                              // Modules/Mapping/diffProp.h:52-83
  std::pair<KeyFrame*, KeyFrame*> KFToKF;
  size_t idx1 = 0, idx2 = 0;
  float I1u, I1v, I2u, I2v, J12a, J12b, J12c, J12d, J21a, J21b, J21c, J21d, H12uux, H12uuy, H12uvx, H12uvy, H12vvx, H12vvy;
};

class WarpDatabase {                           // Modules/Mapping/WarpDatabase.h:39-72
 public:
  typedef std::vector<std::shared_ptr<DiffProp>> kr2krdata;
  virtual ~WarpDatabase() = default;
  virtual void add(KeyFrame* kf) = 0;
  virtual void erase(KeyFrame* kf) = 0;
  virtual void clear() = 0;
  std::map<MapPointM*, kr2krdata>& getDiffDatabase() { return mapPointsDB_; }
  std::map<MapPointM*, bool>& getToProccess() { return newInformation_; }

 protected:
  std::map<MapPointM*, bool> newInformation_;
  std::map<MapPointM*, kr2krdata> mapPointsDB_;
};

}  // namespace standin
