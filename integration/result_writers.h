// On-disk outputs of the reference that its tuning script reads back (scripts/Twiddle.py:38-131): Matches.txt and ErrorGTs<frame>.txt.
// Plain C++11, no dependencies -- part of the host integration shim, linked next to libdefslam_hip.so.
#pragma once
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

namespace defslam_hip {

// One row per tracked frame: "<timestamp, 5 digits, zero filled> <inliers> <outliers> <local map points>" -- the line
// DefTracking::TrackLocalMap appends to this->matches (Modules/Tracking/DefTracking.cc:299-328).  Twiddle.py reads the file with
// sep=' ', names=['frame', 'inliers', 'outliers', 'possibleMatches'] and sums inliers / possibleMatches.
class MatchesWriter {
 public:
  explicit MatchesWriter(const std::string& path) : out_(path.c_str()) {}
  bool ok() const { return out_.good(); }
  // Counts exactly like DefTracking.cc:300-318: key points with a map point that is not bad; inlier = !mvbOutlier[i].
  template <class FrameT>
  void add_frame(const FrameT& frame, int numberLocalMapPoints) {
    int mI = 0, mO = 0;
    for (int i = 0; i < frame.N; i++) {
      if (!frame.mvpMapPoints[i]) continue;
      if (frame.mvpMapPoints[i]->isBad()) continue;
      if (!frame.mvbOutlier[i]) mI++; else mO++;
    }
    add_row((unsigned int)frame.mTimeStamp, mI, mO, numberLocalMapPoints);
  }
  void add_row(unsigned int timestamp, int inliers, int outliers, int numberLocalMapPoints) {
    char buf[96];
    std::snprintf(buf, sizeof buf, "%05u %d %d %d", timestamp, inliers, outliers, numberLocalMapPoints);   // setfill('0') << setw(5)
    out_ << buf << std::endl;
  }

 private:
  std::ofstream out_;
};

// ErrorGTs<frame>.txt: the per-point 3D errors of a frame, one per line, as GroundTruthTools::saveResults writes them
// (Modules/GroundTruth/GroundTruthCalculator.cc:174-186: an Eigen column matrix streamed with the default format -- six
// significant digits, entries right-aligned to the widest one, no trailing newline).  Twiddle.py reads it with
// pd.read_csv(header=None).transpose() and averages all entries (x 1000: metres -> millimetres).
std::string error_gts_name(const std::string& output_path, unsigned int timestamp);   // output_path + "/ErrorGTs" + %05u + ".txt" (GroundTruthFrame.cc:258-262)
bool save_results(const std::vector<float>& errors, const std::string& name);

// per-point error of GroundTruthFrame::Estimate3DError (GroundTruthFrame.cc:243-252): | stereo - s * mono |
std::vector<float> surface_errors(const std::vector<std::vector<float>>& posMono, const std::vector<std::vector<float>>& posStereo, double s);

}  // namespace defslam_hip
