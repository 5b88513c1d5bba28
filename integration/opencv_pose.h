// Pose accessors of the shim for the real ORB_SLAM2::Frame (cv::Mat mTcw, CV_32F 4x4).  Include this header instead of relying on
// the primary template when building inside DefSLAM (it needs OpenCV, which the repository's CI image does not have).
#pragma once
#include <opencv2/core.hpp>

#include "defslam_hip_shim.h"

namespace ORB_SLAM2 { class Frame; }

namespace defslam_hip {
template <>
struct ShimPose<ORB_SLAM2::Frame> {
  template <class F>
  static void get(const F& f, float* T16) {
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) T16[4 * r + c] = f.mTcw.template at<float>(r, c);
  }
  template <class F>
  static void set(F& f, const float* T16) { f.SetPose(cv::Mat(4, 4, CV_32F, const_cast<float*>(T16)).clone()); }   // DefOptimizer.cc:563-565
};
}  // namespace defslam_hip
