#include "result_writers.h"

#include <cmath>
#include <cstring>

namespace defslam_hip {

std::string error_gts_name(const std::string& output_path, unsigned int timestamp) {
  char buf[32];
  std::snprintf(buf, sizeof buf, "%05u", timestamp);
  return output_path + "/ErrorGTs" + buf + ".txt";
}

bool save_results(const std::vector<float>& errors, const std::string& name) {
  // Eigen's operator<< with IOFormat(): every coefficient formatted with the stream's default precision (6 significant digits,
  // %g style), then padded on the left to the width of the widest coefficient; rows separated by '\n'.
  std::vector<std::string> cells;
  size_t width = 0;
  for (float e : errors) {
    char buf[64];
    std::snprintf(buf, sizeof buf, "%g", (double)e);
    cells.emplace_back(buf);
    width = std::max(width, cells.back().size());
  }
  std::ofstream f(name.c_str());
  if (!f.good()) return false;
  for (size_t i = 0; i < cells.size(); i++) {
    if (i) f << "\n";
    f << std::string(width - cells[i].size(), ' ') << cells[i];
  }
  f.close();
  return f.good();
}

std::vector<float> surface_errors(const std::vector<std::vector<float>>& posMono, const std::vector<std::vector<float>>& posStereo, double s) {
  std::vector<float> err;
  err.reserve(posMono.size());
  for (size_t i = 0; i < posMono.size(); i++) {
    const double er = std::sqrt(std::pow(posStereo[i][0] - s * posMono[i][0], 2) + std::pow(posStereo[i][1] - s * posMono[i][1], 2) +
                                std::pow(posStereo[i][2] - s * posMono[i][2], 2));
    err.push_back((float)er);
  }
  return err;
}

}  // namespace defslam_hip
