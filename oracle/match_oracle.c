/*
 * match_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never on the product path).
 *
 * Restatement of the warp-guided match search (second half):
 *   DefORBmatcher::searchBySchwarp ........ Modules/Matching/DefORBmatcher.cc:189-294 (prediction, window, best descriptor)
 *   KeyFrame::GetFeaturesInArea ............ Thirdparty/ORBSLAM_2/src/KeyFrame.cc:618-663 (grid cells, iteration order)
 *   Frame::PosInGrid / grid construction ... Thirdparty/ORBSLAM_2/src/Frame.cc:296-308, 484-496
 *   KeyFrame::IsInImage ..................... Thirdparty/ORBSLAM_2/src/KeyFrame.cc:665-668
 *   ORBmatcher::DescriptorDistance .......... Thirdparty/ORBSLAM_2/src/ORBmatcher.cc:1691-1707 (256-bit Hamming)
 * It keeps the reference's SHAPE (grid of index lists, cells walked column by column, first strictly better candidate
 * wins) so that the device code -- a brute-force scan with an explicit tie-break key -- is checked against a different
 * formulation.  Integer/index work: the parity bar is bit-exact.
 * PARITY UNPINNED: these reference files need OpenCV and cannot be compiled here; the logic is integer bookkeeping and
 * float32 comparisons restated literally.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void bbs_oracle_eval(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, int valdim, const double* ctrl, const double* u, const double* v,
                     int n, int du, int dv, double* val, uint8_t* status);

static int hamming256(const uint8_t* a, const uint8_t* b) {
  const uint32_t* pa = (const uint32_t*)a;
  const uint32_t* pb = (const uint32_t*)b;
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t v = pa[i] ^ pb[i];
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}

/* x[2N]: control points of the warp (first coordinate of all, then the second); kp1: Q normalised key points (float32 pairs);
 * desc1 / desc2: 32-byte ORB descriptors; cam2 = {fx, fy, cx, cy}; bounds2 = {mnMinX, mnMaxX, mnMinY, mnMaxY};
 * kp2: N2 undistorted key points in pixels; has_mp2[j] != 0 when key point j of keyframe 2 already has a map point.
 * match[q] = index in keyframe 2 or -1.  Returns the number of matches. */
int match_oracle_search_by_schwarp(double umin, double umax, int nptsu, double vmin, double vmax, int nptsv, const double* x, int Q, const float* kp1,
                                   const uint8_t* desc1, const float* cam2, const float* bounds2, int grid_cols, int grid_rows, int N2, const float* kp2,
                                   const uint8_t* desc2, const uint8_t* has_mp2, float radius, int th_low, int32_t* match) {
  const int N = nptsu * nptsv;
  const float minX = bounds2[0], maxX = bounds2[1], minY = bounds2[2], maxY = bounds2[3];
  const float winv = (float)grid_cols / (maxX - minX), hinv = (float)grid_rows / (maxY - minY);
  /* grid of index lists, filled in key point order (Frame.cc:296-308) */
  int* count = (int*)calloc((size_t)grid_cols * grid_rows, sizeof(int));
  int* cell_of = (int*)malloc(sizeof(int) * (size_t)N2);
  for (int j = 0; j < N2; j++) {
    const int px = (int)roundf((kp2[2 * j] - minX) * winv), py = (int)roundf((kp2[2 * j + 1] - minY) * hinv);
    cell_of[j] = (px < 0 || px >= grid_cols || py < 0 || py >= grid_rows) ? -1 : px * grid_rows + py;
    if (cell_of[j] >= 0) count[cell_of[j]]++;
  }
  int* start = (int*)malloc(sizeof(int) * ((size_t)grid_cols * grid_rows + 1));
  start[0] = 0;
  for (int c = 0; c < grid_cols * grid_rows; c++) start[c + 1] = start[c] + count[c];
  int* fill = (int*)calloc((size_t)grid_cols * grid_rows, sizeof(int));
  int* items = (int*)malloc(sizeof(int) * (size_t)(N2 > 0 ? N2 : 1));
  for (int j = 0; j < N2; j++)
    if (cell_of[j] >= 0) items[start[cell_of[j]] + fill[cell_of[j]]++] = j;
  /* predictions: Warp::getEstimates = BBS eval of both coordinates, stored as float32 key points */
  double* ctrl = (double*)malloc(sizeof(double) * 2 * (size_t)N);
  for (int i = 0; i < N; i++) { ctrl[2 * i] = x[i]; ctrl[2 * i + 1] = x[N + i]; }
  double* u = (double*)malloc(sizeof(double) * (size_t)Q);
  double* v = (double*)malloc(sizeof(double) * (size_t)Q);
  double* val = (double*)malloc(sizeof(double) * 2 * (size_t)Q);
  for (int q = 0; q < Q; q++) { u[q] = kp1[2 * q]; v[q] = kp1[2 * q + 1]; }
  bbs_oracle_eval(umin, umax, nptsu, vmin, vmax, nptsv, 2, ctrl, u, v, Q, 0, 0, val, 0);
  int nmatches = 0;
  for (int q = 0; q < Q; q++) {
    match[q] = -1;
    const float ex = (float)val[2 * q], ey = (float)val[2 * q + 1];
    const float px = ex * cam2[0] + cam2[2], py = ey * cam2[1] + cam2[3];
    if (!(px >= minX && px < maxX && py >= minY && py < maxY)) continue;
    const int c0 = (int)floorf((px - minX - radius) * winv) > 0 ? (int)floorf((px - minX - radius) * winv) : 0;
    if (c0 >= grid_cols) continue;
    int c1 = (int)ceilf((px - minX + radius) * winv);
    if (c1 > grid_cols - 1) c1 = grid_cols - 1;
    if (c1 < 0) continue;
    const int r0 = (int)floorf((py - minY - radius) * hinv) > 0 ? (int)floorf((py - minY - radius) * hinv) : 0;
    if (r0 >= grid_rows) continue;
    int r1 = (int)ceilf((py - minY + radius) * hinv);
    if (r1 > grid_rows - 1) r1 = grid_rows - 1;
    if (r1 < 0) continue;
    int best = th_low, best_j = -1;
    for (int ix = c0; ix <= c1; ix++)
      for (int iy = r0; iy <= r1; iy++) {
        const int c = ix * grid_rows + iy;
        for (int t = start[c]; t < start[c + 1]; t++) {
          const int j = items[t];
          const float dx = kp2[2 * j] - px, dy = kp2[2 * j + 1] - py;
          if (!(fabsf(dx) < radius && fabsf(dy) < radius)) continue;
          if (has_mp2[j]) continue;
          const int d = hamming256(desc1 + 32 * (size_t)q, desc2 + 32 * (size_t)j);
          if (d < th_low && d < best) { best = d; best_j = j; }
        }
      }
    if (best_j >= 0) { match[q] = best_j; nmatches++; }
  }
  free(count); free(cell_of); free(start); free(fill); free(items); free(ctrl); free(u); free(v); free(val);
  return nmatches;
}
