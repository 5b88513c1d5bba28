// CI driver of integration/schwarp_database_hip.h: keyframes and map points from a text file, SchwarpDatabaseHIP::add for every
// keyframe in turn, ObtainK1K2HIP, and a dump of everything the reference's calls would have changed.
//   usage: mapping_shim_test <input.txt> <output.txt> [device] [devrec]     (devrec: the DiffProp records stay in HBM, dsh_diffdb)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <string>

#include "schwarp_database_hip.h"
#include "standin_mapping_types.h"

using namespace standin;
typedef defslam_hip::SchwarpDatabaseHIP<WarpDatabase, KeyFrame, KeyFrame, MapPointM, DiffProp> DB;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::ifstream in(argv[1]);
  int nKF, P, NCu, NCv, levels;
  double lambda;
  in >> nKF >> P >> NCu >> NCv >> lambda >> levels;
  std::vector<float> sig2(levels);
  for (auto& v : sig2) in >> v;
  std::vector<std::unique_ptr<KeyFrame>> kfs;
  std::vector<std::unique_ptr<MapPointM>> mps;
  for (int p = 0; p < P; p++) { mps.emplace_back(new MapPointM()); mps.back()->id = p; }
  for (int k = 0; k < nKF; k++) {
    int N;
    in >> N;
    kfs.emplace_back(new KeyFrame(N, NCu, NCv));
    KeyFrame& kf = *kfs.back();
    in >> kf.umin >> kf.umax >> kf.vmin >> kf.vmax >> kf.fx >> kf.fy >> kf.cx >> kf.cy >> kf.mnMinX >> kf.mnMaxX >> kf.mnMinY >> kf.mnMaxY;
    kf.mvInvLevelSigma2 = sig2;
    kf.mvKeysUn.resize(N);
    kf.mpKeypointNorm.resize(N);
    kf.desc.resize(32 * (size_t)N);
    for (int i = 0; i < N; i++) {
      int mp;
      in >> kf.mvKeysUn[i].pt.x >> kf.mvKeysUn[i].pt.y >> kf.mvKeysUn[i].octave >> kf.mpKeypointNorm[i].pt.x >> kf.mpKeypointNorm[i].pt.y >> mp;
      kf.mpKeypointNorm[i].octave = kf.mvKeysUn[i].octave;
      for (int b = 0; b < 32; b++) { int v; in >> v; kf.desc[32 * (size_t)i + b] = (uint8_t)v; }
      if (mp >= 0) { kf.mvpMapPoints[i] = mps[mp].get(); mps[mp]->AddObservation(&kf, i); if (k == 0) mps[mp]->refKF = &kf; }
    }
  }
  if (!in.good()) { std::fprintf(stderr, "short input\n"); return 2; }
  dsh_ctx* ctx = nullptr;
  if (dsh_create(&ctx, argc > 3 ? std::atoi(argv[3]) : 0) != DSH_OK) return 3;
  DB db(ctx, lambda);
  const bool devrec = argc > 4 && std::string(argv[4]) == "devrec";
  if (devrec && !db.enable_device_records(1 << 16)) { std::fprintf(stderr, "dsh_diffdb_create failed: %s\n", dsh_last_error(ctx)); return 5; }
  for (auto& kf : kfs) {
    db.add(kf.get());
    if (db.last_status() != DSH_OK) { std::fprintf(stderr, "add failed: %s\n", dsh_last_error(ctx)); return 4; }
  }
  std::ofstream out(argv[2]);
  out << std::setprecision(9);
  // the DiffProp database, per map point id
  size_t nrec = 0;
  for (auto& pr : db.getDiffDatabase()) nrec += pr.second.size();
  if (devrec) { out << dsh_diffdb_count(db.device_records()) << " " << nrec << "\n"; }   // (records on the device, records on the host: 0)
  else out << nrec << "\n";
  for (int p = 0; p < P; p++) {
    auto it = db.getDiffDatabase().find(mps[p].get());
    if (it == db.getDiffDatabase().end()) continue;
    for (auto& d : it->second) {
      int k2 = -1;
      for (int k = 0; k < nKF; k++) if (kfs[k].get() == d->KFToKF.second) k2 = k;
      out << p << " " << k2 << " " << d->idx1 << " " << d->idx2 << " " << d->I1u << " " << d->I1v << " " << d->I2u << " " << d->I2v << " " << d->J12a << " " << d->J12b << " "
          << d->J12c << " " << d->J12d << " " << d->J21a << " " << d->J21b << " " << d->J21c << " " << d->J21d << " " << d->H12uux << " " << d->H12uuy << " " << d->H12uvx
          << " " << d->H12uvy << " " << d->H12vvx << " " << d->H12vvy << "\n";
    }
  }
  // matches per keyframe after the database ran (new matches registered, bad ones erased)
  for (int k = 0; k < nKF; k++) {
    out << kfs[k]->KeyframesRelated;
    for (int i = 0; i < kfs[k]->N; i++) out << " " << (kfs[k]->mvpMapPoints[i] ? kfs[k]->mvpMapPoints[i]->id : -1);
    out << "\n";
  }
  // argv[5] = n: the first n map points that have records are re-anchored in another keyframe that observes them before the normals are solved
  // (what MapPoint::EraseObservation does when the reference keyframe loses the point)
  int reanchor = argc > 5 ? std::atoi(argv[5]) : 0, reanchored = 0;
  for (int p = 0; p < P && reanchored < reanchor; p++) {
    MapPointM* mp = mps[p].get();
    auto flag = db.getToProccess().find(mp);
    if (flag == db.getToProccess().end() || !flag->second) continue;   // (only points that received a record)
    for (int k = 0; k < nKF; k++)
      if (kfs[k].get() != mp->refKF && mp->GetIndexInKeyFrame(kfs[k].get()) >= 0) { mp->refKF = kfs[k].get(); reanchored++; break; }
  }
  int skipped = 0;
  const int solved = devrec ? defslam_hip::ObtainK1K2DeviceHIP<DB, KeyFrame, KeyFrame, MapPointM>(ctx, &db, &skipped)
                            : defslam_hip::ObtainK1K2HIP<DB, KeyFrame, KeyFrame, MapPointM>(ctx, &db);
  out << solved << "\n";
  if (reanchor) std::fprintf(stderr, "reanchored %d skipped %d\n", reanchored, skipped);
  for (int k = 0; k < nKF; k++) {
    out << kfs[k]->surface->writes << "\n";
    for (int i = 0; i < kfs[k]->N; i++) {
      float N[3] = {0, 0, 0};
      const bool h = kfs[k]->surface->getNormalSurfacePoint(i, N);
      out << (int)h << " " << N[0] << " " << N[1] << " " << N[2] << "\n";
    }
  }
  out << std::setprecision(17);
  for (int p = 0; p < P; p++) out << mps[p]->covNorm[0] << " " << mps[p]->covNorm[1] << " " << mps[p]->covNorm[2] << " " << mps[p]->covNorm[3] << "\n";
  int pending = 0;
  for (auto& pr : db.getToProccess()) pending += pr.second;
  out << pending << "\n";
  dsh_destroy(ctx);
  return 0;
}
