// shim_driver.cpp — test driver for the C++ drop-in (visfs_amd/host/Optimizer.{h,cpp}).
// Reads a window dumped by tests/test_cpp_shim.py, fills the reference's std::map containers, calls
// VISFS::Optimizer::Optimizer::localOptimize and dumps what came back.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Optimizer.h"

template <typename T>
static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(3); }
    return v;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: shim_driver in.bin out.bin [Key=Value ...]\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    auto hdr = rd<int64_t>(f, 7);      // root_id n_poses n_links n_points n_refs n_cameras n_laser
    auto cam = rd<double>(f, 5);       // fx fy cx cy baseline
    auto trc = rd<double>(f, 12);
    const size_t Np = hdr[1], Nk = hdr[2], Nl = hdr[3], Nr = hdr[4];
    auto pose_ids = rd<uint64_t>(f, Np); auto pose_T = rd<double>(f, Np * 12);
    auto lf = rd<uint64_t>(f, Nk); auto lt = rd<uint64_t>(f, Nk); auto lT = rd<double>(f, Nk * 12);
    auto pid = rd<uint64_t>(f, Nl); auto pxyz = rd<double>(f, Nl * 3); auto pfix = rd<uint8_t>(f, Nl);
    auto rf = rd<uint64_t>(f, Nr); auto rp = rd<uint64_t>(f, Nr);
    auto ru = rd<float>(f, Nr); auto rv = rd<float>(f, Nr); auto rdep = rd<float>(f, Nr);
    // laser part: resolution max_x max_y | nx ny | cost[ny][nx] | xyz[n_laser][3]
    std::vector<double> glim, lxyz; std::vector<int64_t> gdim; std::vector<float> gcost;
    if (hdr[6] > 0) { glim = rd<double>(f, 3); gdim = rd<int64_t>(f, 2); gcost = rd<float>(f, (size_t)(gdim[0] * gdim[1])); lxyz = rd<double>(f, (size_t)hdr[6] * 3); }
    std::fclose(f);

    VISFS::ParametersMap prm;
    for (int i = 3; i < argc; ++i) { std::string kv(argv[i]); size_t eq = kv.find('='); if (eq != std::string::npos) prm[kv.substr(0, eq)] = kv.substr(eq + 1); }

    auto iso = [](const double* m) { Eigen::Isometry3d T = Eigen::Isometry3d::Identity(); for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) T(r, c) = m[4 * r + c]; return T; };
    std::map<std::size_t, Eigen::Isometry3d> poses;
    for (size_t i = 0; i < Np; ++i) poses.emplace(pose_ids[i], iso(&pose_T[12 * i]));
    std::map<std::size_t, std::tuple<std::size_t, std::size_t, Eigen::Isometry3d>> links;
    for (size_t i = 0; i < Nk; ++i) links.emplace(std::piecewise_construct, std::forward_as_tuple(i + 1), std::forward_as_tuple(lf[i], lt[i], iso(&lT[12 * i])));
    std::vector<std::shared_ptr<VISFS::GeometricCamera>> cams;
    for (int64_t c = 0; c < hdr[5]; ++c) {
        auto cm = std::make_shared<VISFS::GeometricCamera>();
        cm->set(cam[0], cam[1], cam[2], cam[3], (float)cam[4]);
        cm->setTransformImageToRobot(iso(trc.data()));
        cams.push_back(cm);
    }
    std::map<std::size_t, std::tuple<Eigen::Vector3d, bool>> points;
    for (size_t i = 0; i < Nl; ++i) points.emplace(pid[i], std::make_tuple(Eigen::Vector3d(pxyz[3 * i], pxyz[3 * i + 1], pxyz[3 * i + 2]), pfix[i] != 0));
    std::map<std::size_t, std::map<std::size_t, VISFS::Optimizer::FeatureBA>> refs;
    for (size_t i = 0; i < Nr; ++i) refs[rf[i]].emplace(rp[i], VISFS::Optimizer::FeatureBA(cv::KeyPoint(ru[i], rv[i], 1.f), rdep[i]));
    std::vector<VISFS::Sensor::PointCloud> clouds;
    std::shared_ptr<const VISFS::Map::Submap2D> submap;
    if (hdr[6] > 0) {
        clouds.resize(2);                                 // two clouds: the factor concatenates them (Optimizer.cpp:235)
        for (int64_t i = 0; i < hdr[6]; ++i) { VISFS::Sensor::RangefinderPoint p; p.position = Eigen::Vector3d(lxyz[3 * i], lxyz[3 * i + 1], lxyz[3 * i + 2]); clouds[i < hdr[6] / 2 ? 0 : 1].pts_.push_back(p); }
        VISFS::Map::CellLimits cl; cl.numXcells = (int)gdim[0]; cl.numYcells = (int)gdim[1];
        auto grid = std::make_shared<VISFS::Map::Grid2D>(VISFS::Map::MapLimits(glim[0], Eigen::Vector2d(glim[1], glim[2]), cl), gcost);
        submap = std::make_shared<VISFS::Map::Submap2D>(grid);
    }
    std::vector<std::tuple<std::size_t, std::size_t>> outliers;
    outliers.emplace_back(123456, 654321);          // pre-existing entry: localOptimize must APPEND

    VISFS::Optimizer::Optimizer opt(prm);
    auto out = opt.localOptimize((std::size_t)hdr[0], poses, links, cams, points, refs, clouds, submap, outliers);

    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 2;
    int64_t h2[4] = { opt.lastStatus(), (int64_t)out.size(), (int64_t)points.size(), (int64_t)outliers.size() };
    std::fwrite(h2, sizeof(int64_t), 4, o);
    for (auto& kv : out) { uint64_t id = kv.first; std::fwrite(&id, 8, 1, o); double m[12]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m[4 * r + c] = kv.second(r, c); std::fwrite(m, 8, 12, o); }
    for (auto& kv : points) { double p[3] = { std::get<0>(kv.second)[0], std::get<0>(kv.second)[1], std::get<0>(kv.second)[2] }; std::fwrite(p, 8, 3, o); }
    for (auto& t : outliers) { uint64_t a = std::get<0>(t), b = std::get<1>(t); std::fwrite(&a, 8, 1, o); std::fwrite(&b, 8, 1, o); }
    std::fclose(o);
    return 0;
}
