// wrapper_demo.cpp — drives the header-only C++ class surface (include/dmsa_hip.hpp) the way DmsaSlam drives the reference:
//     DmsaOptimizer<PointStampId> slidingWindowOptimizer;  slidingWindowOptimizer.optimizeSet(*currTraj, settings);
// Reads a ContinuousTrajectory dumped by tests/test_cpp_wrapper.py as raw arrays, writes the optimised relative poses back.
// Exit codes: 0 ok, 2 bad arguments / files, 3 the library refused to run (e.g. no GPU: there is no CPU fallback).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

#include "dmsa_hip.hpp"

namespace {
template <class T>
bool read_all(const std::string& path, std::vector<T>& out) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    out.resize((size_t)bytes / sizeof(T));
    return bytes == 0 || (bool)f.read(reinterpret_cast<char*>(out.data()), bytes);
}
template <class T>
bool write_all(const std::string& path, const T* p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    return (bool)f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}
std::vector<std::array<float, 4>> as_points(const std::vector<float>& v) {
    std::vector<std::array<float, 4>> out(v.size() / 4);
    for (size_t k = 0; k < out.size(); ++k) out[k] = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
    return out;
}
}  // namespace

struct PointStampId;  // the reference's point type; the wrapper keeps the template parameter for source compatibility

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "usage: wrapper_demo <dump-dir> [flags]\n";
        return 2;
    }
    const std::string dir = std::string(argv[1]) + "/";
    const unsigned flags = argc > 2 ? (unsigned)std::stoul(argv[2]) : 0u;
    dmsa_hip::ContinuousTrajectory traj;
    dmsa_hip::DmsaOptimSettings settings;
    std::vector<float> local, stat;
    std::vector<double> meta;
    if (!read_all(dir + "ro.f64", traj.controlPoses.relativePoses.Orientations.v) || !read_all(dir + "rt.f64", traj.controlPoses.relativePoses.Translations.v) ||
        !read_all(dir + "stamps.f64", traj.controlPoses.stamps) || !read_all(dir + "trajtime.f64", traj.trajTime) || !read_all(dir + "local.f32", local) ||
        !read_all(dir + "tidx.i32", traj.tformIdPerPoint) || !read_all(dir + "ring.i32", traj.ringIds) || !read_all(dir + "static.f32", stat) ||
        !read_all(dir + "sring.i32", traj.staticRingIds) || !read_all(dir + "meta.f64", meta) || meta.size() < 5) {
        std::cerr << "cannot read the problem dump in " << dir << "\n";
        return 2;
    }
    traj.controlPoses.numPoses = traj.controlPoses.relativePoses.Orientations.cols();
    traj.n_total = (int)traj.trajTime.size();
    traj.localPoints = as_points(local), traj.staticPoints = as_points(stat);
    traj.minGridSize = (float)meta[0];
    settings.num_iter = (int)meta[1], settings.step_length_optim = meta[2], settings.max_step = meta[3], settings.min_num_points_per_set = (int)meta[4];
    try {
        dmsa_hip::DmsaOptimizer<PointStampId> slidingWindowOptimizer(0, flags);
        slidingWindowOptimizer.optimizeSet(traj, settings);
        const dmsa_report& rep = slidingWindowOptimizer.lastReport();
        std::printf("iterations %d stop_reason %d gaussians %d evaluations %d\n", rep.iterations, rep.stop_reason, rep.num_gaussians, rep.evaluations);
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 3;
    }
    if (!write_all(dir + "out_ro.f64", traj.controlPoses.relativePoses.Orientations.data(), traj.controlPoses.relativePoses.Orientations.v.size()) ||
        !write_all(dir + "out_rt.f64", traj.controlPoses.relativePoses.Translations.data(), traj.controlPoses.relativePoses.Translations.v.size()) ||
        !write_all(dir + "out_global.f32", traj.globalPoints.empty() ? nullptr : traj.globalPoints[0].data(), traj.globalPoints.size() * 4))
        return 2;
    return 0;
}
