// shim_check.cpp -- integration/lsgpu_icp_shim.hpp instantiated with the in-tree mirror types: the shim a maintainer
// drops into laser_track.hpp:217 must give the SAME transform as laser_slam_amd::ICP on the same clouds (bit for bit:
// both end in lsgpu_icp_compute), map LSGPU_NO_CONVERGENCE to the PointMatcher exception, and its filter twin must
// thin the descriptors along with the points.
//   usage: shim_check <icp_yaml> <filters_yaml> <reference.bin> <reading.bin>   (float32 N x 4 clouds)
//          shim_check --compile-only
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include "../../integration/lsgpu_icp_shim.hpp"
#include "laser_slam_amd/ros_msgs.hpp"

using namespace laser_slam_amd;

static DataPoints readCloud(const char* path) {
  std::ifstream f(path, std::ios::binary);
  f.seekg(0, std::ios::end);
  const size_t bytes = (size_t)f.tellg();
  f.seekg(0);
  DataPoints d;
  d.features.resize(bytes / 4);
  f.read(reinterpret_cast<char*>(d.features.data()), (std::streamsize)bytes);
  return d;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "--compile-only") { std::printf("shim_check: compiled\n"); return 0; }
  if (argc < 5) return 2;
  int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)
  const DataPoints ref = readCloud(argv[3]), rd = readCloud(argv[4]);
  TransformationParameters T_init = identityTransformation();
  T_init[12] = 0.6f; T_init[13] = 0.1f;                      // a rough guess of the 0.8 m step
  LsgpuICP<LsgpuMirrorPM> shim;
  ICP mirror;
  { std::ifstream y(argv[1]); shim.loadFromYaml(y); }
  { std::ifstream y(argv[1]); mirror.loadFromYaml(y); }
  shim.setSeed(3); mirror.setSeed(3);
  const TransformationParameters Ta = shim.compute(rd, ref, T_init), Tb = mirror.compute(rd, ref, T_init);
  CHECK(std::memcmp(Ta.data(), Tb.data(), sizeof(float) * 16) == 0);
  CHECK(shim.lastStats().iterations == mirror.lastStats().iterations && shim.lastStats().iterations >= 2);
  CHECK(std::fabs(Ta[12] - 0.8f) < 0.05f);
  bool threw = false;
  try { DataPoints empty; shim.compute(empty, ref, T_init); } catch (const ConvergenceError&) { threw = true; }
  CHECK(threw);
  // a guess that is not rigid is refused at step 5 of ICP::compute, i.e. AFTER both filters have consumed their rand()
  // draws (upstream throws TransformationError out of RigidTransformation::compute there): a caller that catches the
  // exception and goes on must see the draws it would have seen after a successful call with the same clouds
  {
    ICP a, b;
    { std::ifstream y(argv[1]); a.loadFromYaml(y); }
    { std::ifstream y(argv[1]); b.loadFromYaml(y); }
    TransformationParameters T_bad = T_init;
    T_bad[0] = 1.1f;                                           // det 1.1
    a.setSeed(9);
    threw = false;
    try { a.compute(rd, ref, T_bad); } catch (const TransformationError&) { threw = true; }
    CHECK(threw);
    a.setSeed(-1);                                             // continue the stream
    const TransformationParameters Tc = a.compute(rd, ref, T_init);
    b.setSeed(9);
    (void)b.compute(rd, ref, T_init);                          // the same filters on the same clouds: the same draws consumed
    b.setSeed(-1);
    const TransformationParameters Td = b.compute(rd, ref, T_init);
    CHECK(std::memcmp(Tc.data(), Td.data(), sizeof(float) * 16) == 0);
    b.setSeed(9);
    const TransformationParameters Te = b.compute(rd, ref, T_init);   // (and the draws matter: from the stream's start the result differs)
    CHECK(std::memcmp(Tc.data(), Te.data(), sizeof(float) * 16) != 0);
  }
  threw = false;
  try { std::istringstream bad("matcher:\n  NullMatcher\n"); shim.loadFromYaml(bad); } catch (const std::runtime_error&) { threw = true; }
  CHECK(threw);
  // the filter twin: same survivors as the mirror's DataPointsFilters, descriptors thinned with them
  std::ifstream fy(argv[2]), fy2(argv[2]);
  LsgpuDataPointsFilters<LsgpuMirrorPM> sf(fy);
  DataPointsFilters mf(fy2);
  sf.setSeed(11); mf.setSeed(11);
  DataPoints a = rd, b = rd;
  a.normals.resize((size_t)a.getNbPoints() * 3);
  for (int64_t i = 0; i < a.getNbPoints(); ++i) { a.normals[3 * i] = (float)i; a.normals[3 * i + 1] = a.features[4 * i]; a.normals[3 * i + 2] = 7.f; }
  sf.apply(a);
  mf.apply(b);
  CHECK(a.getNbPoints() == b.getNbPoints() && a.getNbPoints() > 0 && a.getNbPoints() < rd.getNbPoints());
  CHECK(a.normals.size() == (size_t)a.getNbPoints() * 3);
  bool same = true, desc_ok = true;
  for (int64_t i = 0; i < a.getNbPoints() && i < b.getNbPoints(); ++i) {
    same = same && std::memcmp(&a.features[4 * i], &b.features[4 * i], 12) == 0 && a.features[4 * i + 3] == 1.f;
    const int64_t src = (int64_t)a.normals[3 * i];
    desc_ok = desc_ok && a.normals[3 * i + 1] == a.features[4 * i] && rd.features[4 * src] == a.features[4 * i];
  }
  CHECK(same);
  CHECK(desc_ok);
  {  // the mirror's own DataPointsFilters carries a cloud's normals through as well: same survivors, their own descriptors
    std::ifstream fy3(argv[2]);
    DataPointsFilters mf2(fy3);
    mf2.setSeed(11);
    DataPoints c = rd;
    c.normals.resize((size_t)c.getNbPoints() * 3);
    for (int64_t i = 0; i < c.getNbPoints(); ++i) { c.normals[3 * i] = (float)i; c.normals[3 * i + 1] = c.features[4 * i]; c.normals[3 * i + 2] = 7.f; }
    mf2.apply(c);
    CHECK(c.getNbPoints() == b.getNbPoints() && c.normals.size() == (size_t)c.getNbPoints() * 3);
    bool ok = true;
    for (int64_t i = 0; i < c.getNbPoints() && i < b.getNbPoints(); ++i)
      ok = ok && std::memcmp(&c.features[4 * i], &b.features[4 * i], 16) == 0 && c.normals[3 * i + 1] == c.features[4 * i] &&
           c.normals[3 * i + 2] == 7.f && rd.features[4 * (int64_t)c.normals[3 * i]] == c.features[4 * i];
    CHECK(ok);
  }
  {  // ROS message surface: a Velodyne-style PointCloud2 (x,y,z,intensity,ring: 22-byte records) -> DataPoints -> back
    PointCloud2 msg;
    const int64_t n = rd.getNbPoints();
    msg.width = (uint32_t)n; msg.point_step = 22; msg.row_step = 22 * msg.width; msg.is_dense = false;
    const char* names[3] = {"x", "y", "z"};
    for (int k = 0; k < 3; ++k) { PointField f; f.name = names[k]; f.offset = 4u * (uint32_t)k; msg.fields.push_back(f); }
    { PointField f; f.name = "intensity"; f.offset = 16; msg.fields.push_back(f); }
    msg.data.assign((size_t)n * 22, 0xAB);
    for (int64_t i = 0; i < n; ++i) std::memcpy(&msg.data[(size_t)i * 22], &rd.features[4 * i], 12);
    const float nan = std::nanf("");
    std::memcpy(&msg.data[22 * 5 + 4], &nan, 4);                       // record 5 loses its y
    DataPoints conv = rosMsgToPointMatcherCloud(mirror, msg);
    CHECK(conv.getNbPoints() == n - 1);
    CHECK(std::memcmp(&conv.features[0], &rd.features[0], 16 * 5) == 0 && std::memcmp(&conv.features[4 * 5], &rd.features[4 * 6], 16) == 0);
    PointCloud2 out = pointMatcherCloudToRosMsg(mirror, conv, "world", 42);
    CHECK(out.width == (uint32_t)(n - 1) && out.point_step == 16 && out.data.size() == (size_t)(n - 1) * 16);
    CHECK(std::memcmp(out.data.data(), conv.features.data(), out.data.size()) == 0);
    DataPoints again = rosMsgToPointMatcherCloud(mirror, out);
    CHECK(again.features == conv.features);
  }
  std::printf(fails ? "shim_check: %d FAILED\n" : "shim_check: ok\n", fails);
  return fails ? 1 : 0;
}
