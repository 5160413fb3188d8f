// oracle/partio_ref_driver.cpp — TEST INFRASTRUCTURE (only tests/ and the golden-fixture script run it).
//
// Drives the REFERENCE's own vendored Partio (external/partio, compiled in place by `make -C oracle ref_partio`,
// output oracle/_ref/partio_write) through the attribute sequence of MPM<dim>::write_partio
// (src/visualize.cpp:17-100): position VECTOR3, type INT1, index INT1, limit INT3, v VECTOR3 and, with
// verbose_bgeo, m VECTOR1, boundary_normal VECTOR3, debug VECTOR3, states INT1, boundary_distance FLOAT1,
// near_boundary INT1, apic_frobenius_norm FLOAT1; particles in ascending id (:39-43).  The bytes it writes are the
// reference's bytes for that particle state: they pin the .bgeo encoder of libmpmhip (tests/golden/bgeo_*).
//
// usage: partio_write <in.raw> <out.bgeo>
// in.raw: int32 n, int32 verbose, then per particle (native little-endian, in FILE order, any id order):
//   float pos[3], float v[3], int32 id, int32 is_rigid, int32 limit[3]
//   and if verbose: float mass, float boundary_normal[3], float debug[3], int32 states,
//                   float boundary_distance_in_cells, int32 near_boundary, float apic_b[9] (row-major)
#include <Partio.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

struct Row {
  float pos[3], v[3];
  int32_t id, is_rigid, limit[3];
  float mass, bn[3], debug[3];
  int32_t states;
  float bdist;
  int32_t near_boundary;
  float b[9];
};

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t n = 0, verbose = 0;
  if (std::fread(&n, 4, 1, f) != 1 || std::fread(&verbose, 4, 1, f) != 1) return 4;
  std::vector<Row> rows(n);
  for (auto &r : rows) {
    r = Row{};
    size_t ok = std::fread(r.pos, 4, 3, f) + std::fread(r.v, 4, 3, f) + std::fread(&r.id, 4, 1, f) +
                std::fread(&r.is_rigid, 4, 1, f) + std::fread(r.limit, 4, 3, f);
    if (ok != 11) return 5;
    if (verbose) {
      ok = std::fread(&r.mass, 4, 1, f) + std::fread(r.bn, 4, 3, f) + std::fread(r.debug, 4, 3, f) +
           std::fread(&r.states, 4, 1, f) + std::fread(&r.bdist, 4, 1, f) + std::fread(&r.near_boundary, 4, 1, f) +
           std::fread(r.b, 4, 9, f);
      if (ok != 19) return 6;
    }
  }
  std::fclose(f);
  std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.id < b.id; });  // visualize.cpp:39-43

  Partio::ParticlesDataMutable *parts = Partio::create();
  Partio::ParticleAttribute posH, vH, mH, typeH, normH, statH, boundH, distH, debugH, indexH, limitH, apicH;
  posH = parts->addAttribute("position", Partio::VECTOR, 3);
  typeH = parts->addAttribute("type", Partio::INT, 1);
  indexH = parts->addAttribute("index", Partio::INT, 1);
  limitH = parts->addAttribute("limit", Partio::INT, 3);
  vH = parts->addAttribute("v", Partio::VECTOR, 3);
  if (verbose) {
    mH = parts->addAttribute("m", Partio::VECTOR, 1);
    normH = parts->addAttribute("boundary_normal", Partio::VECTOR, 3);
    debugH = parts->addAttribute("debug", Partio::VECTOR, 3);
    statH = parts->addAttribute("states", Partio::INT, 1);
    distH = parts->addAttribute("boundary_distance", Partio::FLOAT, 1);
    boundH = parts->addAttribute("near_boundary", Partio::INT, 1);
    apicH = parts->addAttribute("apic_frobenius_norm", Partio::FLOAT, 1);
  }
  for (const Row &r : rows) {
    const int idx = parts->addParticle();
    if (verbose) {
      parts->dataWrite<float>(mH, idx)[0] = r.mass;
      for (int k = 0; k < 3; k++) parts->dataWrite<float>(normH, idx)[k] = r.bn[k];
      for (int k = 0; k < 3; k++) parts->dataWrite<float>(debugH, idx)[k] = r.debug[k];
      parts->dataWrite<int>(statH, idx)[0] = r.states;
      parts->dataWrite<int>(boundH, idx)[0] = r.near_boundary;
      parts->dataWrite<float>(distH, idx)[0] = r.bdist;
      float s = 0;  // || 0.5 (B - B^T) ||_F  (visualize.cpp:70-71)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          const float a = 0.5f * (r.b[i * 3 + j] - r.b[j * 3 + i]);
          s += a * a;
        }
      parts->dataWrite<float>(apicH, idx)[0] = std::sqrt(s);
    }
    for (int k = 0; k < 3; k++) parts->dataWrite<float>(vH, idx)[k] = r.v[k];
    parts->dataWrite<int>(typeH, idx)[0] = r.is_rigid;
    parts->dataWrite<int>(indexH, idx)[0] = r.id;
    for (int k = 0; k < 3; k++) parts->dataWrite<int>(limitH, idx)[k] = r.limit[k];
    for (int k = 0; k < 3; k++) parts->dataWrite<float>(posH, idx)[k] = r.pos[k];
  }
  Partio::write(argv[2], *parts);
  parts->release();
  return 0;
}
