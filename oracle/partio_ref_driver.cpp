// oracle/partio_ref_driver.cpp — TEST INFRASTRUCTURE (only tests/ and the golden-fixture script run it).
//
// Feeds a particle state through the REFERENCE's own vendored Partio (external/partio, compiled in place by
// `make -C oracle ref_partio` -> oracle/_ref/partio_write) with the attribute set, attribute order and particle order
// that MPM<dim>::write_partio uses (src/visualize.cpp:17-100): position, type, index, limit[3], v and — verbose_bgeo —
// m, boundary_normal[3], debug[3], states, boundary_distance, near_boundary, apic_frobenius_norm; ascending id.
// The bytes Partio then writes are the reference's bytes for that state: they pin libmpmhip's .bgeo encoder
// (tests/golden/bgeo_*).
//
// usage: partio_write <in.raw> <out.bgeo>
// in.raw: int32 n, int32 verbose, then one record per particle (native little-endian, any id order):
//   float pos[3], float v[3], int32 id, int32 is_rigid, int32 limit[3]
//   verbose only: float mass, float boundary_normal[3], float debug[3], int32 states,
//                 float boundary_distance_in_cells, int32 near_boundary, float apic_b[9] (row-major)
#include <Partio.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

// one output attribute: where its words sit inside the input record (in 4-byte words; -1 = computed)
struct Column {
  const char *name;
  Partio::ParticleAttributeType type;
  int count;
  int word;  // offset into the record
};
// record words: 0 pos, 3 v, 6 id, 7 is_rigid, 8 limit | 11 mass, 12 normal, 15 debug, 18 states, 19 distance, 20 near, 21 apic_b
constexpr int PLAIN_WORDS = 11, VERBOSE_WORDS = 30;
const Column PLAIN[] = {{"position", Partio::VECTOR, 3, 0}, {"type", Partio::INT, 1, 7}, {"index", Partio::INT, 1, 6},
                        {"limit", Partio::INT, 3, 8},        {"v", Partio::VECTOR, 3, 3}};
const Column VERBOSE[] = {{"m", Partio::VECTOR, 1, 11},       {"boundary_normal", Partio::VECTOR, 3, 12},
                          {"debug", Partio::VECTOR, 3, 15},   {"states", Partio::INT, 1, 18},
                          {"boundary_distance", Partio::FLOAT, 1, 19}, {"near_boundary", Partio::INT, 1, 20},
                          {"apic_frobenius_norm", Partio::FLOAT, 1, -1}};

float skew_norm(const uint32_t *rec) {  // || 0.5 (B - B^T) ||_F of the apic_b at words 21..29
  float b[9];
  std::memcpy(b, rec + 21, sizeof b);
  float acc = 0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      const float h = 0.5f * (b[3 * r + c] - b[3 * c + r]);
      acc += h * h;
    }
  return std::sqrt(acc);
}

}  // namespace

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  FILE *in = std::fopen(argv[1], "rb");
  if (!in) return 3;
  int32_t head[2];
  if (std::fread(head, 4, 2, in) != 2) return 4;
  const int n = head[0];
  const bool verbose = head[1] != 0;
  const int words = verbose ? VERBOSE_WORDS : PLAIN_WORDS;
  std::vector<uint32_t> recs((size_t)n * words);
  if (n && std::fread(recs.data(), 4, recs.size(), in) != recs.size()) return 5;
  std::fclose(in);

  std::vector<int> by_id(n);
  std::iota(by_id.begin(), by_id.end(), 0);
  auto id_of = [&](int p) { int32_t v; std::memcpy(&v, &recs[(size_t)p * words + 6], 4); return v; };
  std::sort(by_id.begin(), by_id.end(), [&](int a, int b) { return id_of(a) < id_of(b); });

  std::vector<Column> cols(PLAIN, PLAIN + 5);
  if (verbose) cols.insert(cols.end(), VERBOSE, VERBOSE + 7);
  Partio::ParticlesDataMutable *out = Partio::create();
  std::vector<Partio::ParticleAttribute> handles;
  for (const Column &c : cols) handles.push_back(out->addAttribute(c.name, c.type, c.count));
  for (int p : by_id) {
    const uint32_t *rec = &recs[(size_t)p * words];
    const int row = out->addParticle();
    for (size_t a = 0; a < cols.size(); a++) {
      // FLOAT / VECTOR / INT are all 4-byte words in Partio: copy the bits
      uint32_t *dst = reinterpret_cast<uint32_t *>(out->dataWrite<float>(handles[a], row));
      if (cols[a].word < 0) {
        const float s = skew_norm(rec);
        std::memcpy(dst, &s, 4);
      } else {
        std::memcpy(dst, rec + cols[a].word, 4 * (size_t)cols[a].count);
      }
    }
  }
  Partio::write(argv[2], *out);
  out->release();
  return 0;
}
