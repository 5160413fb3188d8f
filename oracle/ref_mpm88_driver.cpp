// oracle/ref_mpm88_driver.cpp — TEST INFRASTRUCTURE (never shipped, never on the product path).
// The reference's 2D dense-grid demo, /root/reference/mls-mpm88.cpp, compiled from where it lies: this file includes it
// (its main() renamed and never called) and moves particle data in and out of ITS globals (`particles`, `grid`,
// `plastic`) around ITS advance() (mls-mpm88.cpp:16-69) through a small C ABI (binding: oracle/refmpm.py, ref88_*).
// It contains no arithmetic of the path.  "taichi.h" resolves to oracle/taichi_shim/taichi.h.
#include <cstdint>
#include <cstring>

#define main ref_mpm88_unused_main
#include "mls-mpm88.cpp"
#undef main

extern "C" {
int32_t ref88_grid_cells() { return n; }
double ref88_dt() { return dt; }
void ref88_reset(int32_t plastic_flag) {
  particles.clear();
  plastic = plastic_flag != 0;
  std::memset(grid, 0, sizeof(grid));
}
// x, v: count*2; F, C: count*4 ROW-major (F[2*r+c]); Jp: count
void ref88_add(int64_t count, const float *x, const float *v, const float *F, const float *C, const float *Jp) {
  for (int64_t i = 0; i < count; i++) {
    Particle p(Vec(x[2 * i], x[2 * i + 1]), 0, v ? Vec(v[2 * i], v[2 * i + 1]) : Vec(0));
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 2; c++) {
        if (F) p.F[c][r] = F[4 * i + 2 * r + c];  // (the shim's matrices are column-major like taichi's: m[col][row])
        if (C) p.C[c][r] = C[4 * i + 2 * r + c];
      }
    if (Jp) p.Jp = Jp[i];
    particles.push_back(p);
  }
}
int64_t ref88_num_particles() { return (int64_t)particles.size(); }
void ref88_advance(int32_t steps) {
  for (int32_t s = 0; s < steps; s++) advance(dt);
}
void ref88_get(float *x, float *v, float *F, float *C, float *Jp) {
  for (size_t i = 0; i < particles.size(); i++) {
    const Particle &p = particles[i];
    for (int k = 0; k < 2; k++) { x[2 * i + k] = p.x[k]; v[2 * i + k] = p.v[k]; }
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 2; c++) { F[4 * i + 2 * r + c] = p.F[c][r]; C[4 * i + 2 * r + c] = p.C[c][r]; }
    Jp[i] = p.Jp;
  }
}
// (v.x, v.y, m) per node after the last advance(), node (i, j) at [(i * (n + 1) + j) * 3]
void ref88_get_grid(float *out) {
  for (int i = 0; i <= n; i++)
    for (int j = 0; j <= n; j++)
      for (int k = 0; k < 3; k++) out[((size_t)i * (n + 1) + j) * 3 + k] = grid[i][j][k];
}
}
