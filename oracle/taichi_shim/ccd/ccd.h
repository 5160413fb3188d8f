// oracle/taichi_shim (TEST INFRASTRUCTURE): the few libccd declarations src/rigid_body_solver.h names.  Rigid-rigid
// collision detection (libccd's MPR on the bodies' meshes) is OUTSIDE this build: ccdMPRPenetration reports "no
// penetration" (-1), so MPM::rigidify (src/mpm_rigid_body.cpp:289-326) finds no collisions.  The MPM <-> rigid coupling
// (CPIC: src/rigid_transfer.cpp and the rigid branches of src/transfer.cpp) does not depend on it.
#pragma once
#include <cmath>
typedef float ccd_real_t;
struct ccd_vec3_t { ccd_real_t v[3]; };
typedef void (*ccd_support_fn)(const void *obj, const ccd_vec3_t *dir, ccd_vec3_t *vec);
typedef void (*ccd_center_fn)(const void *obj, ccd_vec3_t *center);
struct ccd_t {
  ccd_support_fn support1 = nullptr, support2 = nullptr;
  ccd_center_fn center1 = nullptr, center2 = nullptr;
  unsigned long max_iterations = 0;
  ccd_real_t epa_tolerance = 0, mpr_tolerance = 0, dist_tolerance = 0;
};
#define CCD_INIT(ccd) do { *(ccd) = ccd_t(); } while (0)
inline void ccdVec3Set(ccd_vec3_t *v, ccd_real_t x, ccd_real_t y, ccd_real_t z) { v->v[0] = x; v->v[1] = y; v->v[2] = z; }
inline void ccdVec3Copy(ccd_vec3_t *d, const ccd_vec3_t *s) { *d = *s; }
inline void ccdVec3Normalize(ccd_vec3_t *d) {
  const ccd_real_t n = std::sqrt(d->v[0] * d->v[0] + d->v[1] * d->v[1] + d->v[2] * d->v[2]);
  if (n > 0) { d->v[0] /= n; d->v[1] /= n; d->v[2] /= n; }
}
inline int ccdMPRPenetration(const void *, const void *, const ccd_t *, ccd_real_t *, ccd_vec3_t *, ccd_vec3_t *) { return -1; }
