// taichi_mpm_amd/csrc/k_joints.h — joints between rigid bodies (the reference's Articulation classes, src/articulation.cpp)
// Part of libmpmhip (see mpmhip.hip for the substep overview; k_rigid.h for the bodies).
//
// MPM::articulate (src/mpm.h:278-319) runs once per substep, after the sort and before rasterize_rigid_boundary
// (src/mpm.cpp:466-471): apply(dt) of every joint, `articulation_iterations` (100) Gauss-Seidel sweeps of project()
// over all joints in the order they were added, then penalize(dt) of every joint.  Every step reads the velocities the
// previous one wrote, so the whole thing is ONE sequential chain over a handful of bodies: a single lane walks it on
// the device (the body records live there — they collect the particles' impulses every substep — and a host round trip
// per substep would stall the launch queue).  The arithmetic below is plain C++ on the body records, shared with the
// host build of tests/cpp/test_joints.cpp, which checks it against the reference's compiled joints without a GPU.
//
//   rotation        both bodies get the angular velocity (I0 + I1)^-1 (L0 + L1), world-frame inertias   (:23-46)
//   frozen          obj0: angular velocity x, y and velocity z set to zero (3D)                            (:69-79)
//   distance        two anchor points, one per body, kept at a target distance: project() removes the relative velocity
//                   along their connection by a pair of impulses, penalize() pushes by penalty (target - distance) dt
//                                                                                                         (:83-166)
//   axial_rotation  two distance joints with target 0 at +- axis_length along the axis: a hinge            (:168-232)
//   motor           axial_rotation + a torque of `power` about the axis (in obj1's frame) every substep     (:235-286)
//   stepper         axial_rotation + the relative angular velocity about the axis driven to `angular_velocity` (:289-352)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MPM_HD __host__ __device__ __forceinline__
#else
#define MPM_HD inline
#endif

namespace mpm {

enum JointType { JOINT_ROTATION = 0, JOINT_FROZEN = 1, JOINT_DISTANCE = 2, JOINT_AXIAL = 3, JOINT_MOTOR = 4, JOINT_STEPPER = 5 };
constexpr int MAX_JOINTS = 32;

// what a joint reads and writes of a body (a view of RigidBodyDev: same field meanings)
struct JointBody {
  float pos[3], vel[3], omega[3];
  float R[9];      // body -> world, row-major
  float inv_mass;
  float inv_I[9];  // body frame, row-major
  float Iw[9];     // world-frame inverse inertia R inv_I R^T: filled by articulate()
};
struct JointDev {
  int type, obj0, obj1, n_dist;  // n_dist: distance constraints in use (distance: 1, axial / motor / stepper: 2)
  float off[2][2][3];            // off[d][i]: anchor of constraint d on body i, in that body's frame (DistanceArticulation::offsets)
  float target[2], penalty;
  float axis[3];                 // hinge axis in obj1's frame (AxialRotationArticulation::axis)
  float power, angular_velocity;
  float I[2][9];                 // body-frame inertia of obj0 / obj1 (rotation joint), row-major
};

MPM_HD void j_cross(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
MPM_HD float j_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MPM_HD void j_mat_vec(const float M[9], const float v[3], float o[3]) {
  for (int r = 0; r < 3; r++) o[r] = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2];
}
MPM_HD void j_matT_vec(const float M[9], const float v[3], float o[3]) {
  for (int c = 0; c < 3; c++) o[c] = M[c] * v[0] + M[3 + c] * v[1] + M[6 + c] * v[2];
}
// R M R^T: a body-frame tensor in the world frame (get_transformed_inertia / get_transformed_inversed_inertia)
MPM_HD void j_to_world(const float R[9], const float M[9], float o[9]) {
  float t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = R[3 * r] * M[c] + R[3 * r + 1] * M[3 + c] + R[3 * r + 2] * M[6 + c];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) o[3 * r + c] = t[3 * r] * R[3 * c] + t[3 * r + 1] * R[3 * c + 1] + t[3 * r + 2] * R[3 * c + 2];
}
MPM_HD void j_inverse3(const float m[9], float o[9]) {
  const float c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
  const float id = 1.0f / (m[0] * c0 + m[1] * c1 + m[2] * c2);
  o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
// RigidBody::apply_torque / apply_impulse / get_velocity_at / get_impulse_contribution.  The bodies do not move while
// MPM::articulate runs (only velocities change), so everything that depends on the poses alone is computed ONCE per call:
// a body's world-frame inverse inertia (JointBody::Iw) and, per joint, the anchors, their connection, the impulse
// denominators, the world-frame axis and the combined inertias (JointPre).  The sweeps then only move velocities:
// 264 -> 108 us per hinge and articulate() of 100 sweeps on one lane of an MI355X (profiles/joint_cost.py), same arithmetic.
MPM_HD void j_apply_torque(JointBody &B, const float t[3]) {
  float d[3];
  j_mat_vec(B.Iw, t, d);
  for (int k = 0; k < 3; k++) B.omega[k] += d[k];
}
MPM_HD void j_apply_impulse(JointBody &B, const float imp[3], const float r[3]) {  // r = point of attack - centre of mass
  for (int k = 0; k < 3; k++) B.vel[k] += imp[k] * B.inv_mass;
  float t[3];
  j_cross(r, imp, t);
  j_apply_torque(B, t);
}
MPM_HD void j_velocity_at(const JointBody &B, const float r[3], float o[3]) {  // r as above
  float c[3];
  j_cross(B.omega, r, c);
  for (int k = 0; k < 3; k++) o[k] = B.vel[k] + c[k];
}
MPM_HD float j_impulse_contribution(const JointBody &B, const float r[3], const float n[3]) {
  float rn[3], t[3], u[3];
  j_cross(r, n, rn);
  j_mat_vec(B.Iw, rn, t);
  j_cross(t, r, u);
  return B.inv_mass + j_dot(u, n);
}
MPM_HD void j_anchor(const JointBody &B, const float off[3], float p[3]) {  // transform(get_centroid_to_world(), offset)
  j_mat_vec(B.R, off, p);
  for (int k = 0; k < 3; k++) p[k] += B.pos[k];
}

// what one articulate() call computes once per joint
struct DistPre { int active; float n[3], r0[3], r1[3], dist, den; };  // active = 0: the anchors coincide (distance < 1e-10)
struct JointPre {
  DistPre d[2];
  float aw[3];             // motor: the axis in the world frame; stepper: the same, normalised
  float M0[9], M1[9];      // rotation: the bodies' world-frame inertias
  float Si[9];             // rotation: (M0 + M1)^-1;  stepper: (Iw0 + Iw1)^-1
};
MPM_HD void joint_prepare(const JointDev &J, const JointBody &A, const JointBody &B, JointPre &P) {
  for (int d = 0; d < J.n_dist; d++) {  // DistanceArticulation::project / penalize up to the velocities (:124-135, :143-155)
    DistPre &D = P.d[d];
    float p0[3], p1[3];
    j_anchor(A, J.off[d][0], p0);
    j_anchor(B, J.off[d][1], p1);
    for (int k = 0; k < 3; k++) D.n[k] = p0[k] - p1[k];
    D.dist = sqrtf(j_dot(D.n, D.n));
    D.active = !(D.dist < 1e-10f);
    if (!D.active) continue;
    const float il = 1.0f / D.dist;
    for (int k = 0; k < 3; k++) { D.n[k] *= il; D.r0[k] = p0[k] - A.pos[k]; D.r1[k] = p1[k] - B.pos[k]; }
    D.den = j_impulse_contribution(A, D.r0, D.n) + j_impulse_contribution(B, D.r1, D.n);
  }
  if (J.type == JOINT_MOTOR || J.type == JOINT_STEPPER) j_mat_vec(B.R, J.axis, P.aw);  // transform(obj[1]->get_centroid_to_world(), axis, 0)
  if (J.type == JOINT_STEPPER) {
    const float il = 1.0f / sqrtf(j_dot(P.aw, P.aw));
    for (int k = 0; k < 3; k++) P.aw[k] *= il;
    float S[9];
    for (int k = 0; k < 9; k++) S[k] = A.Iw[k] + B.Iw[k];
    j_inverse3(S, P.Si);
  }
  if (J.type == JOINT_ROTATION) {
    float S[9];
    j_to_world(A.R, J.I[0], P.M0);
    j_to_world(B.R, J.I[1], P.M1);
    for (int k = 0; k < 9; k++) S[k] = P.M0[k] + P.M1[k];
    j_inverse3(S, P.Si);
  }
}

MPM_HD void j_distance_project(const DistPre &D, JointBody &A, JointBody &B) {
  if (!D.active) return;
  float va[3], vb[3];
  j_velocity_at(A, D.r0, va);
  j_velocity_at(B, D.r1, vb);
  const float v01[3] = {va[0] - vb[0], va[1] - vb[1], va[2] - vb[2]};
  const float j = j_dot(D.n, v01) / D.den;
  const float ia[3] = {-j * D.n[0], -j * D.n[1], -j * D.n[2]}, ib[3] = {j * D.n[0], j * D.n[1], j * D.n[2]};
  j_apply_impulse(A, ia, D.r0);
  j_apply_impulse(B, ib, D.r1);
}
MPM_HD void j_distance_penalize(const JointDev &J, int d, const DistPre &D, JointBody &A, JointBody &B, float dt) {
  if (!D.active) return;
  const float j = -dt * J.penalty * (J.target[d] - D.dist);
  const float ia[3] = {-j * D.n[0], -j * D.n[1], -j * D.n[2]}, ib[3] = {j * D.n[0], j * D.n[1], j * D.n[2]};
  j_apply_impulse(A, ia, D.r0);
  j_apply_impulse(B, ib, D.r1);
}

MPM_HD void joint_apply(const JointDev &J, const JointPre &P, JointBody &A, JointBody &B, float dt) {
  if (J.type != JOINT_MOTOR) return;  // (the stepper forwards to AxialRotationArticulation::apply, which does nothing)
  const float t[3] = {P.aw[0] * J.power * dt, P.aw[1] * J.power * dt, P.aw[2] * J.power * dt};
  const float nt[3] = {-t[0], -t[1], -t[2]};
  j_apply_torque(A, t);
  j_apply_torque(B, nt);
}
MPM_HD void joint_project(const JointDev &J, const JointPre &P, JointBody &A, JointBody &B) {
  if (J.type == JOINT_ROTATION) {
    float L0[3], L1[3], w[3];
    j_mat_vec(P.M0, A.omega, L0);
    j_mat_vec(P.M1, B.omega, L1);
    const float L[3] = {L0[0] + L1[0], L0[1] + L1[1], L0[2] + L1[2]};
    j_mat_vec(P.Si, L, w);
    for (int k = 0; k < 3; k++) A.omega[k] = B.omega[k] = w[k];
    return;
  }
  if (J.type == JOINT_FROZEN) {
    A.omega[0] = 0.0f; A.omega[1] = 0.0f; A.vel[2] = 0.0f;
    return;
  }
  for (int d = 0; d < J.n_dist; d++) j_distance_project(P.d[d], A, B);
  if (J.type == JOINT_STEPPER) {
    const float rel[3] = {A.omega[0] - B.omega[0], A.omega[1] - B.omega[1], A.omega[2] - B.omega[2]};
    const float corr = J.angular_velocity - j_dot(rel, P.aw);
    const float ac[3] = {P.aw[0] * corr, P.aw[1] * corr, P.aw[2] * corr};
    float t[3];
    j_mat_vec(P.Si, ac, t);
    const float nt[3] = {-t[0], -t[1], -t[2]};
    j_apply_torque(A, t);
    j_apply_torque(B, nt);
  }
}
MPM_HD void joint_penalize(const JointDev &J, const JointPre &P, JointBody &A, JointBody &B, float dt) {
  if (J.type < JOINT_DISTANCE) return;
  for (int d = 0; d < J.n_dist; d++) j_distance_penalize(J, d, P.d[d], A, B, dt);
}

// MPM::articulate (src/mpm.h:278-319) on the bodies b[0 .. nb): b[0] is the background body; pre: scratch, one per joint
MPM_HD void articulate(JointBody *b, int nb, const JointDev *joints, JointPre *pre, int nj, float dt, int iterations) {
  for (int i = 0; i < nb; i++) j_to_world(b[i].R, b[i].inv_I, b[i].Iw);
  for (int i = 0; i < nj; i++) joint_prepare(joints[i], b[joints[i].obj0], b[joints[i].obj1], pre[i]);
  for (int i = 0; i < nj; i++) joint_apply(joints[i], pre[i], b[joints[i].obj0], b[joints[i].obj1], dt);
  for (int it = 0; it < iterations; it++)
    for (int i = 0; i < nj; i++) joint_project(joints[i], pre[i], b[joints[i].obj0], b[joints[i].obj1]);
  for (int i = 0; i < nj; i++) joint_penalize(joints[i], pre[i], b[joints[i].obj0], b[joints[i].obj1], dt);
}

// ---- joint set-up (the initialize() methods), on the host with the bodies' current poses
struct JointConfig {  // mirror of mpmhip_joint_config (include/mpmhip.h)
  int type, obj0, obj1, has_offset1, has_target;
  float offset0[3], offset1[3], target_distance, penalty, axis[3], axis_length, power, angular_velocity;
};
// DistanceArticulation::initialize (:99-122): world-frame offsets from the bodies' centres -> body frames
inline void joint_init_distance(JointDev &J, int d, const JointBody &A, const JointBody &B, const float off0[3], const float off1[3],
                                bool has_target, float target, float penalty) {
  j_matT_vec(A.R, off0, J.off[d][0]);  // transform(inverse(centroid_to_world), offset, 0)
  j_matT_vec(B.R, off1, J.off[d][1]);
  float p0[3], p1[3];
  j_anchor(A, J.off[d][0], p0);
  j_anchor(B, J.off[d][1], p1);
  const float n[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
  J.target[d] = has_target ? target : sqrtf(j_dot(n, n));
  J.penalty = penalty;
}
// returns 0, or a message
inline const char *joint_init(JointDev &J, const JointConfig &cfg, const JointBody &A, const JointBody &B, const float I0[9],
                              const float I1[9]) {
  J = JointDev{};
  J.type = cfg.type; J.obj0 = cfg.obj0; J.obj1 = cfg.obj1;
  for (int k = 0; k < 9; k++) { J.I[0][k] = I0[k]; J.I[1][k] = I1[k]; }
  const float penalty = cfg.penalty >= 0.0f ? cfg.penalty : 1e3f;  // (negative: the reference's default)
  if (cfg.type == JOINT_ROTATION || cfg.type == JOINT_FROZEN) return nullptr;
  if (cfg.type == JOINT_DISTANCE) {
    if (cfg.obj1 == 0 && !cfg.has_offset1)
      return "When linking to the background rigid body, a distance articulation must have a non-default offset1";  // :101-105
    const float zero[3] = {0, 0, 0};
    J.n_dist = 1;
    joint_init_distance(J, 0, A, B, cfg.offset0, cfg.has_offset1 ? cfg.offset1 : zero, cfg.has_target != 0, cfg.target_distance, penalty);
    return nullptr;
  }
  if (cfg.type == JOINT_AXIAL || cfg.type == JOINT_MOTOR || cfg.type == JOINT_STEPPER) {  // AxialRotationArticulation::initialize (:182-216)
    const float al = sqrtf(j_dot(cfg.axis, cfg.axis));
    if (!(al > 0.0f)) return "axial_rotation / motor / stepper articulations need an 'axis'";
    const float an[3] = {cfg.axis[0] / al, cfg.axis[1] / al, cfg.axis[2] / al};
    j_matT_vec(B.R, an, J.axis);
    const float len = cfg.axis_length >= 0.0f ? cfg.axis_length : 0.1f;
    float offset[3];
    for (int k = 0; k < 3; k++) offset[k] = A.pos[k] + cfg.offset0[k] - B.pos[k];
    J.n_dist = 2;
    for (int d = 0; d < 2; d++) {
      const float s = d == 0 ? len : -len;
      const float o0[3] = {cfg.offset0[0] + an[0] * s, cfg.offset0[1] + an[1] * s, cfg.offset0[2] + an[2] * s};
      const float o1[3] = {offset[0] + an[0] * s, offset[1] + an[1] * s, offset[2] + an[2] * s};
      joint_init_distance(J, d, A, B, o0, o1, true, 0.0f, penalty);
    }
    J.power = cfg.power;
    J.angular_velocity = cfg.angular_velocity;
    return nullptr;
  }
  return "unknown articulation type";
}

}  // namespace mpm
