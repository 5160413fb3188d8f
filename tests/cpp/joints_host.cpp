// tests/cpp/joints_host.cpp — host build of the joint arithmetic the device runs (taichi_mpm_amd/csrc/k_joints.h), as a small
// shared library for tests/test_joints_cpu.py: the same header, compiled by g++, checked against the reference's compiled
// joints (tests/golden/ref_joints.npz) without a GPU.
#include <cstring>
#include <vector>

#include "../../taichi_mpm_amd/csrc/k_joints.h"

extern "C" {
int joints_sizeof_body() { return (int)sizeof(mpm::JointBody); }
int joints_sizeof_config() { return (int)sizeof(mpm::JointConfig); }
// setup / bodies: nb JointBody records each (body 0 = background), inertia: nb x 9 (body frame).  Sets the joints up from `cfg`
// on the poses of `setup` (the initialize() methods run when the joint is added), then runs MPM::articulate(dt) on `bodies` in
// place.  Returns 0, or 1 + the index of the joint whose set-up failed.
int joints_articulate(int nb, const mpm::JointBody *setup, mpm::JointBody *bodies, const float *inertia, int nj,
                      const mpm::JointConfig *cfg, float dt, int iterations) {
  std::vector<mpm::JointDev> joints(nj);
  for (int i = 0; i < nj; i++) {
    const mpm::JointConfig &c = cfg[i];
    if (c.obj0 < 0 || c.obj0 >= nb || c.obj1 < 0 || c.obj1 >= nb) return 1 + i;
    if (mpm::joint_init(joints[i], c, setup[c.obj0], setup[c.obj1], inertia + 9 * c.obj0, inertia + 9 * c.obj1)) return 1 + i;
  }
  std::vector<mpm::JointPre> pre(nj);
  mpm::articulate(bodies, nb, joints.data(), pre.data(), nj, dt, iterations);
  return 0;
}
}
