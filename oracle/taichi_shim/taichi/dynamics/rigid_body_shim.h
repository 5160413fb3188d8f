// oracle/taichi_shim (TEST INFRASTRUCTURE) — rigid bodies for the reference's CPIC coupling.
//
// The reference's rigid-coupling sources (src/rigid_transfer.cpp, src/mpm_rigid_body.cpp, src/boundary_particle.h and the
// rigid branches of src/transfer.cpp) are compiled where they lie; what they need from the un-vendored legacy taichi
// core is a RigidBody<dim> (taichi/dynamics/rigid_body.h), a triangle mesh (taichi/geometry/mesh.h), rotations
// (taichi/math/angular.h) and three Eigen types.  None of those exist in the container, so THIS FILE IS OURS: a plain
// textbook rigid body.  Everything the coupling computes with it — colours, distances, the particle-side projection,
// the impulses handed to a body — is the reference's own code; what a body DOES with an impulse and how a scripted
// body turns its script into a velocity is decided here (stated per function below) and mirrored by the device code
// (taichi_mpm_amd/csrc/k_rigid.h), so parity tests compare like with like.  Included from taichi/common/util.h.
#pragma once

namespace Eigen {  // the three Eigen types src/mpm_rigid_body.cpp:108-114 and src/rigid_body_solver.h touch
template <typename T, int R, int C>
struct Matrix {
  T d[R * C] = {};
  static Matrix UnitX() { Matrix m; m.d[0] = 1; return m; }
  static Matrix UnitY() { Matrix m; m.d[1] = 1; return m; }
  static Matrix UnitZ() { Matrix m; m.d[2] = 1; return m; }
};
template <typename T>
struct Quaternion {
  T qw = 1, qx = 0, qy = 0, qz = 0;
  Quaternion() {}
  Quaternion(T w, T x, T y, T z) : qw(w), qx(x), qy(y), qz(z) {}
  T w() const { return qw; } T x() const { return qx; } T y() const { return qy; } T z() const { return qz; }
  Quaternion operator*(const Quaternion &o) const {
    return Quaternion(qw * o.qw - qx * o.qx - qy * o.qy - qz * o.qz, qw * o.qx + qx * o.qw + qy * o.qz - qz * o.qy,
                      qw * o.qy - qx * o.qz + qy * o.qw + qz * o.qx, qw * o.qz + qx * o.qy - qy * o.qx + qz * o.qw);
  }
  Quaternion conjugate() const { return Quaternion(qw, -qx, -qy, -qz); }
  void normalize() {
    const T n = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= n; qx /= n; qy /= n; qz /= n;
  }
};
template <typename T>
struct AngleAxis {
  T angle;
  Matrix<T, 3, 1> axis;
  AngleAxis(T a, const Matrix<T, 3, 1> &ax) : angle(a), axis(ax) {}
  operator Quaternion<T>() const {
    const T s = std::sin(angle / 2);
    return Quaternion<T>(std::cos(angle / 2), s * axis.d[0], s * axis.d[1], s * axis.d[2]);
  }
  Quaternion<T> operator*(const AngleAxis &o) const { return Quaternion<T>(*this) * Quaternion<T>(o); }
};
template <typename T>
Quaternion<T> operator*(const Quaternion<T> &q, const AngleAxis<T> &a) { return q * Quaternion<T>(a); }
}  // namespace Eigen

namespace taichi {

inline Vector3 cross(const Vector3 &a, const Vector3 &b) {
  return Vector3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline real cross(const Vector2 &a, const Vector2 &b) { return a.x * b.y - a.y * b.x; }
template <int dim> inline VectorND<dim, real> normalize(const VectorND<dim, real> &a) { return a / a.length(); }
template <int dim> inline VectorND<dim, real> lerp(real t, const VectorND<dim, real> &a, const VectorND<dim, real> &b) { return a * (1.0f - t) + b * t; }
inline real inversed(real x) { return 1.0f / x; }
using math::radians;
template <int dim> inline VectorND<dim, real> radians(const VectorND<dim, real> &v) { return v * (real)(M_PI / 180.0); }

// homogeneous transform: w = 1 a point, w = 0 a direction (src/articulation.cpp:111,189,261)
template <int dim>
inline VectorND<dim, real> transform(const MatrixND<dim + 1, real> &m, const VectorND<dim, real> &v, real w = 1.0f) {
  return VectorND<dim, real>(m * VectorND<dim + 1, real>(v, w));
}
template <int dim> inline MatrixND<dim, real> inverse(const MatrixND<dim, real> &m) { return inversed(m); }

// ---------------------------------------------------------------------------------------------- mesh elements
// Element<3> = triangle, Element<2> = segment (taichi/geometry/mesh.h)
template <int dim>
struct ElementShim {
  using Vector = VectorND<dim, real>;
  using MatrixP = MatrixND<dim + 1, real>;
  Vector v[dim];
  Vector get_normal() const {
    if constexpr (dim == 3) return normalized(cross(v[1] - v[0], v[2] - v[0]));
    else { const Vector d = v[1] - v[0]; return normalized(Vector(d.y, -d.x)); }
  }
  ElementShim get_transformed(const MatrixP &m) const {
    ElementShim e;
    for (int k = 0; k < dim; k++) e.v[k] = transform(m, v[k]);
    return e;
  }
};
// maps (world position - v[0]) to (edge coordinates ..., signed distance along the unit normal): the inverse of the
// matrix whose columns are the edge vectors and the normal (src/rigid_transfer.cpp:29-38 reads coord[0..dim-2] as
// barycentric-style coordinates and coord[dim-1] as the distance)
template <int dim>
inline MatrixND<dim, real> world_to_element(const ElementShim<dim> &e) {
  MatrixND<dim, real> m;
  for (int k = 0; k + 1 < dim; k++) m[k] = e.v[k + 1] - e.v[0];
  m[dim - 1] = e.get_normal();
  return inversed(m);
}
template <int dim>
struct ElementMeshShim {
  std::vector<ElementShim<dim>> elements;
  // legacy core: loads `mesh_fn`.  Here the driver hands the elements over: shim_mesh_ptr -> float[n][dim][dim], shim_mesh_n
  void initialize(const Config &config) {
    const float *p = config.get_ptr<float>("shim_mesh_ptr");
    const int n = config.get<int>("shim_mesh_n");
    elements.resize(n);
    for (int e = 0; e < n; e++)
      for (int k = 0; k < dim; k++)
        for (int c = 0; c < dim; c++) elements[e].v[k][c] = p[(e * dim + k) * dim + c];
    if (config.get("reverse_vertices", false))
      for (auto &e : elements) std::swap(e.v[0], e.v[1]);
  }
};

// ---------------------------------------------------------------------------------------------- rotations
template <int dim> struct AngularVelocity;
template <> struct AngularVelocity<2> {
  using ValueType = real;
  real value = 0;
  AngularVelocity() {}
  AngularVelocity(real v) : value(v) {}
  Vector2 cross(const Vector2 &r) const { return Vector2(-r.y, r.x) * value; }
  AngularVelocity &operator+=(const AngularVelocity &o) { value += o.value; return *this; }
};
template <> struct AngularVelocity<3> {
  using ValueType = Vector3;
  Vector3 value;
  AngularVelocity() {}
  AngularVelocity(const Vector3 &v) : value(v) {}
  Vector3 cross(const Vector3 &r) const { return taichi::cross(value, r); }
  AngularVelocity &operator+=(const AngularVelocity &o) { value += o.value; return *this; }
};
template <int dim> struct Rotation;
template <> struct Rotation<2> {
  real value = 0;
  Rotation() {}
  explicit Rotation(real a) : value(a) {}
  Matrix2 get_rotation_matrix() const { const real c = std::cos(value), s = std::sin(value); return Matrix2(Vector2(c, s), Vector2(-s, c)); }
  Vector2 rotate(const Vector2 &v) const { return get_rotation_matrix() * v; }
  void apply_angular_velocity(const AngularVelocity<2> &w, real dt) { value += w.value * dt; }
};
template <> struct Rotation<3> {
  Eigen::Quaternion<real> value;
  Matrix3 get_rotation_matrix() const {
    const real w = value.qw, x = value.qx, y = value.qy, z = value.qz;
    return Matrix3(Vector3(1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)),   // column 0
                   Vector3(2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)),   // column 1
                   Vector3(2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)));  // column 2
  }
  Vector3 rotate(const Vector3 &v) const { return get_rotation_matrix() * v; }
  // q <- exp(omega dt / 2) q with omega in the WORLD frame, renormalised
  void apply_angular_velocity(const AngularVelocity<3> &w, real dt) {
    const real len = w.value.length();
    if (len * dt < 1e-12f) return;
    const real h = 0.5f * len * dt, s = std::sin(h) / len;
    value = Eigen::Quaternion<real>(std::cos(h), s * w.value.x, s * w.value.y, s * w.value.z) * value;
    value.normalize();
  }
};
inline Eigen::Quaternion<real> shim_quat_from_euler_deg(const Vector3 &deg) {  // X * Y * Z as src/mpm_rigid_body.cpp:109-113
  const Vector3 e = radians(deg);
  return Eigen::AngleAxis<real>(e[0], Eigen::Matrix<real, 3, 1>::UnitX()) * Eigen::AngleAxis<real>(e[1], Eigen::Matrix<real, 3, 1>::UnitY()) *
         Eigen::AngleAxis<real>(e[2], Eigen::Matrix<real, 3, 1>::UnitZ());
}

// ---------------------------------------------------------------------------------------------- the body
template <int dim>
struct RigidBody {
  using Vector = VectorND<dim, real>;
  using Matrix = MatrixND<dim, real>;
  using MatrixP = MatrixND<dim + 1, real>;
  using ElementType = ElementShim<dim>;
  using MeshType = ElementMeshShim<dim>;
  using InertiaType = std::conditional_t<dim == 2, real, Matrix>;
  using PositionFunctionType = std::function<Vector(real)>;
  using RotationFunctionType = std::function<std::conditional_t<dim == 2, real, Vector>(real)>;

  int id = 0;  // index into MPM::rigids (the driver assigns it: the legacy core used a process-wide counter)
  Spinlock mutex;
  real mass = 0, inv_mass = 0;
  InertiaType inertia = InertiaType(0.0f), inv_inertia = InertiaType(0.0f);  // body frame, about the centre of mass
  Vector position, velocity, tmp_velocity;
  Rotation<dim> rotation;
  AngularVelocity<dim> angular_velocity, tmp_angular_velocity;
  Vector rotation_axis;
  real frictions[2] = {0, 0};
  real restitution = 0, linear_damping = 0, angular_damping = 0;
  bool codimensional = false;
  Vector3 color;
  int pos_func_id = -1, rot_func_id = -1;
  PositionFunctionType pos_func;
  RotationFunctionType rot_func;
  std::unique_ptr<MeshType> mesh;

  void set_as_background() { id = 0; mass = 1; inv_mass = 0; inertia = InertiaType(1.0f); inv_inertia = InertiaType(0.0f); }
  // a scripted translation (rotation) answers impulses like an infinitely heavy body; the finite mass stays readable
  // (advect_rigid_bodies multiplies gravity by it, src/mpm_rigid_body.cpp:265)
  void set_infinity_mass() { inv_mass = 0; }
  void set_infinity_inertia() { inv_inertia = InertiaType(0.0f); }
  real get_mass() const { return mass; }
  InertiaType get_inertia() const { return inertia; }

  // mass, centre of mass and inertia about it from the (scaled, not yet recentred) mesh; returns the centre of mass.
  // codimensional: a shell of surface density `density` (mass = density * area); otherwise the solid enclosed by the
  // outward-oriented mesh (signed tetrahedra against the origin).  3D; 2D bodies: segments as a shell / polygon area.
  Vector initialize_mass_and_inertia(real density) {
    double M = 0, com[3] = {0, 0, 0}, S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // S = second moments int x x^T dm about the origin
    auto add_point_moments = [&](double w, const double a[3], const double b[3]) {  // w * sym(a b^T)
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[r][c] += w * 0.5 * (a[r] * b[c] + b[r] * a[c]);
    };
    for (const auto &e : mesh->elements) {
      double v[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int k = 0; k < dim; k++) for (int c = 0; c < dim; c++) v[k][c] = e.v[k][c];
      if constexpr (dim == 3) {
        if (codimensional) {
          const double a[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]}, b[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
          const double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
          const double m = 0.5 * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) * density;
          M += m;
          for (int c = 0; c < 3; c++) com[c] += m * (v[0][c] + v[1][c] + v[2][c]) / 3.0;
          // int over a triangle of x x^T dm = m/12 * (sum_i v_i v_i^T + (sum_i v_i)(sum_j v_j)^T)
          double s[3] = {v[0][0] + v[1][0] + v[2][0], v[0][1] + v[1][1] + v[2][1], v[0][2] + v[1][2] + v[2][2]};
          for (int i = 0; i < 3; i++) add_point_moments(m / 12.0, v[i], v[i]);
          add_point_moments(m / 12.0, s, s);
        } else {
          const double det = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                             v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
          const double m = det / 6.0 * density;  // signed tetrahedron (0, v0, v1, v2)
          M += m;
          for (int c = 0; c < 3; c++) com[c] += m * (v[0][c] + v[1][c] + v[2][c]) / 4.0;
          // int over the tetrahedron of x x^T dm = m/20 * (sum_i v_i v_i^T + (sum_i v_i)(sum_j v_j)^T), 4th vertex = 0
          double s[3] = {v[0][0] + v[1][0] + v[2][0], v[0][1] + v[1][1] + v[2][1], v[0][2] + v[1][2] + v[2][2]};
          for (int i = 0; i < 3; i++) add_point_moments(m / 20.0, v[i], v[i]);
          add_point_moments(m / 20.0, s, s);
        }
      } else {
        const double d[2] = {v[1][0] - v[0][0], v[1][1] - v[0][1]};
        const double m = std::sqrt(d[0] * d[0] + d[1] * d[1]) * density;
        M += m;
        for (int c = 0; c < 2; c++) com[c] += m * 0.5 * (v[0][c] + v[1][c]);
        double s[3] = {v[0][0] + v[1][0], v[0][1] + v[1][1], 0};
        for (int i = 0; i < 2; i++) add_point_moments(m / 6.0, v[i], v[i]);
        add_point_moments(m / 6.0, s, s);
      }
    }
    if (!(std::abs(M) > 0)) shim_fail("RigidBody: mesh without mass", __FILE__, __LINE__);
    for (int c = 0; c < 3; c++) com[c] /= M;
    const double flip = M < 0 ? -1.0 : 1.0;  // a solid with inward-facing triangles: every signed term flips together
    M *= flip;
    // second moments about the centre of mass, then I = tr(Sc) 1 - Sc
    double Sc[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Sc[r][c] = flip * S[r][c] - M * com[r] * com[c];
    const double tot = Sc[0][0] + Sc[1][1] + Sc[2][2];
    mass = (real)M;
    inv_mass = (real)(1.0 / M);
    if constexpr (dim == 3) {
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) inertia[c][r] = (real)((r == c ? tot : 0.0) - Sc[r][c]);
      inv_inertia = inversed(inertia);
    } else {
      inertia = (real)tot;
      inv_inertia = (real)(1.0 / tot);
    }
    Vector out;
    for (int c = 0; c < dim; c++) out[c] = (real)com[c];
    return out;
  }

  MatrixP get_centroid_to_world() const {
    MatrixP m(1.0f);
    const Matrix R = rotation.get_rotation_matrix();
    for (int c = 0; c < dim; c++) for (int r = 0; r < dim; r++) m[c][r] = R[c][r];
    for (int r = 0; r < dim; r++) m[dim][r] = position[r];
    return m;
  }
  MatrixP get_mesh_to_world() const { return get_centroid_to_world(); }  // the mesh is recentred on its centre of mass

  InertiaType get_transformed_inversed_inertia() const {
    if constexpr (dim == 2) return inv_inertia;
    else { const Matrix R = rotation.get_rotation_matrix(); return R * inv_inertia * transposed(R); }
  }
  InertiaType get_transformed_inertia() const {
    if constexpr (dim == 2) return inertia;
    else { const Matrix R = rotation.get_rotation_matrix(); return R * inertia * transposed(R); }
  }
  typename AngularVelocity<dim>::ValueType get_angular_momemtum() const { return get_transformed_inertia() * angular_velocity.value; }
  void apply_torque(const typename AngularVelocity<dim>::ValueType &t) { angular_velocity.value += get_transformed_inversed_inertia() * t; }
  Vector get_velocity_at(const Vector &p) const { return velocity + angular_velocity.cross(p - position); }
  // change of the velocity at `position + r` along n per unit impulse along n applied there
  real get_impulse_contribution(const Vector &r, const Vector &n) const {
    if constexpr (dim == 2) return inv_mass + inv_inertia * sqr(cross(r, n));
    else return inv_mass + dot(cross(get_transformed_inversed_inertia() * cross(r, n), r), n);
  }
  void apply_impulse(const Vector &impulse, const Vector &orig) {
    velocity += impulse * inv_mass;
    if constexpr (dim == 2) angular_velocity.value += inv_inertia * cross(orig - position, impulse);
    else angular_velocity.value += get_transformed_inversed_inertia() * cross(orig - position, impulse);
  }
  // impulses collected during a transfer (from many threads) and applied to the body afterwards
  void reset_tmp_velocity() { tmp_velocity = Vector(0.0f); tmp_angular_velocity = AngularVelocity<dim>(); }
  void apply_tmp_impulse(const Vector &impulse, const Vector &orig) {
    mutex.lock();
    tmp_velocity += impulse * inv_mass;
    if constexpr (dim == 2) tmp_angular_velocity.value += inv_inertia * cross(orig - position, impulse);
    else tmp_angular_velocity.value += get_transformed_inversed_inertia() * cross(orig - position, impulse);
    mutex.unlock();
  }
  void apply_tmp_velocity() { velocity += tmp_velocity; angular_velocity += tmp_angular_velocity; }
  void enforce_angular_velocity_parallel_to(const Vector &axis) {  // world-frame axis
    if constexpr (dim == 3) { const Vector a = normalized(axis); angular_velocity.value = a * dot(a, angular_velocity.value); }
  }
  // one substep from t to t + dt.  Scripted translation: the body sits on its script, position = pos_func(t + dt), and
  // moves with the secant velocity of the step just taken, (pos_func(t + dt) - pos_func(t)) / dt.  Scripted rotation
  // (Euler angles in degrees, X * Y * Z): rotation = q(t + dt), angular velocity = the axis-angle of q(t + dt) q(t)^-1
  // over dt.  Free motion: exponential damping, then explicit Euler for the position and the exact exponential map for
  // the rotation.
  void advance(real t, real dt) {
    if (pos_func) {
      const Vector p0 = pos_func(t), p1 = pos_func(t + dt);
      velocity = (p1 - p0) / dt;
      position = p1;
    } else {
      velocity *= std::exp(-linear_damping * dt);
      position += velocity * dt;
    }
    if (rot_func) {
      if constexpr (dim == 3) {
        const Eigen::Quaternion<real> q0 = shim_quat_from_euler_deg(rot_func(t)), q1 = shim_quat_from_euler_deg(rot_func(t + dt));
        Eigen::Quaternion<real> dq = q1 * q0.conjugate();
        if (dq.qw < 0) dq = Eigen::Quaternion<real>(-dq.qw, -dq.qx, -dq.qy, -dq.qz);
        const real s = std::sqrt(dq.qx * dq.qx + dq.qy * dq.qy + dq.qz * dq.qz);
        const real ang = 2.0f * std::atan2(s, dq.qw);
        angular_velocity.value = s > 1e-12f ? Vector3(dq.qx, dq.qy, dq.qz) * (ang / (s * dt)) : Vector3(0.0f);
        rotation.value = q1;
      } else {
        const real a0 = radians(rot_func(t)), a1 = radians(rot_func(t + dt));
        angular_velocity.value = (a1 - a0) / dt;
        rotation.value = a1;
      }
    } else {
      if constexpr (dim == 3) angular_velocity.value *= std::exp(-angular_damping * dt);
      else angular_velocity.value *= std::exp(-angular_damping * dt);
      rotation.apply_angular_velocity(angular_velocity, dt);
    }
  }
};

}  // namespace taichi
