"""Joints between rigid bodies (the reference's src/articulation.cpp): the arithmetic the device runs
(taichi_mpm_amd/csrc/k_joints.h) compiled for the host by g++ and checked against output of the reference's own joints
(tests/golden/ref_joints.npz: src/articulation.cpp compiled in place into oracle/_ref/libmpm_ref.so, bodies = the shim's
RigidBody) — no GPU needed; tests/test_gpu_cpic.py runs the same scenes through the device."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "joints_host.cpp")
HDR = os.path.join(ROOT, "taichi_mpm_amd", "csrc", "k_joints.h")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libjoints_host.so")
JOINT_TYPES = {"rotation": 0, "frozen": 1, "distance": 2, "axial_rotation": 3, "motor": 4, "stepper": 5}


class JointBody(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("vel", C.c_float * 3), ("omega", C.c_float * 3), ("R", C.c_float * 9),
                ("inv_mass", C.c_float), ("inv_I", C.c_float * 9), ("Iw", C.c_float * 9)]


class JointConfig(C.Structure):
    _fields_ = [("type", C.c_int), ("obj0", C.c_int), ("obj1", C.c_int), ("has_offset1", C.c_int), ("has_target", C.c_int),
                ("offset0", C.c_float * 3), ("offset1", C.c_float * 3), ("target_distance", C.c_float), ("penalty", C.c_float),
                ("axis", C.c_float * 3), ("axis_length", C.c_float), ("power", C.c_float), ("angular_velocity", C.c_float)]


def host_lib():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(OUT):
        # -ffp-contract=off: the device build contracts a*b+c on its own terms; the comparison below has room for either
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-shared", "-fPIC", SRC, "-o", OUT])
    L = C.CDLL(OUT)
    assert L.joints_sizeof_body() == C.sizeof(JointBody) and L.joints_sizeof_config() == C.sizeof(JointConfig)
    L.joints_articulate.argtypes = [C.c_int, C.POINTER(JointBody), C.POINTER(JointBody), C.POINTER(C.c_float), C.c_int,
                                    C.POINTER(JointConfig), C.c_float, C.c_int]
    return L


def quat_to_matrix(q):
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float32)


def bodies_of(rows):
    """fixture rows (position 3, quaternion 4, velocity 3, angular velocity 3, mass, inv_mass, inertia 9, inv_inertia 9)"""
    arr = (JointBody * len(rows))()
    for b, r in zip(arr, rows):
        b.pos[:] = r[0:3]
        b.vel[:] = r[7:10]
        b.omega[:] = r[10:13]
        b.R[:] = quat_to_matrix(r[3:7]).ravel()
        b.inv_mass = r[14]
        b.inv_I[:] = r[24:33]
    return arr


def joint_config(j):
    c = JointConfig()
    c.type = JOINT_TYPES[j["type"]]
    c.obj0, c.obj1 = int(j["obj0"]), int(j.get("obj1", 0))
    c.offset0[:] = j.get("offset0", (0, 0, 0))
    c.has_offset1 = int("offset1" in j)
    c.offset1[:] = j.get("offset1", (0, 0, 0))
    c.has_target = int("target_distance" in j)
    c.target_distance = j.get("target_distance", 0.0)
    c.penalty = j.get("penalty", -1.0)
    c.axis[:] = j.get("axis", (0, 0, 0))
    c.axis_length = j.get("axis_length", -1.0)
    c.power = j.get("power", 0.0)
    c.angular_velocity = j.get("angular_velocity", 0.0)
    return c


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_joints.npz"))
    return g, json.loads(str(g["cases"]))


def run_host(L, setup, run, joints, dt):
    sb, rb = bodies_of(setup), bodies_of(run)
    inertia = np.ascontiguousarray(setup[:, 15:24], np.float32)
    cfg = (JointConfig * len(joints))(*[joint_config(j) for j in joints])
    rc = L.joints_articulate(len(setup), sb, rb, inertia.ctypes.data_as(C.POINTER(C.c_float)), len(joints), cfg, dt, 100)
    assert rc == 0, rc
    return np.array([list(b.vel) + list(b.omega) for b in rb], np.float32)


def test_joint_arithmetic_matches_the_reference(golden):
    g, cases = golden
    L = host_lib()
    dt = float(g["dt"])
    for name, joints in cases.items():
        got = run_host(L, g[name + "/setup"], g[name + "/drifted"], joints, dt)
        want = g[name + "/articulated"][:, 7:13]
        before = g[name + "/drifted"][:, 7:13]
        assert np.abs(want - before).max() > 1e-3, name  # the joint did something in this scene
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 2e-5 * scale, (name, np.abs(got - want).max())


def test_the_background_body_needs_an_anchor_for_a_distance_joint(golden):
    g, cases = golden
    L = host_lib()
    setup = g["distance/setup"]
    sb, rb = bodies_of(setup), bodies_of(setup)
    inertia = np.ascontiguousarray(setup[:, 15:24], np.float32)
    cfg = (JointConfig * 1)(joint_config(dict(type="distance", obj0=1, obj1=0)))
    assert L.joints_articulate(len(setup), sb, rb, inertia.ctypes.data_as(C.POINTER(C.c_float)), 1, cfg, 1e-3, 100) == 1
