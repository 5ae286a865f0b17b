#!/usr/bin/env python3
"""Extract the rigid-body table of the Dual-UR5 from the reference's MJCF scene (build container only: reads
/root/reference, which never travels) and write it as DATA to irl_control_amd/models/dual_ur5.json.

    python tools/parse_mjcf.py                      # scenes/dual_ur5.xml -> models/dual_ur5.json
    python tools/parse_mjcf.py /root/reference/irl_control/scenes/ur5.xml ur5 arm=EE      # a second tree: models/ur5.json

What is taken (scenes/dual_ur5.xml:51-265 in ir-lab/irl_control @ 2024_10_08): the body tree in MuJoCo's numbering
(document order, depth first), each body's frame relative to its parent (pos + quat; `euler` attributes are converted
with MuJoCo's default intrinsic x-y-z sequence, scene compilers say angle="radian"), its hinge joint (axis and anchor in
the body frame), its explicit <inertial> (mass, frame, principal moments), the F/T sites, the actuator -> joint list
(:267-287).  No <default> block exists in these scenes, so armature = damping = 0; gravity is MuJoCo's default.

Bodies without <inertial> get their inertia from their geoms in MuJoCo (inertiafromgeom = "auto").  Only
base_link_ur5right / base_link_ur5left are in that case (one STL mesh each, meshes/ur5/link0.stl, rigidly attached to the
stand; scenes/dual_ur5.xml:63,164): their mass, centre and inertia are integrated from the mesh HERE, the way MuJoCo
documents it for mesh geoms -- uniform density 1000 kg/m^3, the solid as a union of triangular pyramids with the apex at
the area-weighted centroid of the surface; volumes taken positive (the legacy algorithm of the MuJoCo 2.x the reference's
mujoco_py wraps: exact for convex meshes) -- and stored as the body's inertial (mass, ipos, iquat, principal moments).
The signed-volume variant (compiler exactmeshinertia="true" / MuJoCo 3) is stored next to it under "mesh_inertia_exact"
for comparison: 0.314 kg instead of 0.343 kg per link, which moves the yaw inertia M[0][0] of the stand joint by 1e-4
relative.  Only numbers leave this script: the STL stays in /root/reference.  Massless frames (ur_stand_dummy, ur_EE_*,
EE_*) carry no geoms and are massless in MuJoCo too.
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/irl_control/scenes/dual_ur5.xml"
NAME = sys.argv[2] if len(sys.argv) > 2 else "dual_ur5"
# device name -> end-effector body: the `EE:` entries of irl_control/robot_configs/default_xyz*.yaml for the Dual-UR5
EE = dict(a.split("=") for a in sys.argv[3:]) if len(sys.argv) > 3 else \
    {"base": "ur_stand_dummy", "ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irl_control_amd", "models", NAME + ".json")


def vec(s, n, default):
    if s is None:
        return list(default)
    v = [float(x) for x in s.split()]
    assert len(v) == n, (s, n)
    return v


def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2]


def frame_quat(el):
    if el.get("quat") is not None:
        q = np.array(vec(el.get("quat"), 4, (1, 0, 0, 0)))
        return (q / np.linalg.norm(q)).tolist()
    if el.get("euler") is not None:                     # intrinsic x, y, z (MuJoCo eulerseq "xyz"), radians
        e = vec(el.get("euler"), 3, (0, 0, 0))
        q = [1.0, 0.0, 0.0, 0.0]
        for ax, ang in enumerate(e):
            r = [np.cos(ang / 2), 0.0, 0.0, 0.0]
            r[1 + ax] = np.sin(ang / 2)
            q = qmul(q, r)
        return q
    return [1.0, 0.0, 0.0, 0.0]


def load_stl(path):
    """Binary STL -> triangles [n, 3 vertices, 3]."""
    import struct
    with open(path, "rb") as f:
        raw = f.read()
    n = struct.unpack("<I", raw[80:84])[0]
    assert len(raw) == 84 + 50 * n, "binary STL expected"
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    return rec["v"].astype(np.float64)


def mesh_inertia(tri, density=1000.0, exact=False):
    """(mass, centre of mass, inertia tensor about it) of the solid bounded by the triangles: pyramids over every face with
    the apex at the area-weighted centroid of the surface; exact=False takes every pyramid volume positive (MuJoCo's legacy
    mesh inertia), exact=True signed (exactmeshinertia)."""
    v0, v1, v2 = tri[:, 0], tri[:, 1], tri[:, 2]
    area = 0.5 * np.linalg.norm(np.cross(v1 - v0, v2 - v0), axis=1)
    cen = (area[:, None] * (v0 + v1 + v2) / 3.0).sum(0) / area.sum()
    a, b, c = v0 - cen, v1 - cen, v2 - cen
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)) / 6.0
    if not exact:
        vol = np.abs(vol)
    V = vol.sum()
    d = (vol[:, None] * (a + b + c) / 4.0).sum(0) / V            # centre of mass relative to cen
    P = np.zeros((3, 3))                                          # second moments about cen: tetrahedra with one vertex there
    for x, w in zip(np.stack([a, b, c], axis=1), vol):
        sm = x.sum(0)
        P += w / 20.0 * (x.T @ x + np.outer(sm, sm))
    P -= V * np.outer(d, d)
    return density * V, cen + d, density * (np.trace(P) * np.eye(3) - P)


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mat_to_quat(R):
    """Unit quaternion (w x y z) of a proper rotation matrix."""
    t = np.trace(R)
    if t > 0:
        s = 2 * np.sqrt(1 + t)
        q = [s / 4, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2 * np.sqrt(1 + R[i, i] - R[j, j] - R[k, k])
        q = [0.0, 0.0, 0.0, 0.0]
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = s / 4
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return (q / np.linalg.norm(q)).tolist()


def geoms_inertial(el, meshes, exact):
    """The <inertial> MuJoCo infers for a body from its mesh geoms: (mass, ipos, iquat, principal moments) in the body frame."""
    m_tot, mc, parts = 0.0, np.zeros(3), []
    for g in el.findall("geom"):
        assert g.get("type") == "mesh", "only mesh geoms carry inferred inertia in this scene"
        m, com, I = mesh_inertia(load_stl(meshes[g.get("mesh")]), exact=exact)
        R = quat_to_mat(frame_quat(g))
        c = np.array(vec(g.get("pos"), 3, (0, 0, 0))) + R @ com
        parts.append((m, c, R @ I @ R.T))
        m_tot += m
        mc += m * c
    c0 = mc / m_tot
    I0 = sum(I + m * (np.dot(c - c0, c - c0) * np.eye(3) - np.outer(c - c0, c - c0)) for m, c, I in parts)
    w, V = np.linalg.eigh(I0)
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return float(m_tot), c0.tolist(), mat_to_quat(V), w.tolist()


def main():
    root = ET.parse(SRC).getroot()
    bodies, sites = [], []
    comp = root.find("compiler")
    meshdir = comp.get("meshdir", "") if comp is not None else ""
    meshes = {m.get("name", os.path.splitext(os.path.basename(m.get("file")))[0]):
              os.path.normpath(os.path.join(os.path.dirname(SRC), meshdir, m.get("file"))) for m in root.find("asset").findall("mesh")}

    def walk(el, parent, moved=False):
        idx = len(bodies)
        b = dict(name=el.get("name"), parent=parent, pos=vec(el.get("pos"), 3, (0, 0, 0)), quat=frame_quat(el),
                 joint=None, mass=0.0, ipos=[0.0, 0.0, 0.0], iquat=[1.0, 0.0, 0.0, 0.0], inertia=[0.0, 0.0, 0.0])
        joints = el.findall("joint")
        assert len(joints) <= 1, "one hinge per body in this scene"
        if joints:
            j = joints[0]
            assert j.get("type", "hinge") == "hinge"
            ax = np.array(vec(j.get("axis"), 3, (0, 0, 1)))
            b["joint"] = dict(name=j.get("name"), axis=(ax / np.linalg.norm(ax)).tolist(), pos=vec(j.get("pos"), 3, (0, 0, 0)),
                              armature=float(j.get("armature", 0.0)), range=vec(j.get("range"), 2, (0, 0)))
        ine = el.find("inertial")
        if ine is not None:
            assert ine.get("fullinertia") is None
            b["mass"] = float(ine.get("mass"))
            b["ipos"] = vec(ine.get("pos"), 3, (0, 0, 0))
            b["iquat"] = frame_quat(ine)
            b["inertia"] = vec(ine.get("diaginertia"), 3, (0, 0, 0))
        elif el.findall("geom") and not (moved or joints):
            b["inertia_from"] = "fixed to the world (no hinge above it): its inertia enters nothing and is not integrated"
        elif el.findall("geom"):                          # inertiafromgeom = "auto": integrate the mesh geoms
            b["mass"], b["ipos"], b["iquat"], b["inertia"] = geoms_inertial(el, meshes, exact=False)
            me, pe, qe, ie = geoms_inertial(el, meshes, exact=True)
            b["inertia_from"] = "mesh geoms, legacy (positive-volume) pyramids, density 1000"
            b["mesh_inertia_exact"] = dict(mass=me, ipos=pe, iquat=qe, inertia=ie)
        bodies.append(b)
        for s in el.findall("site"):
            sites.append(dict(name=s.get("name"), body=idx, pos=vec(s.get("pos"), 3, (0, 0, 0)), quat=frame_quat(s)))
        for child in el.findall("body"):
            walk(child, idx, moved or bool(joints))

    for top in root.find("worldbody").findall("body"):
        walk(top, -1)
    joint_names = [b["joint"]["name"] for b in bodies if b["joint"]]
    acts = [a.get("joint") for a in root.find("actuator")]
    out = dict(source=f"ir-lab/irl_control @ 2024_10_08, irl_control/scenes/{os.path.basename(SRC)}, via tools/parse_mjcf.py",
               gravity=[0.0, 0.0, -9.81], bodies=bodies, sites=sites, joint_names=joint_names, actuator_joints=acts,
               ee_bodies=EE,
               notes="parent -1 = world; body frames relative to the parent; euler -> quat with MuJoCo's intrinsic xyz; bodies "
                     "without <inertial> (inertia_from set) carry the inertia MuJoCo infers from their mesh geoms")
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{len(bodies)} bodies, {len(joint_names)} joints, {len(sites)} sites, {len(acts)} actuators -> {OUT}")
    print("inertia integrated from mesh geoms:", [(b["name"], round(b["mass"], 4)) for b in bodies if b.get("mesh_inertia_exact")])


if __name__ == "__main__":
    main()
