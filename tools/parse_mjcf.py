#!/usr/bin/env python3
"""Extract the rigid-body table of the Dual-UR5 from the reference's MJCF scene (build container only: reads
/root/reference, which never travels) and write it as DATA to irl_control_amd/models/dual_ur5.json.

    python tools/parse_mjcf.py [/root/reference/irl_control/scenes/dual_ur5.xml]

What is taken (scenes/dual_ur5.xml:51-265 in ir-lab/irl_control @ 2024_10_08): the body tree in MuJoCo's numbering
(document order, depth first), each body's frame relative to its parent (pos + quat; `euler` attributes are converted
with MuJoCo's default intrinsic x-y-z sequence, scene compilers say angle="radian"), its hinge joint (axis and anchor in
the body frame), its explicit <inertial> (mass, frame, principal moments), the F/T sites, the actuator -> joint list
(:267-287).  No <default> block exists in these scenes, so armature = damping = 0; gravity is MuJoCo's default.

What is NOT reproduced: bodies without <inertial> get their inertia from their geoms in MuJoCo (inertiafromgeom =
"auto").  Only base_link_ur5right / base_link_ur5left are in that case (one STL mesh each, rigidly attached to the
stand): their mesh-derived mass is missing here, which changes exactly one number, the yaw inertia M[0][0] of the
stand joint.  Massless frames (ur_stand_dummy, ur_EE_*, EE_*) carry no geoms and are massless in MuJoCo too.
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/irl_control/scenes/dual_ur5.xml"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irl_control_amd", "models", "dual_ur5.json")


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


def main():
    root = ET.parse(SRC).getroot()
    bodies, sites = [], []

    def walk(el, parent):
        idx = len(bodies)
        b = dict(name=el.get("name"), parent=parent, pos=vec(el.get("pos"), 3, (0, 0, 0)), quat=frame_quat(el),
                 joint=None, mass=0.0, ipos=[0.0, 0.0, 0.0], iquat=[1.0, 0.0, 0.0, 0.0], inertia=[0.0, 0.0, 0.0],
                 geom_inertia_missing=False)
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
        elif el.findall("geom"):
            b["geom_inertia_missing"] = True
        bodies.append(b)
        for s in el.findall("site"):
            sites.append(dict(name=s.get("name"), body=idx, pos=vec(s.get("pos"), 3, (0, 0, 0)), quat=frame_quat(s)))
        for child in el.findall("body"):
            walk(child, idx)

    for top in root.find("worldbody").findall("body"):
        walk(top, -1)
    joint_names = [b["joint"]["name"] for b in bodies if b["joint"]]
    acts = [a.get("joint") for a in root.find("actuator")]
    out = dict(source="ir-lab/irl_control @ 2024_10_08, irl_control/scenes/dual_ur5.xml:51-297, via tools/parse_mjcf.py",
               gravity=[0.0, 0.0, -9.81], bodies=bodies, sites=sites, joint_names=joint_names, actuator_joints=acts,
               # device name -> end-effector body (the `EE:` entries of irl_control/robot_configs/default_xyz*.yaml)
               ee_bodies={"base": "ur_stand_dummy", "ur5right": "ur_EE_ur5right", "ur5left": "ur_EE_ur5left"},
               notes="parent -1 = world; body frames relative to the parent; euler -> quat with MuJoCo's intrinsic xyz; "
                     "bodies with geom_inertia_missing get their inertia from mesh geoms in MuJoCo (not reproduced)")
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{len(bodies)} bodies, {len(joint_names)} joints, {len(sites)} sites, {len(acts)} actuators -> {OUT}")
    print("bodies whose mesh-derived inertia is not reproduced:", [b["name"] for b in bodies if b["geom_inertia_missing"]])


if __name__ == "__main__":
    main()
