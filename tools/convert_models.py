#!/usr/bin/env python3
"""Generate dart_amd/assets/*.json from the reference's model descriptions (run in the build container only;
/root/reference does not exist on the GPU box).

Kinematics (frame tree, joint types/axes/limits, positions, orientations, DH offsets), the thumb contact point
and the potential-intersection matrix are taken verbatim as DATA from
  /root/reference/models/{allegro,HumanHand,Barrett,leftHand}/*.{xml,txt}.
The link meshes of allegro / HumanHand / Barrett are NOT in the reference tree (their XMLs point into sibling
repositories) and leftHand's palm.obj needs the mesh pipeline, so every mesh geom is
replaced by a DECLARED primitive proxy (ellipsoid / cylinder) sized from the kinematic offsets. Proxies only
change what the SDFs look like, not any arithmetic of the hot path.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dart_amd.model import GEOM_CYLINDER, GEOM_SPHERE, read_model_xml  # noqa: E402

REF = "/root/reference/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dart_amd", "assets")


def ell(center, radii):
    return {"type": GEOM_SPHERE, "scale": list(radii), "trans": list(center), "rot": [0, 0, 0]}


def cyl(radius, length):
    # voxelizeFrame's unrotated cylinder: axis +z from 0 to length, translation ignored
    return {"type": GEOM_CYLINDER, "scale": [radius, radius, length], "trans": [0, 0, 0], "rot": [0, 0, 0]}


def allegro_proxy(frame, a, model):
    name = os.path.basename(a["meshFile"])
    if name.startswith("base_link"):
        return [ell((-0.009, 0.0, -0.0475), (0.022, 0.060, 0.052))]
    if "tip" in name:
        return [ell((float(a["tx"]), float(a["ty"]), float(a["tz"])), (0.012, 0.012, 0.012))]
    lengths = {"link_0.0": 0.0164, "link_1.0": 0.054, "link_2.0": 0.0384, "link_3.0": 0.0267,
               "link_12.0_right": 0.0399, "link_13.0": 0.0177, "link_14.0": 0.0514, "link_15.0": 0.0423}
    key = name[:-4]
    return [cyl(0.0105 if not key.startswith("link_1") or key in ("link_1.0",) else 0.012, lengths[key])]


def dh_proxy(palm, radius):
    def fn(frame, a, model):
        if frame == 0:
            return [ell(*palm)]
        link = abs(float(model.dh[frame - 1][0]))
        sign = -1.0 if float(model.dh[frame - 1][0]) > 0 else 1.0
        return [ell((sign * link / 2, 0.0, 0.0), (link / 2 + 0.004, radius, radius))]
    return fn


def left_hand_proxy(frame, a, model):
    return [ell((0.005, 0.05, 0.0), (0.04, 0.05, 0.012))]


def read_matrix(path):
    with open(path) as fh:
        return [[int(v) for v in line.split()] for line in fh if line.strip()]


def read_thumb(path):
    with open(path) as fh:
        v = fh.read().split()
    return {"sdf": int(v[0]), "point": [float(x) for x in v[1:4]]}


def main():
    os.makedirs(OUT, exist_ok=True)
    jobs = [
        ("allegro", "allegro/allegro.xml", allegro_proxy, "allegro"),
        ("human_hand", "HumanHand/human_hand.xml", dh_proxy(((-0.08, 0.0, 0.0), (0.06, 0.014, 0.045)), 0.009), "HumanHand"),
        ("barrett", "Barrett/Barrett.xml", dh_proxy(((0.0, 0.0, -0.04), (0.045, 0.045, 0.04)), 0.011), "Barrett"),
        ("left_hand", "leftHand/leftHand.xml", left_hand_proxy, None),
    ]
    for name, xml, proxy, aux in jobs:
        m = read_model_xml(os.path.join(REF, xml), mesh_proxy=proxy)
        spec = m.to_json()
        spec["name"] = name
        spec["source"] = "models/" + xml
        spec["geometry"] = "declared primitive proxies for missing meshes" if aux else "primitives from the XML; palm mesh replaced by an ellipsoid proxy"
        if aux:
            spec["thumb"] = read_thumb(os.path.join(REF, aux, "thumb.txt"))
            spec["intersection_potential"] = read_matrix(os.path.join(REF, aux, "intersection_potential_matrix.txt"))
        if aux == "HumanHand":
            with open(os.path.join(REF, aux, "keypoints.txt")) as fh:
                spec["keypoints"] = [line.split() for line in fh if line.strip()]
        with open(os.path.join(OUT, name + ".json"), "w") as fh:
            json.dump(spec, fh, indent=1)
        print(name, "frames", m.n_frames, "joints", m.n_joints, "geoms", len(m.geoms))


if __name__ == "__main__":
    main()
