#!/usr/bin/env python3
"""Extract a PR2 arm serial chain (`right_arm`: torso_lift_link -> r_gripper_tool_frame, or `left_arm`:
torso_lift_link -> l_gripper_tool_frame) from the reference's URDF fixture into a small JSON data file.
Usage: extract_pr2_chain.py [r|l] [out.json]

Source (read-only, only available in the build container):
  /root/reference/trajopt_common/data/arm_around_table.urdf:1479-1866   (joint origins / axes / limits)
  /root/reference/trajopt_common/data/pr2.srdf:12-17                   (groups left_arm / right_arm = chains from torso_lift_link)
  /root/reference/trajopt_common/data/arm_around_table.urdf:121-125, 735-745  (base_footprint -> base_link -> torso_lift_link)
Output: trajopt_amd/data/pr2_{right,left}_arm.json (numbers only — kinematic DATA, no reference code).
`base_footprint_xyz` is the static frame base_footprint expressed in the chain base (torso_lift_joint at its default 0).
Continuous joints have no URDF limits; they get +-2*pi here [NOT IN REFERENCE: tesseract's choice is not pinned].
"""
import json, math, re, sys

URDF = "/root/reference/trajopt_common/data/arm_around_table.urdf"
BASE = "torso_lift_link"

def main(side, out):
    TIP = side + "_gripper_tool_frame"
    s = open(URDF).read()
    J = {}
    for m in re.finditer(r'<joint name="([^"]+)" type="([^"]+)">(.*?)</joint>', s, re.S):
        name, typ, body = m.groups()
        par = re.search(r'<parent link="([^"]+)"', body); ch = re.search(r'<child link="([^"]+)"', body)
        if not par or not ch:
            continue
        o = re.search(r'<origin ([^/]*)/>', body); a = re.search(r'<axis xyz="([^"]+)"', body)
        lim = re.search(r'<limit ([^/]*)/>', body)
        def attr(txt, key, default):
            mm = re.search(key + r'="([^"]+)"', txt or "")
            return [float(v) for v in mm.group(1).split()] if mm else default
        J[ch.group(1)] = dict(name=name, type=typ, parent=par.group(1),
                              xyz=attr(o.group(1) if o else "", "xyz", [0, 0, 0]),
                              rpy=attr(o.group(1) if o else "", "rpy", [0, 0, 0]),
                              axis=[float(v) for v in a.group(1).split()] if a else None,
                              lower=attr(lim.group(1) if lim else "", "lower", [None])[0],
                              upper=attr(lim.group(1) if lim else "", "upper", [None])[0])
    chain = []
    l = TIP
    while l != BASE:
        chain.append(dict(child=l, **J[l])); l = J[l]["parent"]
    chain.reverse()
    # fold fixed joints into the following moving joint's origin (all rpy are zero in this chain)
    joints, pend = [], [0.0, 0.0, 0.0]
    for j in chain:
        assert all(abs(v) < 1e-12 for v in j["rpy"]), "non-zero rpy not handled"
        pend = [pend[k] + j["xyz"][k] for k in range(3)]
        if j["type"] == "fixed":
            continue
        lo, hi = j["lower"], j["upper"]
        if j["type"] == "continuous":
            lo, hi = -2 * math.pi, 2 * math.pi
        joints.append(dict(name=j["name"], type=1 if j["type"] == "prismatic" else 0, origin_xyz=pend,
                           axis=j["axis"], lower=lo, upper=hi, child=j["child"]))
        pend = [0.0, 0.0, 0.0]
    # base_footprint in the chain base frame: walk torso_lift_link up to base_footprint (pure translations, prismatic at 0)
    up, l = [0.0, 0.0, 0.0], BASE
    while l != "base_footprint":
        assert all(abs(v) < 1e-12 for v in J[l]["rpy"]) and J[l]["type"] in ("fixed", "prismatic")
        up = [up[k] + J[l]["xyz"][k] for k in range(3)]; l = J[l]["parent"]
    data = dict(source="arm_around_table.urdf (PR2), group %s_arm" % ("right" if side == "r" else "left"), base_link=BASE,
                tip_link=TIP, joints=joints, tool_xyz=pend, base_footprint_xyz=[0.0 - v for v in up])
    json.dump(data, open(out, "w"), indent=1)
    print("wrote", out, "with", len(joints), "joints; tool offset", pend)

if __name__ == "__main__":
    side = sys.argv[1] if len(sys.argv) > 1 else "r"
    assert side in ("r", "l")
    main(side, sys.argv[2] if len(sys.argv) > 2 else "trajopt_amd/data/pr2_%s_arm.json" % ("right" if side == "r" else "left"))
