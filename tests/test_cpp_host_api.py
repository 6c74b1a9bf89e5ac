"""C++ host layer (include/tmx_trajopt.hpp: the reference's ProblemConstructionInfo / TermInfo / ConstructProblem /
BasicTrustRegionSQP interface above the C-ABI) driven by tests/cpp/host_api_test.cpp.

The C++ program builds the problems the way the reference's own C++ tests do and checks their assertions itself
(joint_costs_unit.cpp equality_jointPos / inequality_jointPos, numerical_ik_unit.cpp, cart_position_optimization_unit.cpp,
interface_unit.cpp initial trajectory + bitmask, error behaviour).  Here we
additionally require that the C++ lowering and the Python lowering (trajopt_amd/problem.py) are THE SAME problem:
both front ends on the same library must return bit-identical trajectories, statuses and counters.
CPU tier: linked against the kernel sources built for the host (tests/hostemu).  GPU tier: libtrajopt_mi355x.so."""
import os
import subprocess

import numpy as np
import pytest

from trajopt_amd import configs, runtime
from trajopt_amd.problem import pr2_base_footprint, pr2_left_arm, pr2_right_arm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BUILD = os.path.join(HERE, "cpp", "_build")
SRC = os.path.join(HERE, "cpp", "host_api_test.cpp")
HDRS = [os.path.join(ROOT, "include", "tmx_trajopt.hpp"), os.path.join(ROOT, "include", "tmx_trajopt_json.hpp"),
        os.path.join(ROOT, "include", "tmx.h")]
PRODUCT_LIB = os.path.join(ROOT, "trajopt_amd", "_build", "libtrajopt_mi355x.so")
N0, N1 = 3, 3     # seeds per config


def _build(lib_path, tag):
    os.makedirs(BUILD, exist_ok=True)
    # (the executable is bound to ONE library by name and rpath: another build of the sources handed in through TMX_HOSTEMU_LIB gets
    # its own)
    exe = os.path.join(BUILD, "host_api_test_" + tag + "_" + os.path.splitext(os.path.basename(lib_path))[0])
    newest = max(os.path.getmtime(p) for p in [SRC, lib_path] + HDRS)
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC,
                               "-o", exe, "-L" + libdir, "-l:" + libname, "-Wl,-rpath," + libdir, "-fopenmp"])
    return exe


def _fmt(v):
    return " ".join(float(x).hex() for x in np.asarray(v, dtype=np.float64).reshape(-1))


def _robot_block(name, rob, tip):
    out = [f"robot {name} {rob.n_dof} {tip}"]
    for j in range(rob.n_dof):
        out.append(f"{rob.joint_types[j]} {_fmt(rob.origins[j])} {_fmt(rob.axes[j])} {_fmt(rob.lower[j])} {_fmt(rob.upper[j])}")
    out.append(_fmt(rob.base))
    out.append(_fmt(rob.tool))
    out.append(str(len(rob.link_spheres)))
    for link, c, r in rob.link_spheres:
        out.append(f"{link} {_fmt(c)} {_fmt(r)}")
    return out


def time_json_texts():
    """time-parameterised problem descriptions for both JSON readers: (1) the reference's arm_around_table_time.json with its two
    string-valued "use_time" members written as booleans (as a string they make the reference throw, json_marshal.cpp:10-20);
    (2) a problem whose terms use the time column - squared velocity cost with time, TotalTime hinge cost, velocity limits with time"""
    import json
    fx = json.load(open(os.path.join(HERE, "golden", "json", "arm_around_table_time.json")))
    for it in fx["costs"] + fx["constraints"]:
        if isinstance(it.get("use_time"), str):
            it["use_time"] = it["use_time"].lower() == "true"
    terms = {
        "basic_info": {"n_steps": 8, "manip": "right_arm", "fixed_timesteps": [0], "use_time": True, "dt_lower_lim": 0.5, "dt_upper_lim": 5.0},
        "costs": [{"type": "joint_vel", "name": "vel_t", "use_time": True, "params": {"coeffs": [1], "targets": [0, 0, 0, 0, 0, 0, 0]}},
                  {"type": "total_time", "name": "total_time", "use_time": True, "params": {"coeff": 0.5, "limit": 3.0}}],
        "constraints": [{"type": "joint_pos", "name": "joint_pos", "params": {"coeffs": [1, 1, 1, 1, 1, 1, 1],
                                                                                "targets": [0.062, 1.287, 0.1, -1.554, -3.011, -0.268, 2.988],
                                                                                "first_step": 7, "last_step": 7}},
                        {"type": "joint_vel", "name": "vel_lim", "use_time": True,
                         "params": {"targets": [0, 0, 0, 0, 0, 0, 0], "upper_tols": [0.9] * 7, "lower_tols": [-0.9] * 7}}],
        "init_info": {"type": "joint_interpolated", "dt": 1.5, "endpoint": [0.062, 1.287, 0.1, -1.554, -3.011, -0.268, 2.988]},
    }
    return {"arm_around_table_time_bool": json.dumps(fx, indent=1), "time_terms": json.dumps(terms, indent=1)}


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    pci0, s0, g0 = configs.config0()
    pci1, s1, g1 = configs.config1()
    x0 = configs.seeds_for(0, pci0, s0, g0, N0)
    x1 = configs.seeds_for(1, pci1, s1, g1, N1)
    lines = []
    lines += _robot_block("pr2_right_arm", pr2_right_arm(), "r_gripper_tool_frame")
    lines += _robot_block("pr2_right_arm_upright", pci1.robot, "r_gripper_tool_frame")
    lines += _robot_block("pr2_left_arm", pr2_left_arm(), "l_gripper_tool_frame")
    lines.append("frame world " + _fmt(np.hstack([np.eye(3), np.zeros((3, 1))])))
    lines.append("frame base_footprint " + _fmt(pr2_base_footprint()))
    lines.append(f"obstacles {len(pci1.obstacles)}")
    for c, r in pci1.obstacles:
        lines.append(f"{_fmt(c)} {_fmt(r)}")
    for name in ("planning_unit_cfg0", "glass_upright_cfg1", "numerical_ik1", "arm_around_table_time"):
        lines.append(f"file {name} {os.path.join(HERE, 'golden', 'json', name + '.json')}")
    tdir = tmp_path_factory.mktemp("time_json")
    for name, text in time_json_texts().items():
        (tdir / (name + ".json")).write_text(text)
        lines.append(f"file {name} {tdir / (name + '.json')}")
    for name, v in (("cfg0_start", s0), ("cfg0_goal", g0), ("cfg1_start", s1), ("cfg1_goal", g1), ("cfg0_seeds", x0), ("cfg1_seeds", x1)):
        lines.append(f"vector {name} {np.asarray(v).size} {_fmt(v)}")
    path = tmp_path_factory.mktemp("cpp") / "input.txt"
    path.write_text("\n".join(lines) + "\n")
    return dict(path=str(path), pci0=pci0, pci1=pci1, x0=x0, x1=x1, s0=s0, g0=g0, s1=s1)


def _run(exe, inp, cases, timeout=600):
    p = subprocess.run([exe, inp, cases], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"{cases}: rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    res, init = {}, {}
    for ln in p.stdout.splitlines():
        tok = ln.split()
        if tok and tok[0] == "RESULT":
            res.setdefault(tok[1], []).append(dict(status=int(tok[3]), nfe=int(tok[4]), nqp=int(tok[5]), cost=float.fromhex(tok[6]),
                                                   x=np.array([float.fromhex(t) for t in tok[7:]])))
        elif tok and tok[0] == "INIT":
            init[tok[1]] = np.array([float.fromhex(t) for t in tok[2:]])
    return res, init, p.stdout


def _python_path(pci, x0, lib_path):
    opt = runtime.BatchedTrustRegionSQP(pci, lib_path=lib_path)
    opt.initialize(x0)
    opt.optimize()
    r = opt.results()
    opt.ctx.close()
    return r


def _same(cpp, py):
    assert len(cpp) == py["x"].shape[0]
    for b, c in enumerate(cpp):
        assert c["status"] == py["status"][b] and c["nfe"] == py["n_func_evals"][b] and c["nqp"] == py["n_qp_solves"][b]
        assert c["cost"] == py["total_cost"][b]
        assert np.array_equal(c["x"], py["x"][b].reshape(-1)), "C++ and Python front ends must describe the same problem"


def _check_front_ends(exe, inputs, lib_path):
    res, init, _ = _run(exe, inputs["path"], "cfg0,cfg1")
    _same(res["cfg0"], _python_path(inputs["pci0"], inputs["x0"], lib_path))
    _same(res["cfg1"], _python_path(inputs["pci1"], inputs["x1"], lib_path))
    # JOINT_INTERPOLATED init trajectory: LinSpaced(start, goal) per joint
    line = np.linspace(inputs["s0"], inputs["g0"], 10)
    assert np.abs(init["cfg0"].reshape(10, 7) - line).max() < 1e-14


def test_cpp_front_end_equals_python_front_end_on_host_build(hostemu_lib, inputs):
    _check_front_ends(_build(hostemu_lib, "hostemu"), inputs, hostemu_lib)


def _check_json_front_ends(exe, inputs, lib_path):
    """the reference's problem-description JSON through include/tmx_trajopt_json.hpp and through trajopt_amd/json_io.py"""
    from trajopt_amd import json_io
    res, _, out = _run(exe, inputs["path"], "json")
    assert "JSON done" in out
    ident = np.hstack([np.eye(3), np.zeros((3, 1))])
    for name, fname, pci, x0, start in (("json_cfg0", "planning_unit_cfg0.json", inputs["pci0"], inputs["x0"], inputs["s0"]),
                                        ("json_cfg1", "glass_upright_cfg1.json", inputs["pci1"], inputs["x1"], inputs["s1"])):
        env = json_io.Environment(manipulators={"right_arm": pci.robot}, tip_links={"right_arm": "r_gripper_tool_frame"},
                                  link_frames={"base_footprint": ident}, joint_state={"right_arm": list(start)},
                                  obstacles=list(pci.obstacles))
        pp = json_io.construct_problem(open(os.path.join(HERE, "golden", "json", fname)).read(), env)
        opt = runtime.BatchedTrustRegionSQP(pp.pci, lib_path=lib_path)
        opt.setParameters(pp.sqp_params)
        opt.initialize(x0)
        opt.optimize()
        _same(res[name], opt.results())
        opt.ctx.close()
    assert res["json_numerical_ik1"][0]["status"] == 0


def test_cpp_json_front_end_equals_python_json_front_end_on_host_build(hostemu_lib, inputs):
    _check_json_front_ends(_build(hostemu_lib, "hostemu"), inputs, hostemu_lib)


def _check_time_front_ends(exe, inputs, lib_path):
    """time-parameterised problems (basic_info.use_time) through both JSON readers: same description, same run"""
    from trajopt_amd import json_io
    res, init, out = _run(exe, inputs["path"], "time")
    assert "TIME done" in out
    pci1 = inputs["pci1"]
    env = json_io.Environment(manipulators={"right_arm": pr2_right_arm()}, tip_links={"right_arm": "r_gripper_tool_frame"},
                              link_frames={"base_footprint": np.hstack([np.eye(3), np.zeros((3, 1))])},
                              joint_state={"right_arm": list(inputs["s0"])}, obstacles=list(pci1.obstacles))
    texts = time_json_texts()
    with pytest.raises(ValueError, match="expected: bool"):
        json_io.construct_problem(open(os.path.join(HERE, "golden", "json", "arm_around_table_time.json")).read(), env)
    for name, key in (("time_fixture", "arm_around_table_time_bool"), ("time_terms", "time_terms")):
        pp = json_io.construct_problem(texts[key], env)
        assert pp.pci.basic_info.use_time and pp.init_traj.shape[1] == 8
        # (the two readers restate Eigen's LinSpaced differently in the last bit: both runs start from the C++ reader's trajectory)
        x_init = init[name].reshape(pp.init_traj.shape)
        assert np.abs(x_init - pp.init_traj).max() < 1e-14 and np.array_equal(x_init[:, 7], pp.init_traj[:, 7])
        opt = runtime.BatchedTrustRegionSQP(pp.pci, lib_path=lib_path)
        opt.setParameters(pp.sqp_params)
        opt.initialize(x_init[None])
        opt.optimize()
        _same(res[name], opt.results())
        if name == "time_terms":
            assert pp.pci.cost_names() == [f"vel_t_j{j}" for j in range(7)] + ["total_time"]
            assert pp.pci.cnt_names() == ["joint_pos"] + [f"vel_lim_j{j}" for j in range(7)]
        opt.ctx.close()


def test_cpp_time_problems_equal_python_time_problems_on_host_build(hostemu_lib, inputs):
    _check_time_front_ends(_build(hostemu_lib, "hostemu"), inputs, hostemu_lib)


def test_cpp_reference_kats_on_host_build(hostemu_lib, inputs, orc):
    exe = _build(hostemu_lib, "hostemu")
    res, _, out = _run(exe, inputs["path"], "joint_costs,numerical_ik,cart_position,interface,joint_vel,errors")
    assert "ERRORS done" in out and "INTERFACE done" in out and "JOINTVEL done" in out and "LINKROWS refused" not in out
    assert set(res) == {"equality_jointPos", "inequality_jointPos", "numerical_ik1", "cart_position", "equality_jointVel",
                        "inequality_jointVel", "function_terms", "kinematic_terms"}   # the host build has the two-waypoint rows (TMX_LINK_ROWS=1)
    assert all(r[0]["status"] == 0 for r in res.values())
    # the AvoidSingularity problem of the C++ program through the Python front end: the same lowering
    from trajopt_amd.problem import AvoidSingularityTermInfo, BasicInfo, JointVelTermInfo, ProblemConstructionInfo
    pci = ProblemConstructionInfo(pr2_right_arm(), BasicInfo(n_steps=10))
    pci.cost_infos.append(JointVelTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, first_step=0, last_step=9))
    pci.cost_infos.append(AvoidSingularityTermInfo(link=6, first_step=0, last_step=9, coeffs=[5.0], lambda_=0.1, name="sing"))
    start = np.array([0.3, -0.2, 0.4, -1.0, 0.3, -0.5, 0.2])
    _same(res["kinematic_terms"], _python_path(pci, np.tile(start, (1, 10, 1)), hostemu_lib))


def test_cpp_two_waypoint_terms_are_refused_by_the_product_configuration(hostemu_lib_nolink, inputs):
    """kernel sources in the product's configuration (TMX_LINK_ROWS=0): the JointVel constraint / hinge KATs end in the
    library's explicit refusal, exactly what the GPU tier expects from libtrajopt_mi355x.so this round"""
    exe = _build(hostemu_lib_nolink, "hostemu_nolink")
    res, _, out = _run(exe, inputs["path"], "joint_vel")
    assert out.count("LINKROWS refused") == 3 and "JOINTVEL done" in out and set(res) == {"kinematic_terms"}   # + the acceleration cost of function_terms


def test_cpp_optimizer_fails_loudly_without_a_device(inputs):
    """linked against the PRODUCT library on a machine without a GPU: no CPU path may take over"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert os.path.exists(PRODUCT_LIB), "libtrajopt_mi355x.so missing: run __graft_entry__.build()"
    exe = _build(PRODUCT_LIB, "product")
    _, _, out = _run(exe, inputs["path"], "errors_nodevice")
    assert "ERRORS done" in out


@pytest.mark.gpu
def test_cpp_front_end_on_device(inputs):
    assert os.path.exists(PRODUCT_LIB), "libtrajopt_mi355x.so missing: run __graft_entry__.build()"
    exe = _build(PRODUCT_LIB, "product")
    _check_front_ends(exe, inputs, None)
    _check_json_front_ends(exe, inputs, None)
    _check_time_front_ends(exe, inputs, None)
    res, _, out = _run(exe, inputs["path"], "joint_costs,numerical_ik,cart_position,interface,joint_vel,errors")
    assert "ERRORS done" in out and "INTERFACE done" in out and all(r[0]["status"] == 0 for r in res.values())
    # the product lowers the rows on two consecutive waypoints: the reference's jointVel tests run on the device
    assert "LINKROWS refused" not in out and "JOINTVEL done" in out and "equality_jointVel" in res and "inequality_jointVel" in res
