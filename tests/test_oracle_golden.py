"""The CPU oracle against the golden vectors captured from the reference's own
Python (transformation.py, rewards.py, observation_wrappers.py, BaseDroneEnv,
SimpleDrone).  This pins every part of the oracle except the physics step."""
import json

import numpy as np
import pytest

TOL = 1e-12


def test_transformation_quat2rpy_quat2dcm(golden, orc):
    for q, rpy, dcm in zip(golden["tr_quat"], golden["tr_quat2rpy"], golden["tr_quat2dcm"]):
        np.testing.assert_allclose(orc.quat2dcm(q), dcm, atol=TOL)
        got = orc.quat2rpy(q)
        if abs(abs(rpy[1]) - np.pi / 2) < 1e-6:
            # gimbal lock: scipy zeroes one angle; compare the rotation instead
            np.testing.assert_allclose(orc.quat2dcm(orc.rpy2quat(got)), dcm, atol=1e-7)
        else:
            np.testing.assert_allclose(got, rpy, atol=1e-11)


def test_transformation_rpy2quat_pendrp_dcm2quat(golden, orc):
    for r, q in zip(golden["tr_rpy"], golden["tr_rpy2quat"]):
        np.testing.assert_allclose(orc.rpy2quat(r), q, atol=TOL)
    np.testing.assert_allclose(orc.rpy2quat([0.1, -0.2, 0.3]), [0.98185617, 0.06407135, -0.09115755, 0.1534393],
                               atol=1e-8)  # anchor quoted in SURVEY.md 8c
    for p, q in zip(golden["tr_prp"], golden["tr_pendrp2quat"]):
        np.testing.assert_allclose(orc.pendrp2quat(p), q, atol=TOL)
    for q, q2 in zip(golden["tr_quat"], golden["tr_dcm2quat"]):
        got = orc.dcm2quat(orc.quat2dcm(q))
        # same rotation; sign convention as scipy's from_matrix
        np.testing.assert_allclose(got, q2, atol=1e-10)


REWARDS = ["default_reward_fcn", "distance_reward_fcn", "distance_energy_reward",
           "distance_energy_reward_pendulum_angle", "distance_energy_reward_pendulum_angle2",
           "distance_energy_reward_pendulum_angle3", "distance_energy_reward_pendulum_en",
           "distance_energy_reward_pendulum_en2", "distance_energy_reward_pendulum_en3",
           "distance_energy_reward_pendulum_en4", "distance_time_energy_reward", "reward_1", "reward_pendulum_dist",
           "reward_pendulumDistHeading", "reward_2", "reward_2_penergy", "reward_3"]


@pytest.mark.parametrize("name", REWARDS)
def test_rewards_33(golden, orc, name):
    kind = orc.REWARD_KINDS.index(name)
    S, A, K, ref = golden["st33"], golden["st_actions"], golden["st_num_steps"], golden["st_ref"]
    md = float(golden["st_max_distance"])
    got = np.array([orc.reward(kind, S[i], A[i], K[i], ref, md) for i in range(len(S))])
    np.testing.assert_allclose(got, golden["rew_" + name], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("name", REWARDS[:6] + ["distance_time_energy_reward", "reward_1"])
def test_rewards_29_noload_quirk(golden, orc, name):
    kind = orc.REWARD_KINDS.index(name)
    S, A, K, ref = golden["st29"], golden["st_actions"], golden["st_num_steps"], golden["st_ref"]
    md = float(golden["st_max_distance"])
    got = np.array([orc.reward(kind, S[i], A[i], K[i], ref, md) for i in range(len(S))])
    np.testing.assert_allclose(got, golden["rew29_" + name], rtol=1e-11, atol=1e-11)


def test_truncation(golden, orc):
    S, K, ref = golden["st33"], golden["st_num_steps"], golden["st_ref"]
    got = [orc.truncated(S[i], ref, K[i], float(golden["st_max_distance"]), int(golden["st_max_steps"]))
           for i in range(len(S))]
    assert got == list(golden["trunc33"])
    assert any(got) and not all(got)


OBS = ["GlobalFrameRPYEnv", "LocalFramePRYEnv", "LocalFrameFullStateEnv", "LocalFrameFullStateZvecEnv",
       "LocalFramePRYaccEnv", "LocalFramePRYParamsEnv", "LocalFramePRYaccParamsEnv", "LocalFrameRPYParamsEnv",
       "LocalFrameRPYFakeParamsEnv", "LocalFrameRPYEnv", "LocalFramePRYaccNoPendEnv", "LocalFrameRmParamsEnv",
       "LocalFrameZvecEnv"]


@pytest.mark.parametrize("name", OBS)
@pytest.mark.parametrize("tag", ["33", "29"])
def test_observations(golden, orc, name, tag):
    kind = orc.OBS_KINDS.index(name)
    S, ref = golden["st" + tag], golden["st_ref"]
    want = golden["obs%s_%s" % (tag, name)]
    got = np.array([orc.obs(kind, S[i], ref) for i in range(len(S))])
    assert got.shape == want.shape
    assert orc.obs_dim(kind, int(tag)) == want.shape[1]
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11)


def test_broken_variant_raises_like_reference(golden, orc):
    assert str(golden["obs_broken_variant_error"]) == "NameError"
    with pytest.raises(NameError):
        orc.obs(orc.OBS_KINDS.index("LocalFramePRYaccParamsNoPendEnv"), golden["st33"][0], golden["st_ref"])


@pytest.mark.parametrize("load", [1, 0])
def test_get_drone_states(golden, orc, load):
    k = "gds%d_" % load
    nq, nv = (9, 8) if load else (7, 6)
    qpos, qvel, sens, act = golden[k + "qpos"], golden[k + "qvel"], golden[k + "sens"], golden[k + "act"]
    want = golden[k + "states"]
    for i in range(len(want)):
        got = orc.drone_state(load, qpos[nq * i:nq * (i + 1)], qvel[nv * i:nv * (i + 1)], sens[3 * i:3 * i + 3],
                              act[4 * i:4 * i + 4], golden["st_ref"], golden[k + "params"][i])
        assert len(got) == (33 if load else 29)
        np.testing.assert_allclose(got, want[i], atol=1e-11)


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_sample_state_transform(golden, orc, case):
    k = "ss%d_" % case
    cc = json.loads(str(golden[k + "cfg"]))
    sd = cc["state_difficulty"]
    cfg = orc.sample_cfg(cc["pendulum"], cc["random_start_pos"], cc["start_pos"], sd * cc["max_random_offset"],
                         sd * np.array(cc["angle_variance"]), sd * np.array(cc["vel_variance"]),
                         sd * np.array(cc["ang_vel_variance"]), sd * np.array(cc["pendulum_rp_variance"]),
                         sd * np.array(cc["pendulum_ang_vel_variance"]))
    for z, u, qp, qv in zip(golden[k + "z"], golden[k + "u"], golden[k + "qpos"], golden[k + "qvel"]):
        gp, gv = orc.sample_state_from_draws(cfg, z, u)
        np.testing.assert_allclose(gp, qp, atol=1e-12)
        np.testing.assert_allclose(gv, qv, atol=1e-12)


def test_sample_state_survey_anchor(golden):
    np.testing.assert_allclose(golden["ss_anchor_qpos"],
                               [0.1984509, 0.55105994, 14.52422279, 0.70311417, 0, 0, -0.71107698, -0.07133828,
                                -0.15972511], atol=1e-7)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_generate_drone_params_transform(golden, case):
    k = "gp%d_" % case
    cc = json.loads(str(golden[k + "cfg"]))
    c, w, u = golden["gp_centers"], golden["gp_widths"], golden[k + "u"]
    want = golden[k + "params"]
    for i in range(want.shape[0]):
        if cc["random_params"]:
            raw = c + (-w + (w - (-w)) * u[:, i]) * cc["param_difficulty"]
        else:
            raw = c.copy()
        if not cc["pendulum"]:
            raw[4:] = 0.0
        np.testing.assert_allclose(raw, want[i], rtol=0, atol=1e-15)


def test_simple_drone_obs_reward(golden, orc):
    qpos, want = golden["sd_qpos"], golden["sd_obs"]
    got = np.concatenate([orc.simple_obs(qpos[7 * i:7 * i + 7]) for i in range(len(qpos) // 7)])
    np.testing.assert_allclose(got, want, atol=1e-11)
    d = np.linalg.norm(got[:3] - np.array([0, 0, 1.0]))
    assert abs((0.1 - d) - float(golden["sd_reward"])) < 1e-12
    assert bool(d > 0.5) == bool(golden["sd_terminated"])
    r = orc.reward(orc.REWARD_KINDS.index("simple_drone"), np.concatenate([got[:3], np.zeros(30)]), np.zeros(4), 0,
                   [0, 0, 1, 0], 0.5)
    assert abs(r - float(golden["sd_reward"])) < 1e-12


def test_philox_known_answers(orc):
    # Random123 known-answer vectors for philox4x32-10
    assert orc.philox4x32([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


# ---------------------------------------------------------------- 8f-3: analytic PID cascade
def test_pid_cascade_matches_reference_controllers(golden, orc):
    G, oracle = golden, orc
    """models/Analytic/PositionController.py + AttitudeController.py over a 12-step sequence with state
    (first-step derivative suppression, integrators, error / output clips, the non-orthonormal Rd)."""
    pid = oracle.Pid(G["pid_masses"], G["pid_forces"])
    ref = G["pid_ref"]
    T = G["pid_xyz"].shape[1]
    for t in range(T):
        pa = pid.position(ref[:3], G["pid_xyz"][:, t].T)
        np.testing.assert_allclose(pa.T, G["pid_pos_action"][t], rtol=0, atol=1e-12)
        rz = oracle.Pid.tilts2rpy(pa, ref[3])
        np.testing.assert_allclose(rz.T, G["pid_rpyz"][t], rtol=0, atol=1e-12)
        ct = pid.attitude(rz, G["pid_rpy"][:, t].T)
        np.testing.assert_allclose(ct, G["pid_ctrl"][t], rtol=0, atol=1e-12)
    # the fused entry point gives clip(ctrl - 0.1, 0, 1) of the same sequence
    pid2 = oracle.Pid(G["pid_masses"], G["pid_forces"])
    for t in range(T):
        a = pid2.action(ref, G["pid_xyz"][:, t].T, G["pid_rpy"][:, t].T)
        np.testing.assert_allclose(a, np.clip(G["pid_ctrl"][t] - 0.1, 0, 1), rtol=0, atol=1e-12)


# ---------------------------------------------------------------- 8f-4: waypoint generators
def test_trajectory_generators_match_reference(golden, orc):
    """gen_circle_trajectory / gen_step_trajectory / gen_ramp_trajectory (evaluation.py:135-152)"""
    G = golden
    T, f, r, h = G["traj_circle_args"]
    z = np.zeros(4)
    np.testing.assert_allclose(orc.trajectory(1, [f, r, h], z, z, 0.01, len(G["traj_circle"])), G["traj_circle"], atol=1e-13)
    s, e = G["traj_start"], G["traj_end"]
    np.testing.assert_array_equal(orc.trajectory(2, [G["traj_step_args"][0]], s, e, 0.01, len(G["traj_step"])), G["traj_step"])
    np.testing.assert_allclose(orc.trajectory(3, G["traj_ramp_args"], s, e, 0.01, len(G["traj_ramp"])), G["traj_ramp"], atol=1e-13)
    np.testing.assert_array_equal(orc.trajectory(2, [5.0], [0, 0, 0, 0], [0, 0, 1, 0], 0.01, 1000), G["traj_step_default"])
    np.testing.assert_allclose(orc.trajectory(3, [5.0, 10.0], [0, 0, 0, 0], [0, 0, 1, 0], 0.01, 1000), G["traj_ramp_default"], atol=1e-13)
