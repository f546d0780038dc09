"""A small cascaded PD hover controller used by the tests as a deterministic CLOSED-LOOP action source
(written for these tests; any stabilising controller serves the purpose).  Works on batches of the
33-element state vectors of BaseDroneEnv.get_drone_states()."""
import numpy as np


def hover_actions(states, ref, rot=None):
    s = np.asarray(states, dtype=np.float64)
    p, rpy, v, w = s[:, 0:3], s[:, 3:6], s[:, 6:9], s[:, 9:12]
    par = s[:, 27:33]
    mass = par[:, 0] * (0.56 + 4 * 0.07 + 4 * 0.04) + 0.01 + 0.2 * par[:, 4] + par[:, 5]
    arm = 1.4142135623730951 * 0.05 + par[:, 1]
    rot = arm * 0.7071067811865476 if rot is None else rot
    F = par[:, 2]
    g = 9.81
    a_des = np.clip(1.0 * (np.asarray(ref)[:3] - p) - 2.2 * v, -3.0, 3.0)
    yaw = rpy[:, 2]
    # swing damping from the tether rates (hinge 2 about body y swings along x, hinge 1 about body x along -y)
    pw = s[:, 14:16]
    sx, sy = -pw[:, 1], pw[:, 0]
    ax = a_des[:, 0] + sx * np.cos(yaw) - sy * np.sin(yaw)
    ay = a_des[:, 1] + sx * np.sin(yaw) + sy * np.cos(yaw)
    pitch_des = np.clip((ax * np.cos(yaw) + ay * np.sin(yaw)) / g, -0.35, 0.35)
    roll_des = np.clip((ax * np.sin(yaw) - ay * np.cos(yaw)) / g, -0.35, 0.35)
    T = mass * (g + a_des[:, 2]) / np.maximum(0.5, np.cos(rpy[:, 0]) * np.cos(rpy[:, 1]))
    yaw_err = (ref[3] - yaw + np.pi) % (2 * np.pi) - np.pi
    tx = 1.0 * (roll_des - rpy[:, 0]) - 0.16 * w[:, 0]
    ty = 1.0 * (pitch_des - rpy[:, 1]) - 0.16 * w[:, 1]
    tz = 0.02 * yaw_err - 0.02 * w[:, 2]
    mix = np.array([[-1, 1, 1, -1], [-1, -1, 1, 1], [1, -1, 1, -1]], dtype=np.float64)
    f = T[:, None] / 4 + tx[:, None] * mix[0] / (4 * rot[:, None]) + ty[:, None] * mix[1] / (4 * rot[:, None]) + \
        tz[:, None] * mix[2] / (4 * 0.01)
    ctrl = f / F[:, None]
    return np.clip((ctrl - 0.1) / 0.9, 0.0, 1.0)
