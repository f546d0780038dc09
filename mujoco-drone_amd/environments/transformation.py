"""Attitude conversions with the reference's names and conventions
(environments/transformation.py:5-29), evaluated on the GPU (qd_transform).

MuJoCo quaternions are (w,x,y,z); rpy is [roll, pitch, yaw] of the intrinsic ZYX
decomposition R = Rz(yaw) Ry(pitch) Rx(roll); the pendulum rotation is Rx(r) Ry(p).
Each function takes one vector/matrix like the reference, or a batch of rows."""
from .. import _lib as L
from ._device import transform


def mujoco_DCM2quat(DCM):
    """rotation matrix -> MuJoCo quaternion (transformation.py:5-8)"""
    return transform(L.TF_DCM2QUAT, DCM, 9, 4)


def mujoco_quat2DCM(quat):
    """MuJoCo quaternion -> rotation matrix (transformation.py:11-13)"""
    import numpy as np
    out = transform(L.TF_QUAT2DCM, quat, 4, 9)
    return out.reshape(3, 3) if np.ndim(quat) == 1 else out.reshape(-1, 3, 3)


def mujoco_quat2rpy(quat):
    """MuJoCo quaternion -> [roll, pitch, yaw] (transformation.py:16-18)"""
    return transform(L.TF_QUAT2RPY, quat, 4, 3)


def mujoco_rpy2quat(rpy):
    """[roll, pitch, yaw] -> MuJoCo quaternion (transformation.py:21-24)"""
    return transform(L.TF_RPY2QUAT, rpy, 3, 4)


def mujoco_pendulumrp2quat(pendulum_rp):
    """pendulum [roll, pitch] (intrinsic XY) -> MuJoCo quaternion (transformation.py:27-29)"""
    return transform(L.TF_PENDRP2QUAT, pendulum_rp, 2, 4)
