"""Attitude conversions evaluated on the GPU (C ABI `qd_transform`), exported under the names the reference's
`environments/transformation.py` uses (reference lines 5-29) so that `from environments.transformation import ...`
keeps working.

Conventions (pinned by golden vectors from the reference): quaternions are MuJoCo-ordered (w, x, y, z);
`rpy` = [roll, pitch, yaw] of the intrinsic Z-Y-X decomposition R = Rz(yaw) Ry(pitch) Rx(roll); the tether
orientation is R = Rx(roll) Ry(pitch).  Every function accepts one vector / matrix or a batch of rows and
returns float64 numpy arrays."""
import numpy as np

from .. import _lib as L
from ._device import transform

_TABLE = {
    # exported name: (transform id, input row length, output row length, output item shape or None)
    "mujoco_DCM2quat": (L.TF_DCM2QUAT, 9, 4, None),
    "mujoco_quat2DCM": (L.TF_QUAT2DCM, 4, 9, (3, 3)),
    "mujoco_quat2rpy": (L.TF_QUAT2RPY, 4, 3, None),
    "mujoco_rpy2quat": (L.TF_RPY2QUAT, 3, 4, None),
    "mujoco_pendulumrp2quat": (L.TF_PENDRP2QUAT, 2, 4, None),
}


def _make(name):
    which, n_in, n_out, item = _TABLE[name]

    def convert(x):
        out = transform(which, x, n_in, n_out)
        if item is None:
            return out
        single = np.ndim(x) == 1
        return out.reshape(item) if single else out.reshape((-1,) + item)

    convert.__name__ = convert.__qualname__ = name
    convert.__doc__ = "%s: rows of %d values -> rows of %d values, computed by the device kernel k_transform" % (name, n_in, n_out)
    return convert


for _name in _TABLE:
    globals()[_name] = _make(_name)
del _name
__all__ = list(_TABLE)
