"""State-divergence report shared by the trajectory tests: per component group of (qpos, qvel, act) the three numbers the
BASELINE target can be read as --
  abs   max |delta|                         (native units: m, rad, m/s, rad/s, activation)
  rel   max |delta| / max |reference|       (pure relative, against the group's own scale over the compared samples; the
                                            scale is floored at 1e-3 so that a group that stays at zero -- the angular rate
                                            of a hover -- is not divided by rounding noise)
  mixed max |delta| / max(1, |reference|)   (element-wise; what the 1e-4 bar of round 1 was measured with)
"""
import numpy as np

GROUPS_LOAD = {"pos": ("qpos", slice(0, 3)), "quat": ("qpos", slice(3, 7)), "hinge": ("qpos", slice(7, 9)),
               "vel": ("qvel", slice(0, 3)), "angvel": ("qvel", slice(3, 6)), "hinge_rate": ("qvel", slice(6, 8)),
               "act": ("act", slice(0, 4))}
GROUPS_NOLOAD = {k: v for k, v in GROUPS_LOAD.items() if k not in ("hinge", "hinge_rate")}


class Divergence:
    def __init__(self, load=True):
        self.groups = GROUPS_LOAD if load else GROUPS_NOLOAD
        self.worst = {g: dict(abs=0.0, rel=0.0, mixed=0.0, scale=0.0) for g in self.groups}

    def update(self, got, want):
        """got / want: dicts with 'qpos' [n,nq], 'qvel' [n,nv], 'act' [n,4] (float64)"""
        for g, (key, sl) in self.groups.items():
            a, b = np.asarray(got[key], dtype=np.float64)[:, sl], np.asarray(want[key], dtype=np.float64)[:, sl]
            d = np.abs(a - b)
            w = self.worst[g]
            w["abs"] = max(w["abs"], float(d.max()))
            w["scale"] = max(w["scale"], float(np.abs(b).max()))
            w["mixed"] = max(w["mixed"], float((d / np.maximum(1.0, np.abs(b))).max()))
        for w in self.worst.values():
            w["rel"] = w["abs"] / max(w["scale"], 1e-3)

    def max(self, what):
        return max(w[what] for w in self.worst.values())

    def table(self, title=""):
        rows = ["%s%-11s %10s %10s %10s %10s" % (title + "\n" if title else "", "group", "abs", "rel", "mixed", "scale")]
        for g, w in self.worst.items():
            rows.append("%-11s %10.2e %10.2e %10.2e %10.2e" % (g, w["abs"], w["rel"], w["mixed"], w["scale"]))
        return "\n".join(rows)
