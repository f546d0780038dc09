#!/usr/bin/env python3
"""Print the constexpr SProg table (csrc/qd_policy_static.h) of a policy family from its layer program in
mujoco_drone_amd/policy.py, so the compile-time specialisation and the host program cannot drift apart.

    python tools/emit_policy_arch.py DSN_LSTM_model ArchDsnLstm
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_drone_amd import _lib as L  # noqa: E402
from mujoco_drone_amd.policy import compile_program, random_weights  # noqa: E402

KIND = {L.POL_DENSE: "POL_DENSE", L.POL_AFFINE: "POL_AFFINE", L.POL_COPY_OBS: "POL_COPY_OBS", L.POL_COPY_PREV: "POL_COPY_PREV",
        L.POL_RING_LOAD: "POL_RING_LOAD", L.POL_RING_PUSH: "POL_RING_PUSH", L.POL_LSTM_CELL: "POL_LSTM_CELL"}


def main():
    family, name = sys.argv[1], sys.argv[2]
    dims = {"CNNestimator": dict(obs_dim=23, num_states=23), "CNNestimator_estimate": dict(obs_dim=23, num_states=23),
            "LSTMestimator": dict(obs_dim=19, num_states=19), "LSTMestimator_estimate": dict(obs_dim=19, num_states=19)}.get(family, {})
    d, ops, _ = compile_program(family, random_weights(family, 0), **dims)
    rows = []
    for o in ops:
        rows.append("{%s, %d, %d, %d, %d, %d, %d, %s, %s}" % (KIND[o.kind], o.in_buf, o.in_off, o.in_dim, o.out_buf, o.out_off, o.out_dim,
                                                             {0: "0", 1: "TANH", 2: "POL_ACT_RELU"}[o.act], "SV" if o.flags else "0"))
    lines, cur = [], "      {"
    for k, r in enumerate(rows):
        piece = r + (", " if k + 1 < len(rows) else "},")
        if len(cur) + len(piece) > 150:
            lines.append(cur.rstrip())
            cur = "       "
        cur += piece
    lines.append(cur)
    widths = ", ".join(str(d.buf_width[b]) for b in range(d.n_bufs))
    rings = ", ".join("{%d, %d, %d}" % (d.ring[r].rows, d.ring[r].width, d.ring[r].period) for r in range(d.n_rings))
    print("struct %s {  // %s (mujoco_drone_amd/policy.py; table printed by tools/emit_policy_arch.py)" % (name, family))
    print("  static constexpr SProg prog = {%d," % d.n_ops)
    print("\n".join(lines))
    print("      %d, {%s}, %d, %d, %d, %d, %d, %d, %d," % (d.n_bufs, widths, d.obs_dim, d.act_dim, d.logits_buf, d.logits_off, d.n_logits,
                                                          d.value_buf, d.value_off))
    print("      %d, {%s}, %d, %d, %d};" % (d.n_rings, rings, d.aux_buf, d.aux_off, d.aux_dim))
    print("};")


if __name__ == "__main__":
    main()
