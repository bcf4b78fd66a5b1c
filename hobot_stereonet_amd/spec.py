"""SN-K4 network specification — the layer table shared by the
weight-file writer and the host-side weight packer.

The reference ships no network arithmetic (its model is the opaque BPU binary
``hobot_stereonet.hbm``, stereonet_infer/include/stereonet_node.h:121); the only
pinned facts are the tensor contract (int8 1x6xHxW in, int32 1x1xHxW out,
stereonet_infer/src/stereonet_node.cpp:63-72,282-288) and ``16*12``
(stereonet_infer/src/parser.cpp:86) = 1/16-resolution cost volume x 12 planes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

C = 32                 # feature channels
N_DOWN = 4             # 5x5 stride-2 convs -> 1/16 resolution
N_FEAT_RES = 6
N_AGG = 4
N_REF_RES = 6
REF_DILATIONS = (1, 2, 4, 8, 1, 1)
LRELU_SLOPE = 0.2
OUT_SCALE = 2.60443857769133e-6   # stereonet_node.cpp:282, publisher_member_function.py:29
DEFAULT_W, DEFAULT_H, DEFAULT_D = 1280, 720, 192
# Hierarchical ("multi") refinement, SURVEY.md appendix A: the same tower with SEPARATE weights applied at 1/8, 1/4, 1/2
# and full resolution, x2 bilinear between the levels.  Level k works at 1/2^k resolution; level 0 is the `single` tower.
MULTI_LEVELS = 4


@dataclass(frozen=True)
class Layer:
    name: str
    w_shape: Tuple[int, ...]   # PyTorch layout: [co, ci, (kd,) kh, kw]
    b_shape: Tuple[int, ...]

    @property
    def w_numel(self) -> int:
        n = 1
        for s in self.w_shape:
            n *= s
        return n

    @property
    def b_numel(self) -> int:
        return self.b_shape[0]

    @property
    def fan_in(self) -> int:
        return self.w_numel // self.w_shape[0]


def ref_prefix(level: int) -> str:
    """Name prefix of the refinement tower of `level` (0 = full resolution = the `single` tower)."""
    return "ref" if level == 0 else f"ref{level}"


def tower_layers(prefix: str) -> List[Layer]:
    out: List[Layer] = [Layer(f"{prefix}.in", (C, 4, 3, 3), (C,))]
    for i in range(N_REF_RES):
        for j in (1, 2):
            out.append(Layer(f"{prefix}.res{i}.{j}", (C, C, 3, 3), (C,)))
    out.append(Layer(f"{prefix}.out", (1, C, 3, 3), (1,)))
    return out


def layers(levels: int = 1) -> List[Layer]:
    """Canonical tensor order of the .snw weight file (weight, then bias, per layer).  `levels` > 1: the towers of
    the coarser refinement levels 1 .. levels-1 follow the single-scale network, so a multi file starts with a
    complete single file."""
    out: List[Layer] = []
    for i in range(N_DOWN):
        out.append(Layer(f"feat.down{i}", (C, 3 if i == 0 else C, 5, 5), (C,)))
    for i in range(N_FEAT_RES):
        for j in (1, 2):
            out.append(Layer(f"feat.res{i}.{j}", (C, C, 3, 3), (C,)))
    out.append(Layer("feat.out", (C, C, 3, 3), (C,)))
    for i in range(N_AGG):
        out.append(Layer(f"agg.conv{i}", (C, C, 3, 3, 3), (C,)))
    out.append(Layer("agg.out", (1, C, 3, 3, 3), (1,)))
    for level in range(levels):
        out.extend(tower_layers(ref_prefix(level)))
    return out


def offsets(levels: int = 1) -> dict:
    """name + '.w' / '.b' -> (offset, shape) into the flat fp32 blob."""
    off = 0
    table = {}
    for l in layers(levels):
        table[l.name + ".w"] = (off, l.w_shape)
        off += l.w_numel
        table[l.name + ".b"] = (off, l.b_shape)
        off += l.b_numel
    table["__total__"] = (off, ())
    return table


def param_count(levels: int = 1) -> int:
    return offsets(levels)["__total__"][0]


def levels_of(n_params: int) -> int:
    """Refinement levels of a blob of n_params floats (1 or MULTI_LEVELS); ValueError otherwise."""
    for lv in (1, MULTI_LEVELS):
        if n_params == param_count(lv):
            return lv
    raise ValueError(f"{n_params} parameters match neither the single nor the multi SN-K4 network")


def ceil16(v: int) -> int:
    return (v + 15) // 16 * 16


def flops_per_pair(w: int, h: int, d: int, refine: bool = True, levels: int = 1) -> float:
    """Algorithmic FLOPs (2*MAC, convs only) for one stereo pair — SURVEY.md appendix A."""
    wp, hp = ceil16(w), ceil16(h)
    wl, hl, dl = wp // 16, hp // 16, d // 16
    mac = 0
    for k in range(1, N_DOWN + 1):
        cin = 3 if k == 1 else C
        mac += 2 * (wp * hp // 4 ** k) * C * cin * 25
    mac += 2 * (2 * N_FEAT_RES + 1) * wl * hl * C * C * 9
    mac += N_AGG * dl * hl * wl * C * C * 27 + dl * hl * wl * C * 27
    if refine:
        for k in range(levels):
            mac += (wp * hp // 4 ** k) * (4 * C * 9 + 2 * N_REF_RES * C * C * 9 + C * 9)
    return 2.0 * mac
