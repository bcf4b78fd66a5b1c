"""SN_PREC_AUTO's decision (include/stereonet_hip.h: sn_auto_*) as pure functions — no GPU.  The reference loads an opaque
model_file (stereonet_infer/src/stereonet_node.cpp:131-136); the default precision of this library therefore watches the
refinement statistic and leaves the fp16 tower when it says the tower's error would exceed north_star's 1e-3 px.  Here: the
state machine (switch up at once, switch down only after SN_AUTO_CALM_CALLS calm calls), the effect of the self-check's
measured slope on the limit, and the struct layouts the Python binding mirrors."""
import ctypes as C
import re
import os

import pytest

from hobot_stereonet_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "stereonet_hip.h")).read()


def _define(name):
    m = re.search(rf"#define\s+{name}\s+([0-9.eE+-]+)", HEADER)
    assert m, name
    return float(m.group(1))


BUDGET, REENTRY, CALM = _define("SN_AUTO_BUDGET_PX"), _define("SN_AUTO_REENTRY"), int(_define("SN_AUTO_CALM_CALLS"))


@pytest.fixture(scope="module")
def lib():
    return api.load_library()


def _state(lib, levels=1):
    s = api.SnAutoState()
    assert lib.sn_auto_init(C.byref(s), levels) == 0
    return s


def test_abi_version_and_default_precision_are_the_header_s(lib):
    assert lib.sn_abi_version() == api.ABI_VERSION == int(_define("SN_ABI_VERSION"))
    assert re.search(r"SN_PREC_AUTO\s*=\s*4", HEADER) and api.PREC_AUTO == 4
    assert "selects SN_PREC_AUTO" in HEADER


def test_init(lib):
    for levels in (1, 4):
        s = _state(lib, levels)
        assert s.mode == api.PREC_F16 and s.calm == 0 and s.switches == 0 and s.epe_per_px == 0.0 and s.running_px < 0
        assert s.envelope_px == lib.sn_auto_envelope_px(levels) > 0
        assert lib.sn_auto_limit_px(C.byref(s)) == s.envelope_px
    assert lib.sn_auto_init(None, 1) < 0 and lib.sn_auto_observe(None, 1.0) < 0


def test_inside_the_envelope_nothing_happens(lib):
    s = _state(lib)
    for _ in range(100):
        assert lib.sn_auto_observe(C.byref(s), 0.7 * s.envelope_px) == api.PREC_F16
    assert s.switches == 0
    assert abs(s.running_px - 0.7 * s.envelope_px) < 1e-12


def test_one_call_outside_switches_at_once_and_the_way_back_needs_calm_calls(lib):
    s = _state(lib)
    env = s.envelope_px
    assert lib.sn_auto_observe(C.byref(s), 1.01 * env) == api.PREC_F16X3          # the caller repeats that call
    assert s.switches == 1
    # inside the envelope but outside the re-entry band: stays
    for _ in range(3 * CALM):
        assert lib.sn_auto_observe(C.byref(s), 0.9 * env) == api.PREC_F16X3
    # calm calls are counted consecutively: one call outside the band resets the count
    for _ in range(CALM - 1):
        assert lib.sn_auto_observe(C.byref(s), 0.5 * REENTRY * env) == api.PREC_F16X3
    assert lib.sn_auto_observe(C.byref(s), 0.95 * env) == api.PREC_F16X3
    for i in range(CALM):
        m = lib.sn_auto_observe(C.byref(s), 0.5 * REENTRY * env)
        assert m == (api.PREC_F16 if i == CALM - 1 else api.PREC_F16X3)
    assert s.switches == 2 and s.calm == 0


def test_the_self_check_tightens_the_limit(lib):
    s = _state(lib)
    env = s.envelope_px
    # a model whose fp16 tower loses 2e-3 px per pixel of residual: the limit is the budget over that slope
    s.epe_per_px = 2e-3
    lim = lib.sn_auto_limit_px(C.byref(s))
    assert abs(lim - BUDGET / 2e-3) < 1e-12 and lim < env      # (every class envelope is above 0.425 px)
    assert lib.sn_auto_observe(C.byref(s), 0.9 * lim) == api.PREC_F16
    assert lib.sn_auto_observe(C.byref(s), 1.1 * lim) == api.PREC_F16X3
    # a benign model widens it, but only up to the cap
    t = _state(lib)
    t.epe_per_px = BUDGET / (2.0 * env)
    assert abs(lib.sn_auto_limit_px(C.byref(t)) - 2.0 * env) < 1e-9
    t.epe_per_px = 1e-6
    assert lib.sn_auto_limit_px(C.byref(t)) == _define("SN_AUTO_ENVELOPE_CAP") * env


def test_garbage_statistics_are_not_trusted(lib):
    for bad in (float("nan"), -1.0, float("inf")):
        s = _state(lib)
        assert lib.sn_auto_observe(C.byref(s), bad) == api.PREC_F16X3


def test_struct_layouts_match_the_header():
    """Field order of the three structs the binding mirrors (a reordered header would silently shift every value)."""
    def fields(name):
        body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", HEADER, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith("uint64_t") and not decl.startswith("size_t") else decl.split(None, 1)[1]
            for n in names.split(","):
                out.append(re.sub(r"\[.*\]", "", n.strip().lstrip("*")))
        return out
    assert fields("sn_auto_state") == [f for f, _ in api.SnAutoState._fields_]
    assert fields("sn_refine_stats") == [f for f, _ in api.SnRefineStats._fields_]
    assert fields("sn_io_info") == [f for f, _ in api.SnIoInfo._fields_]
    assert fields("sn_config") == [f for f, _ in api.SnConfig._fields_]
