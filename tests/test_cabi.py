"""The C-ABI library builds, loads without a GPU, exports every symbol include/stereonet_hip.h declares,
and fails loudly (never falls back to a CPU path) when no gfx950 device is present."""
import ctypes
import os
import re

import pytest

from hobot_stereonet_amd import api, build, weights

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "stereonet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(build.build())
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in stereonet_hip.h but not exported"


def test_strerror_and_codes():
    assert api.error_string(0) == "ok"
    assert "gfx950" in api.error_string(-4)
    assert api.error_string(-99) == "unknown error"


def test_create_without_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = str(tmp_path / "m.snw")
    weights.save_snw(p, weights.synthetic(0))
    with pytest.raises(api.StereoNetError) as e:
        api.StereoNetHIP(p)
    assert e.value.code == -4        # SN_ERR_DEVICE: no silent CPU path


def test_product_does_not_touch_oracle():
    # the oracle is test infrastructure: nothing under hobot_stereonet_amd/ may reference it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hobot_stereonet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in txt and "libstereonet_oracle" not in txt and "so_forward" not in txt, f
