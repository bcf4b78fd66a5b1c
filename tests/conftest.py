import os
import sys

import pytest
import torch  # noqa: F401  first HIP-linked import of the session: one HIP runtime per process (api.load_library)

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def weights_blob():
    from hobot_stereonet_amd import weights
    return weights.synthetic(0)


@pytest.fixture(scope="session")
def weights_multi():
    """Seed-0 blob of the hierarchical-refinement (`multi`) network: starts with weights_blob."""
    from hobot_stereonet_amd import spec, weights
    return weights.synthetic(0, spec.MULTI_LEVELS)


@pytest.fixture(scope="session")
def golden_multi():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "network_multi_golden.npz"))


@pytest.fixture(scope="session")
def golden_net():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "network_golden.npz"))


@pytest.fixture(scope="session")
def golden_pre():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "preprocess_golden.npz"))


@pytest.fixture(scope="session")
def model_factory(tmp_path_factory, weights_blob):
    """-> f(w, h, d, multi=False) -> path of an SNW1 model file with seed-0 synthetic weights."""
    from hobot_stereonet_amd import spec, weights
    made = {}

    def make(w, h, d, multi=False):
        key = (w, h, d, multi)
        if key not in made:
            p = str(tmp_path_factory.mktemp("models") / f"sn_{w}x{h}_d{d}{'_multi' if multi else ''}.snw")
            weights.save_snw(p, weights.synthetic(0, spec.MULTI_LEVELS) if multi else weights_blob, w, h, d)
            made[key] = p
        return made[key]

    return make
