import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _parse(v):
    if v in ("True", "False"):
        return v == "True"
    try:
        return int(v)
    except ValueError:
        try:
            return float(v)
        except ValueError:
            return v


def load_model_fixture(name):
    """tests/golden/model_<name>.npz -> (npz, model_config dict)."""
    z = np.load(os.path.join(GOLDEN, "model_%s.npz" % name))
    mc = {k: _parse(str(v)) for k, v in zip(z["model_config_keys"], z["model_config_vals"])}
    mc["ARCHITECTURE"] = str(dict(zip(z["model_config_keys"], z["model_config_vals"]))["ARCHITECTURE"])
    return z, mc


MODEL_CASES = ["j17_rf27_s3", "j17_rf243_s3", "j17_rf9_s1", "j14_rf9_s3", "j15_rf9_s3",
               "j17_f2_rf27_noemb_s3", "j17_rf81_s2_big", "j17_rf27_dilated_s3", "j17_rf81_causal_s3",
               "j17_rf27_dense_s3", "j14_rf9_dense_causal_s2"]


def case_out_scale(name):
    return 8.0 if name.endswith("_big") else 1.0


def synth_states(mc, out_scale=1.0):
    """Deterministic weights exactly as tests/golden/make_golden.py generated them."""
    from ray3d_amd import synth
    from ray3d_amd.spec import config_from_dicts
    cp, ct = config_from_dicts(mc, "pos"), config_from_dicts(mc, "trj")
    return (cp, synth.synth_state(cp, seed=1, out_scale=out_scale)), (ct, synth.synth_state(ct, seed=2, out_scale=out_scale))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
