import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _product_library_by_default():
    """Every test starts on the product library (ray3d_amd/libray3d_hip.so).  A test that needs a development switch or an
    r3d_debug_* entry point switches itself to the hooks build (dev_switch / hooks_library below) BEFORE it creates handles;
    the choice ends with the test."""
    from ray3d_amd import _capi
    _capi.use_hooks(False)
    yield
    _capi.use_hooks(False)


def hooks_library():
    """libray3d_hip_hooks.so (the same sources with -DR3D_TEST_HOOKS) for the rest of this test; returns the CDLL."""
    from ray3d_amd import _capi
    _capi.use_hooks(True)
    return _capi.load()


def dev_switch(monkeypatch, name, value):
    """Set a development switch (R3D_NO_SMALL_PLAN, R3D_NO_GEMV, R3D_FAULT_TILE, ...): only the hooks build reads them."""
    hooks_library()
    monkeypatch.setenv(name, str(value))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Measured parity errors, one line per checked configuration: tests call record_parity(); the terminal summary prints
# them (so a run's tail shows numbers, not dots) and gpurun_out/parity_errors.json keeps them when that directory exists.
PARITY = []


def record_parity(label, err, tol, ref_max=None):
    PARITY.append((str(label), float(err), float(tol), None if ref_max is None else float(ref_max)))


def check_parity(got, ref, what="", tol=None, atol=1e-4):
    """max |got - ref| <= tol, recorded under the running test's name (+ `what`).  Default bound: north_star's literal
    1e-4 abs, whatever the magnitude of the outputs (the over-scaled `_big` fixture, outputs of up to 108 m, included)."""
    label = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::", 1)[-1].split(" (")[0] + ((" " + what) if what else "")
    if hasattr(got, "detach"):
        got = got.detach().cpu().numpy()
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (label, got.shape, ref.shape)
    ref_max = float(np.abs(ref).max()) if ref.size else 0.0
    if tol is None:
        tol = atol
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) if ref.size else 0.0
    record_parity(label, err, tol, ref_max)
    assert np.isfinite(got).all(), label
    assert err <= tol, "%s: max abs err %.3e > %.3e (|ref| max %.2f)" % (label, err, tol, ref_max)
    return err


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not PARITY:
        return
    tr = terminalreporter
    tr.section("parity: measured max abs error per configuration (bound: the literal 1e-4 abs unless the row says otherwise)")
    worst = {}
    for label, err, tol, ref_max in PARITY:
        w = worst.get(label)
        if w is None or err > w[0]:
            worst[label] = (err, tol, ref_max)
    LITERAL = 1e-4
    relaxed = {k: v for k, v in worst.items() if v[1] > LITERAL * (1 + 1e-9)}
    strict = {k: v for k, v in worst.items() if k not in relaxed}
    for label, (err, tol, ref_max) in strict.items():
        tr.write_line("parity %-88s err %.2e  bound %.2e%s" % (label, err, tol, "" if ref_max is None else "  |ref|max %.2f" % ref_max))
    for label, (err, tol, ref_max) in relaxed.items():
        tr.write_line("parity RELAXED-BOUND %-74s err %.2e  bound %.2e%s" % (label, err, tol, "" if ref_max is None else "  |ref|max %.2f" % ref_max))
    lit = [e for (e, t, r) in strict.values() if r is not None and r <= 10.0]
    tr.write_line("parity summary: %d configurations at the literal 1e-4 bound (or tighter), worst err %.2e; %d of them with |ref| <= 10 m, worst %.2e"
                  % (len(strict), max([e for e, _, _ in strict.values()] or [0.0]), len(lit), max(lit) if lit else 0.0))
    if relaxed:
        tr.write_line("parity summary: %d configuration(s) under a labelled RELAXED bound (rows above: HIP-against-HIP comparisons whose bound scales with "
                      "|ref|, and the opt-in bf16x3 mode on the over-scaled 108 m fixture), worst err %.2e, widest bound %.2e"
                      % (len(relaxed), max(e for e, _, _ in relaxed.values()), max(t for _, t, _ in relaxed.values())))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "parity_errors.json"), "w") as f:
            json.dump([{"config": k, "max_abs_err": v[0], "bound": v[1], "ref_max": v[2]} for k, v in worst.items()], f, indent=1)


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
