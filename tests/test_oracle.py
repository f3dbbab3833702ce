"""CPU suite, part 1: the oracle itself is pinned against the reference-generated golden vectors
(tests/golden/*.npz, produced by tests/golden/make_golden.py which imports the reference)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, MODEL_CASES, case_out_scale, load_model_fixture, synth_states
from oracle import metrics_oracle, oracle, torch_port

ORACLE_ATOL = 5e-5   # C oracle (double accumulation) vs reference fp32, outputs up to ~14 m


def _tap_from_oracle(taps, name):
    if "_bn" in name:   # reference tap "<Block>.expand_bn" / "<Block>.layers_bn.<i>" (channels-first, window 0)
        blk, rest = name.split(".", 1)
        lvl = 0 if rest == "expand_bn" else int(rest.split(".")[1]) + 1
        return taps["%s.level%d.pre" % (blk, lvl)][0].T
    return taps[name]


@pytest.mark.parametrize("name", MODEL_CASES)
def test_c_oracle_matches_reference(name):
    z, mc = load_model_fixture(name)
    scale = case_out_scale(name)
    for kind, (cfg, st) in zip(("pos", "trj"), synth_states(mc, scale)):
        taps = {}
        out = oracle.forward(cfg, st, z["x"], z["param"], taps=taps)
        ref = z["out_" + kind]
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() <= ORACLE_ATOL * max(1.0, np.abs(ref).max() / 10.0)
        checked = 0
        for k in z.files:
            if k.startswith(kind + "/"):
                r = z[k]
                o = _tap_from_oracle(taps, k.split("/", 1)[1]).reshape(r.shape)
                assert np.abs(o - r).max() <= 5e-5 * max(1.0, np.abs(r).max()), k
                checked += 1
        assert checked >= (5 if not mc["DISABLE_OPTIMIZATIONS"] else 4)
    assert int(z["receptive_field"]) == cfg.receptive_field


@pytest.mark.parametrize("name", MODEL_CASES)
def test_torch_port_matches_reference(name):
    z, mc = load_model_fixture(name)
    x, p = torch.from_numpy(z["x"]), torch.from_numpy(z["param"])
    for kind, (cfg, st) in zip(("pos", "trj"), synth_states(mc, case_out_scale(name))):
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}
        with torch.no_grad():
            out = torch_port.forward(cfg, sd, x, p).numpy()
        assert np.abs(out - z["out_" + kind]).max() <= 1e-6 * max(1.0, np.abs(z["out_" + kind]).max())


def test_state_counts_match_reference_modules():
    # number of tensors in the reference state_dict (incl. num_batches_tracked) per configuration
    from ray3d_amd.spec import state_entries
    for name in MODEL_CASES:
        z, mc = load_model_fixture(name)
        (cp, _), (ct, _) = synth_states(mc)
        assert len(state_entries(cp)) == int(z["n_state_pos"])
        assert len(state_entries(ct)) == int(z["n_state_trj"])


def test_quirk_q1_current_frame_is_not_the_centre():
    from ray3d_amd.spec import config_from_dicts, default_model_config
    cfg = config_from_dicts(default_model_config(ARCHITECTURE="3,3,3,3,3"), "pos")
    assert cfg.receptive_field == 243 and cfg.current_frame == 81   # true centre would be 121
    cfg2 = config_from_dicts(default_model_config(ARCHITECTURE="3,3,3", INPUT_DIM=2), "pos")
    assert cfg2.current_frame == 13                                  # F=2: happens to be the centre


def test_oracle_camera_matches_reference():
    z = np.load(os.path.join(GOLDEN, "cameras.npz"))
    for tag in z["tags"]:
        cam = oracle.Camera(z[tag + "/K"], z[tag + "/R"], z[tag + "/t"])
        assert abs(cam.height - float(z[tag + "/height"])) < 1e-12
        assert abs(cam.pitch - float(z[tag + "/pitch"])) < 1e-12
        for n in ("Rc2n", "Tc2n", "Rn2w", "Tn2w", "Rw2n", "Tw2n"):
            assert np.abs(getattr(cam, n) - z[tag + "/" + n]).max() < 1e-12, (tag, n)
        assert np.abs(cam.rays_from_uv(z[tag + "/uv"]) - z[tag + "/rays"]).max() < 1e-12
        assert np.abs(cam.uv_from_rays(z[tag + "/rays"]) - z[tag + "/uv_back"]).max() < 1e-9
        assert np.abs(cam.world2normalized(z[tag + "/Xw"]) - z[tag + "/Xn"]).max() < 1e-12
        assert np.abs(cam.normalized2world(z[tag + "/Xn"]) - z[tag + "/Xw_back"]).max() < 1e-12


def test_metrics_oracle_matches_reference():
    z = np.load(os.path.join(GOLDEN, "losses.npz"))
    a, b = z["pred"], z["target"]
    assert abs(metrics_oracle.mpjpe(a, b) - float(z["mpjpe"])) < 1e-12
    assert abs(metrics_oracle.n_mpjpe(a, b) - float(z["n_mpjpe"])) < 1e-12
    assert abs(metrics_oracle.p_mpjpe(a.reshape(-1, 17, 3), b.reshape(-1, 17, 3)) - float(z["p_mpjpe"])) < 1e-10
    assert abs(metrics_oracle.mean_velocity_error(a.reshape(-1, 17, 3), b.reshape(-1, 17, 3)) - float(z["mpjve"])) < 1e-12


def test_undistort_is_unpinned_but_self_consistent():
    # cv2.undistortPoints has no fixture (OpenCV absent): only the weak known-answer relations of SURVEY 8c
    z = np.load(os.path.join(GOLDEN, "cameras.npz"))
    K = z["h36m_S9_0/K"]
    dist = np.array([-0.207098910824901, 0.247775183068982, -0.00142447157470321, -0.000975698859470499,
                     -0.00307515035078854])
    pts = np.stack(np.meshgrid(np.linspace(100, 900, 9), np.linspace(100, 900, 9)), -1).reshape(-1, 2)
    und = oracle.undistort_points(K, dist, pts)
    back = oracle.distort_points(K, dist, und)
    assert np.abs(back - pts).max() < 2e-2          # 5 fixed-point iterations: centi-pixel round trip
    pp = np.array([[K[0, 2], K[1, 2]]])
    assert np.abs(oracle.undistort_points(K, dist, pp) - pp).max() < 1e-9   # principal point is a fixed point


def test_oracle_reports_missing_tensor():
    _, mc = load_model_fixture("j17_rf9_s1")
    (cfg, st), _ = synth_states(mc)
    st = dict(st)
    st.pop("GlobalInfo.fc_2.bias")
    x = np.zeros((1, 9, 17, 3), np.float32)
    with pytest.raises(RuntimeError, match="GlobalInfo.fc_2.bias"):
        oracle.forward(cfg, st, x, np.zeros((1, 2), np.float32))
