"""-m gpu: the HIP path (through the C ABI) against the reference-generated golden fixtures and
the CPU oracle.  Tolerance: 1e-4 abs on fp32 3D joint positions of metre scale (BASELINE.json
north_star); for the deliberately over-scaled '_big' fixture the same bound relative to 10 m."""
import numpy as np
import pytest
import torch

from conftest import MODEL_CASES, case_out_scale, load_model_fixture, synth_states

pytestmark = pytest.mark.gpu

ATOL = 1e-4


def tol_for(ref):
    return ATOL * max(1.0, float(np.abs(ref).max()) / 10.0)


def build_modules(mc, out_scale=1.0, device="cuda:0"):
    import ray3d_amd
    (cp, sp), (ct, st) = synth_states(mc, out_scale)
    fac = ray3d_amd.Model(mc, {}, is_train=False)
    pos, trj = fac.get_pos_model(), fac.get_trj_model()
    ray3d_amd.load_weight(pos, {k: torch.from_numpy(np.asarray(v)) for k, v in sp.items()})
    ray3d_amd.load_weight(trj, {k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    pos.eval(), trj.eval()
    return pos, trj, (cp, sp), (ct, st)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_modules_match_reference_fixture(name):
    z, mc = load_model_fixture(name)
    pos, trj, _, _ = build_modules(mc, case_out_scale(name))
    x = torch.from_numpy(z["x"]).cuda()
    p = torch.from_numpy(z["param"]).cuda()
    with torch.no_grad():
        op = pos(x, p).cpu().numpy()
        ot = trj(x, p).cpu().numpy()
    assert op.shape == z["out_pos"].shape and ot.shape == z["out_trj"].shape
    ep, et = np.abs(op - z["out_pos"]).max(), np.abs(ot - z["out_trj"]).max()
    print(name, "pos err %.2e trj err %.2e" % (ep, et))
    assert ep <= tol_for(z["out_pos"]), ep
    assert et <= tol_for(z["out_trj"]), et


@pytest.mark.parametrize("name", MODEL_CASES)
def test_lifter_pair_matches_reference_fixture(name):
    import ray3d_amd
    z, mc = load_model_fixture(name)
    pos, trj, _, _ = build_modules(mc, case_out_scale(name))
    lifter = ray3d_amd.Ray3DLifter(pos, trj).eval()
    x = torch.from_numpy(z["x"]).cuda()
    p = torch.from_numpy(z["param"]).cuda()
    with torch.no_grad():
        out, out_trj = lifter(x, p, return_trj=True)
    ref = z["out_pos"] + z["out_trj"]
    assert np.abs(out.cpu().numpy() - ref).max() <= tol_for(ref)
    assert np.abs(out_trj.cpu().numpy() - z["out_trj"]).max() <= tol_for(z["out_trj"])
