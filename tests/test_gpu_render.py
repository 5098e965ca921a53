"""Whole-image rendering (svs_hip/renderer.py) against the reference's chunk loop over the same model."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(dev, beta):
    from volsdf.utils.conf import dtu_model_conf
    from volsdf.model.network import VolSDFNetwork
    params = synth.make_params(0)
    m = VolSDFNetwork(dtu_model_conf())
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["density.beta"] = torch.tensor(beta, dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()


@pytest.mark.parametrize("beta", [0.1, 0.05])
def test_render_image_equals_chunk_loop(dev, beta):
    """One 1500-ray launch with 3 convergence groups of 500 rays == the reference's loop of 500-ray forward calls
    (split_input / merge_output, volsdf/utils/general.py:24-58), bit for bit.  The first 500 rays look at an image
    corner (background only): with beta = 0.05 the three groups stop after 1 / 4 / 3 sampler rounds."""
    from svs_hip.renderer import depth_image, render_image
    m = _model(dev, beta)
    K, pose = synth.make_camera()
    N = 1372       # ragged last group, ragged last launch
    uv = synth.make_uv(N, seed=3)
    uv[:500] = np.random.default_rng(0).uniform(0, 30, (500, 2)).astype(np.float32)
    inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(uv)[None].to(dev),
           "pose": torch.from_numpy(pose)[None].to(dev)}
    keys = ("rgb_values", "normal_map", "depth_values", "depth_vals", "weights", "xyz")
    with torch.no_grad():
        res = []
        for lo in range(0, N, 500):                 # the reference's chunking: consecutive 500-ray slices of uv
            o = m(dict(inp, uv=inp["uv"][:, lo:lo + 500]), fast=-1)
            res.append({k: o[k] for k in keys})
        ref = {k: torch.cat([r[k] for r in res], 0) for k in keys}
    for per_launch in (1000, 8000):
        got = render_image(m, inp, N, split_n_pixels=500, rays_per_launch=per_launch)
        for k in keys:
            a, b = got[k].cpu().numpy(), ref[k].cpu().numpy().reshape(got[k].shape)
            assert np.array_equal(a, b), (k, per_launch, float(np.abs(a - b).max()))
    d = depth_image(got, (28, 49))
    assert d.shape == (28, 49) and bool(torch.isfinite(d).all())
    # the groups must really differ in their number of rounds for the small beta (otherwise the test proves nothing)
    if beta < 0.1:
        ctl = m.ray_sampler._ws.ctl.cpu().numpy().reshape(-1, 17)
        assert len(set(ctl[:, 16].tolist())) > 1, ctl[:, 16]
