"""Checkpoint interchange of the optimiser state (volsdf/vsdf.py:143-145,181-195 of the reference): torch.optim.Adam
numbers its state by position in `model.parameters()` (bias, weight_g, weight_v for a weight-normed layer), the fused
optimiser keeps flat buffers in kernel order (weight_v, weight_g, bias).  `AdamStateView` must translate by parameter
identity.  Host logic only: runs on CPU tensors, no kernel is launched."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))


def _setup(conf_name):
    from svs_hip.trainer import FlatParams, FusedAdam
    from volsdf.utils import conf
    from volsdf.vsdf import AdamStateView
    if conf_name == "dtu":
        from volsdf.model.network import VolSDFNetwork as Model
        c = conf.dtu_model_conf()
    else:
        from volsdf.model.network_bg import VolSDFNetworkBG as Model
        c = conf.bmvs_model_conf()
    torch.manual_seed(0)
    model = Model(c)
    fused = FusedAdam(FlatParams(model._flat_param_list()), lr=5e-4)
    return model, fused, AdamStateView(fused, model)


@pytest.mark.parametrize("conf_name", ["dtu", "bmvs"])
def test_state_round_trip_through_a_real_torch_adam(conf_name):
    model, fused, view = _setup(conf_name)
    names = [n for n, _ in model.named_parameters()]
    params = list(model.parameters())
    flat_names = {id(p): n for n, p in model.named_parameters()}
    # torch's order really differs from the flat order (otherwise this test proves nothing)
    assert [flat_names[id(p)] for p in fused.fp.params] != names
    assert sorted(flat_names[id(p)] for p in fused.fp.params) == sorted(names)

    # a reference optimiser over model.parameters() -- what vsdf.py:101 builds -- takes two real steps
    ref = torch.optim.Adam(params, lr=5e-4)
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)
        ref.step()
    sd = ref.state_dict()
    view.load_state_dict(sd)
    assert fused.step_count == 2
    mom = dict(zip((flat_names[id(p)] for p in fused.fp.params), zip(fused.fp.views(fused.exp_avg), fused.fp.views(fused.exp_avg_sq))))
    for i, n in enumerate(names):
        assert mom[n][0].shape == params[i].shape, n
        assert torch.equal(mom[n][0], sd["state"][i]["exp_avg"]), n
        assert torch.equal(mom[n][1], sd["state"][i]["exp_avg_sq"]), n

    # and back: what the view writes loads into a fresh torch Adam over model.parameters() with identical moments
    out = view.state_dict()
    ref2 = torch.optim.Adam(params, lr=1e-3)
    ref2.load_state_dict(out)
    st2 = ref2.state_dict()["state"]
    assert len(st2) == len(params) and ref2.param_groups[0]["lr"] == pytest.approx(5e-4)
    for i in range(len(params)):
        assert float(st2[i]["step"]) == 2.0
        assert torch.equal(st2[i]["exp_avg"], sd["state"][i]["exp_avg"]), names[i]
        assert torch.equal(st2[i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"]), names[i]


def test_state_in_the_wrong_order_is_rejected():
    model, fused, view = _setup("dtu")
    before = fused.exp_avg.clone()
    fused.exp_avg.fill_(3.0)
    # a state numbered in the flat (kernel) order: shapes disagree with model.parameters() at index 0
    state = {i: {"step": torch.tensor(1.0), "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
             for i, p in enumerate(fused.fp.params)}
    with pytest.raises(ValueError, match="has shape"):
        view.load_state_dict({"state": state, "param_groups": [{"lr": 1e-3}]})
    assert torch.equal(fused.exp_avg, torch.full_like(before, 3.0))          # nothing was copied before the check
    with pytest.raises(ValueError, match="parameters"):
        view.load_state_dict({"state": {999: state[0]}, "param_groups": []})
