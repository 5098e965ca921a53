"""Host logic of the configuration plumbing behind VolOpt (volsdf/utils/conf.py): pyhocon-style accessors and the
attribute / item view of the hydra args object."""
import types

import pytest

from volsdf.utils.conf import Conf, attr_view, to_plain


def test_conf_accessors():
    c = Conf(dict(train=dict(expname="ours", num_pixels="512", lr=5.0e-4, flag=0, dims=(1, 2)), model=dict(a=dict(b=3))))
    assert c.get_string("train.expname") == "ours" and c.get_int("train.num_pixels") == 512
    assert c.get_float("train.lr") == 5e-4 and c.get_bool("train.flag") is False and c.get_list("train.dims") == [1, 2]
    sub = c.get_config("model")
    assert isinstance(sub, Conf) and sub.get_int("a.b") == 3 and sub.get_int("a.c", default=7) == 7
    assert c.get_string("train.ckpt_dir", "") == "" and c.get_int("dataset.scan_id", default=-1) == -1
    with pytest.raises(KeyError):
        c.get_int("train.missing")
    with pytest.raises(KeyError):
        c.get_config("nope")


def test_attr_view_and_to_plain():
    args = attr_view(dict(vol=dict(train=dict(x=1)), exps_folder="exps", use_mvs=False))
    assert args.exps_folder == "exps" and args["vol"]["train"]["x"] == 1 and args.vol.train.x == 1
    args.use_mvs = True
    assert args.use_mvs is True and "vol" in args
    with pytest.raises(AttributeError):
        args.nothing
    ns = types.SimpleNamespace(a=1, b=types.SimpleNamespace(c=[1, (2, 3)]))
    assert to_plain(ns) == {"a": 1, "b": {"c": [1, [2, 3]]}}
    assert to_plain(args)["vol"] == {"train": {"x": 1}}
    assert attr_view(ns) is ns


def test_grouped_step_results_are_lazy_mappings():
    """TrainStep returns lazy containers when a step ran as ray groups (trainer._GroupedLosses / _GroupedOutputs): loss
    terms are the sum over the groups, per-ray outputs the concatenation in ray order, both formed on first access."""
    import torch
    from svs_hip.trainer import _GroupedLosses, _GroupedOutputs
    res = [({"loss": torch.tensor(1.0), "rgb_loss": torch.tensor(0.5)}, {"rgb_values": torch.zeros(3, 3), "n": 4}),
           ({"loss": torch.tensor(2.0), "rgb_loss": torch.tensor(0.25)}, {"rgb_values": torch.ones(2, 3), "n": 4})]
    lo, out = _GroupedLosses(res), _GroupedOutputs(res)
    assert "loss" in lo and "nope" not in lo and len(lo) == 2 and list(lo) == ["loss", "rgb_loss"]
    assert float(lo["loss"]) == 3.0 and float(dict(lo.items())["rgb_loss"]) == 0.75
    assert lo.get("nope", 7) == 7 and [float(v) for v in lo.values()] == [3.0, 0.75]
    assert lo["loss"] is lo["loss"]                       # formed once
    assert out["rgb_values"].shape == (5, 3) and float(out["rgb_values"][3:].min()) == 1.0 and out["n"] == 4
