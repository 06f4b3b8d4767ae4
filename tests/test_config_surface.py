"""The reference's config surface, consumed unchanged (train.py:30-34,70; utils.py:175-176; configs/*.yml).

tests/golden/config_surface.json was written by oracle/pin_against_reference.py from the REAL reference: every shipped yaml
parsed the way train.py does, the reference's own model built from it through the module-name hook, and the resulting
state_dict inventory.  Here the same yaml dictionaries go through the same hook expression into the MI355X build (the
drop-in shim of INTEGRATION.md, committed as integration/models/M2Trans_network.py)."""
import importlib
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def surface(golden_dir):
    return json.load(open(os.path.join(golden_dir, "config_surface.json")))


@pytest.fixture()
def shim_on_path(monkeypatch):
    """`models` must resolve to integration/models (what a reference checkout with the shim dropped in looks like)."""
    for name in [n for n in sys.modules if n == "models" or n.startswith("models.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(os.path.join(ROOT, "integration"))
    yield
    for name in [n for n in sys.modules if n == "models" or n.startswith("models.")]:
        sys.modules.pop(name, None)


def import_module(name):                    # utils.py:175-176, verbatim behaviour
    return importlib.import_module(name)


def _namespace(cfg, path):
    """train.py:28-34: the argparse namespace (--config, --resume) updated with the yaml dictionary."""
    args = types.SimpleNamespace(config=path, resume=None)
    opt = vars(args)
    opt.update(cfg)
    return args


def test_six_shipped_configs_are_pinned(surface):
    assert sorted(surface) == ["M2Trans_x2.yml", "M2Trans_x2_test.yml", "M2Trans_x3.yml", "M2Trans_x3_test.yml",
                               "M2Trans_x4.yml", "M2Trans_x4_test.yml"]
    for name, rec in surface.items():
        y = rec["yaml"]
        assert y["model"] == "M2Trans" and y["n_feats"] == 64 and y["colors"] == 3 and y["n_blocks"] == 8, name
        assert y["scale"] == int(name.split("_x")[1][0]), name
        assert "num_heads" in y            # present in every yaml, read by nothing (models/M2Trans_network.py:281: heads = 1)


@pytest.mark.parametrize("name", ["M2Trans_x2.yml", "M2Trans_x2_test.yml", "M2Trans_x3.yml", "M2Trans_x3_test.yml",
                                  "M2Trans_x4.yml", "M2Trans_x4_test.yml"])
def test_model_from_every_shipped_yaml_through_the_module_name_hook(surface, shim_on_path, name):
    rec = surface[name]
    args = _namespace(rec["yaml"], "configs/" + name)
    torch.manual_seed(33)
    model = import_module("models.{}_network".format(args.model)).create_model(args)      # the exact expression of train.py:70
    mod = sys.modules["models.M2Trans_network"]
    assert os.path.realpath(mod.__file__).startswith(os.path.realpath(os.path.join(ROOT, "integration")))
    from m2trans_amd.M2Trans_network import M2Trans
    assert isinstance(model, M2Trans) and mod.M2Trans is M2Trans          # test.py:16 imports the class by name
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]
    assert got == rec["state_dict"]                                       # names, shapes, dtypes AND order of the reference
    assert sum(p.numel() for p in model.parameters()) == rec["n_params"]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == rec["n_trainable"]
    assert model.scale == rec["yaml"]["scale"] and model.n_blocks == 8 and model.rgb_range == 1.0
    # keys the reference's model ignores are ignored here too: the model is the same with them changed
    other = dict(rec["yaml"], num_heads=1, window_sizes=[4, 8], batch_size=7, gpu_ids=[0, 1])
    torch.manual_seed(33)
    m2 = import_module("models.{}_network".format(args.model)).create_model(_namespace(other, "x"))
    assert [k for k in m2.state_dict()] == [k for k, _, _ in rec["state_dict"]]
    assert torch.equal(m2.flat_params, model.flat_params)                 # same seed, same draws: nothing else consumed RNG
    # the oracle's inventory is the same one
    from oracle import m2trans_oracle as O
    p = O.closed_form_params(64, rec["yaml"]["scale"], 8)
    assert [[k, list(v.shape)] for k, v in p.items()] == [[k, s] for k, s, _ in rec["state_dict"]]


def test_plan_cache_is_a_small_lru(monkeypatch):
    """metrics.evaluate over images of arbitrary sizes must not keep one workspace per distinct shape forever."""
    from m2trans_amd import M2Trans_network as N

    class FakePlan:
        made = []

        def __init__(self, B, H0, W0, scale, n_blocks, dtype, device):
            self.key = (B, H0, W0)
            FakePlan.made.append(self.key)

        def query(self, key):
            raise KeyError(key)

    monkeypatch.setattr(N, "Plan", FakePlan)
    monkeypatch.setattr(N.M2Trans, "_check_plan", lambda self, plan: None)
    m = N.M2Trans(types.SimpleNamespace(n_feats=64, scale=4, rgb_range=1.0, n_blocks=1, colors=3))

    class X:                                   # shape-only stand-in for a device tensor
        is_cuda = True

        def __init__(self, B, H, W, dev):
            self.shape = (B, 3, H, W)
            self.device = dev

        def dim(self):
            return 4

    dev = m.flat_params.device
    monkeypatch.setattr(N.M2Trans, "_device_ok", lambda self, x: None)
    cap = N.PLAN_CACHE_SIZE
    assert cap >= 2
    for i in range(cap + 3):
        m._plan_for(X(1, 32 + i, 32, dev))
    assert len(m._plans) == cap
    first_alive = (1, 32 + 3, 32)
    assert first_alive in [p.key for p in m._plans.values()]
    m._plan_for(X(1, 32 + 3, 32, dev))         # a hit refreshes the entry ...
    n_made = len(FakePlan.made)
    m._plan_for(X(1, 99, 32, dev))             # ... so the next miss evicts the second-oldest, not it
    assert first_alive in [p.key for p in m._plans.values()]
    assert (1, 32 + 4, 32) not in [p.key for p in m._plans.values()]
    assert len(FakePlan.made) == n_made + 1 and len(m._plans) == cap


def test_plan_cache_keeps_the_training_plan_through_a_validation_sweep(monkeypatch):
    """A validation pass over more than PLAN_CACHE_SIZE distinct image sizes (test.py:77-122 between epochs) must not evict the plan
    the train step runs on (rebuilding it = a workspace allocation, the descriptor uploads and a first-backward pass every epoch):
    plans that have run a backward go only when no forward-only plan is left."""
    from m2trans_amd import M2Trans_network as N

    class FakePlan:
        def __init__(self, B, H0, W0, scale, n_blocks, dtype, device):
            self.key = (B, H0, W0)
            self.trained = False

    monkeypatch.setattr(N, "Plan", FakePlan)
    monkeypatch.setattr(N.M2Trans, "_check_plan", lambda self, plan: None)
    monkeypatch.setattr(N.M2Trans, "_device_ok", lambda self, x: None)
    m = N.M2Trans(types.SimpleNamespace(n_feats=64, scale=4, rgb_range=1.0, n_blocks=1, colors=3))

    class X:
        is_cuda = True

        def __init__(self, B, H, W, dev):
            self.shape = (B, 3, H, W)
            self.device = dev

        def dim(self):
            return 4

    dev = m.flat_params.device
    train = m._plan_for(X(16, 128, 128, dev))
    train.trained = True                                    # what TrainStep.forward_backward / the autograd node set
    for i in range(3 * N.PLAN_CACHE_SIZE):
        m._plan_for(X(1, 33 + i, 47, dev))
    assert m._plan_for(X(16, 128, 128, dev)) is train and len(m._plans) == N.PLAN_CACHE_SIZE
    # when every cached plan has trained, the least recently used one goes after all
    for pl in m._plans.values():
        pl.trained = True
    m._plan_for(X(2, 64, 64, dev))
    assert len(m._plans) == N.PLAN_CACHE_SIZE
