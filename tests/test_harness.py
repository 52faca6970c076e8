"""Harness / config-surface tests: experiment modules (ours and, when present, the reference's own file)
load against the native package; the train-step contract runs on the GPU."""
import os
import types

import pytest
import torch

import unet_zoo_amd  # noqa: F401
from unet_zoo_amd import train_model as TM

EXP_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unet-zoo_amd", "models", "experiments")
REQUIRED = ["experiment_name", "log_dir_name", "filter_channels", "latent_levels", "n_classes", "no_convs_fcomb", "beta",
            "use_reversible", "input_channels", "model"]


@pytest.mark.parametrize("name", ["phiseg_7_5_12", "phiseg_7_5_32", "prob_unet", "unet"])
def test_experiment_modules_load(name):
    m = TM.load_experiment(os.path.join(EXP_DIR, name + ".py"))
    for attr in REQUIRED:
        assert hasattr(m, attr), attr
    assert m.model.__module__.startswith("unet_zoo_amd.models")


@pytest.mark.skipif(not os.path.exists("/root/reference/models/experiments/phiseg_7_5_12.py"), reason="reference tree not present")
def test_reference_experiment_file_loads_unchanged():
    m = TM.load_experiment("/root/reference/models/experiments/phiseg_7_5_12.py")
    assert m.model.__name__ == "PHISeg" and m.model.__module__ == "unet_zoo_amd.models.phiseg"
    assert m.filter_channels == [32, 64, 128, 192, 192, 192, 192] and m.batch_size == 12


def _small_cfg(model, filters):
    cfg = types.SimpleNamespace(experiment_name="t", log_dir_name="t", filter_channels=filters, latent_levels=5, n_classes=2,
                                no_convs_fcomb=3, beta=1.0, use_reversible=False, input_channels=1, image_size=(1, 128, 128),
                                batch_size=4, iterations=4, logging_frequency=2, model=model)
    return cfg


def test_harness_builds_on_cpu_structure_only():
    from unet_zoo_amd.models.phiseg import PHISeg
    if torch.cuda.is_available():
        pytest.skip("CPU-tier check")
    h = TM.UNetModel(_small_cfg(PHISeg, [4, 8, 8, 8, 8, 8, 8]))
    assert h.optimizer.param_groups[0]["lr"] == 1e-3 and h.optimizer.param_groups[0]["weight_decay"] == 1e-5
    assert h.scheduler.patience == 50000 and h.scheduler.min_lrs == [1e-4]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["phiseg", "probunet", "unet"])
def test_train_step_contract_runs(which, tmp_path):
    from unet_zoo_amd.models import PHISeg, ProbabilisticUnet, Unet
    model, filters = {"phiseg": (PHISeg, [8, 16, 16, 16, 16, 16, 16]), "probunet": (ProbabilisticUnet, [32, 8, 8, 8, 8, 8, 8]),
                      "unet": (Unet, [8, 16, 16, 16])}[which]
    h = TM.UNetModel(_small_cfg(model, filters), log_root=str(tmp_path))
    data = TM.SyntheticData(None, h.exp_config, n_train=16, n_val=4)
    before = {k: v.detach().clone() for k, v in h.net.state_dict().items() if v.dtype.is_floating_point}
    h.train(data)                                   # 3 iterations (range(1, iterations))
    assert torch.isfinite(h.loss)
    after = h.net.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)
    path = h.save_model("last")
    sd = torch.load(path)
    assert list(sd.keys()) == list(after.keys())
    h.net.load_state_dict(sd)
    if which in ("phiseg", "probunet"):
        h.exp_config.validation_samples, h.exp_config.num_validation_images = 4, 2
        m = h.validate(data)
        assert 0.0 <= m["dice"] <= 1.0 and -1.0 <= m["ncc"] <= 1.0 and m["ged"] >= -1e-9
        assert os.path.exists(os.path.join(str(tmp_path), "t", "t", "t_best_ged.pth"))
