"""Drop-in counterpart of the reference harness ``train_model.py`` for the native models.

Scope: the TRAIN-STEP CONTRACT of ``UNetModel`` - build
``exp_config.model(**8 kwargs)`` (train_model.py:34-42), Adam(lr=1e-3, weight_decay=1e-5) (:49),
ReduceLROnPlateau('min', min_lr=1e-4, patience=50000) stepped on the loss (:50-51,:134), and per
iteration: to-device, ``forward(patch, mask, training=True)``, ``loss(mask)``, ``zero_grad``,
``backward``, ``step`` (:100-134) - plus ``save_model`` (:558-564).  Validation metrics (GED / NCC / Dice via medpy),
the device-side validation and test loops (N prior samples, GED / NCC / Dice) are `validate` / `test`
(train_model.py:138-275,333-475); image dumps and the UZH / BraTS loaders stay outside the hot path.

Experiment files of the reference are plain Python modules that import ``models.phiseg``,
``data.lidc_data`` and ``utils`` (e.g. models/experiments/phiseg_7_5_12.py:1-5).  ``load_experiment``
executes such a file unchanged with those names aliased to the native package, so
``python -m unet_zoo_amd.train_model EXP_PATH LOCAL dummy`` keeps the reference's CLI (:568-572).
With torchrun (WORLD_SIZE > 1) every rank trains on its own shard and gradients are averaged over
RCCL inside ``loss.backward()``.
"""
import argparse
import importlib.util
import logging
import os
import sys
import types

import numpy as np
import torch

from . import dp
from .optim import FusedAdam
from .synthetic import synthetic_batch


# --------------------------------------------------------------------------- data stand-in
class _Split:
    def __init__(self, images, labels, seed=0):
        self.images, self.labels = images, labels
        self._rng = np.random.default_rng(seed)

    def reseed(self, seed):
        self._rng = np.random.default_rng(seed)

    def next_batch(self, batch_size):
        """(x (B,1,H,W) float32 in [-0.5,0.5], s (B,H,W) labels) like BatchProvider.next_batch (batch_provider.py:43-67)."""
        idx = self._rng.choice(len(self.images), size=batch_size, replace=len(self.images) < batch_size)
        return self.images[idx], self.labels[idx]


class SyntheticData:
    """Stand-in for ``exp_config.data_loader(sys_config, exp_config)`` when LIDC-IDRI is not on disk:
    same surface (``.train.next_batch``, ``.validation.images/labels``), synthetic LIDC-like content."""

    def __init__(self, sys_config=None, exp_config=None, n_train=256, n_val=32):
        size = getattr(exp_config, "image_size", (1, 128, 128))
        h, w = size[1], size[2]
        x, m, _ = synthetic_batch(n_train + n_val, h, w, seed=7)
        self.train = _Split(x[:n_train], m[:n_train, 0])
        if torch.cuda.is_available():
            # the native input pipeline (HBM-resident split, batch assembly + augmentation in one HIP kernel) also serves the
            # synthetic stand-in, so the training loop exercises it without the LIDC files
            from .data.batch_provider import BatchProvider
            n_ann_t = getattr(exp_config, "num_labels_per_subject", 4)
            lab = np.stack([np.roll(m[:n_train, 0], shift=(a, -a), axis=(1, 2)) for a in range(n_ann_t)], axis=-1).astype(np.uint8)
            self.train = BatchProvider(x[:n_train, 0], lab, np.arange(n_train), add_dummy_dimension=True, do_augmentations=True,
                                       augmentation_options=getattr(exp_config, "augmentation_options", None) or
                                       dict(do_rotations=True, do_scaleaug=True, nlabels=getattr(exp_config, "n_classes", 2)),
                                       num_labels_per_subject=n_ann_t, annotator_range=range(n_ann_t))
        # validation labels carry several annotators per image, (n, H, W, A), like the LIDC HDF5 (lidc_data_loader.py:92-110)
        n_ann = getattr(exp_config, "num_labels_per_subject", 4)
        ann = [m[n_train:, 0]]
        for a in range(1, n_ann):
            ann.append(np.roll(m[n_train:, 0], shift=(a, -a), axis=(1, 2)))
        self.validation = _Split(x[n_train:, 0], np.stack(ann, axis=-1))
        self.test = self.validation
        if exp_config is not None and not hasattr(exp_config, "annotator_range"):
            exp_config.annotator_range = range(n_ann)            # injected by the loader in the reference (lidc_data.py:31-32)


def lidc_data(sys_config=None, exp_config=None):
    """The reference's `data.lidc_data.lidc_data`: the real LIDC loader when `sys_config.data_root` points at the pickle, the
    synthetic stand-in otherwise (no dataset ships with this repository)."""
    root = getattr(sys_config, "data_root", None)
    if root and os.path.exists(root):
        from .data.lidc_data import lidc_data as real
        return real(sys_config, exp_config)
    logging.getLogger("unet_zoo_amd").warning(
        "!!!! LIDC data not found (sys_config.data_root = %r): training on the SYNTHETIC stand-in data set. "
        "Pass --data-root / --preproc-folder (or a sys_config module via LOCAL) to train on LIDC-IDRI.", root)
    return SyntheticData(sys_config, exp_config)


def normalise_image(image):
    """utils.normalise_image (utils.py:104-115): zero mean, unit standard deviation."""
    img = np.float32(image.copy())
    return (img - img.mean()) / (img.std() + 1e-20)


def load_experiment(path):
    """Execute a reference-style experiment module (train_model.py:584) against the native package."""
    from . import models as native_models
    alias = {"models": native_models, "models.phiseg": native_models.phiseg, "models.unet": native_models.unet,
             "models.probabilistic_unet": native_models.probabilistic_unet, "models.phiseg3D": native_models.phiseg3D}
    data_pkg, lidc_mod, utils_mod = types.ModuleType("data"), types.ModuleType("data.lidc_data"), types.ModuleType("utils")
    uzh_mod = types.ModuleType("data.uzh_prostate_data")            # private UZH prostate set: same loader surface, stand-in content
    lidc_mod.lidc_data = lidc_data
    uzh_mod.uzh_prostate_data = lidc_data
    data_pkg.lidc_data, data_pkg.uzh_prostate_data = lidc_mod, uzh_mod
    utils_mod.normalise_image = normalise_image
    alias.update({"data": data_pkg, "data.lidc_data": lidc_mod, "data.uzh_prostate_data": uzh_mod, "utils": utils_mod})
    saved = {k: sys.modules.get(k) for k in alias}
    sys.modules.update(alias)
    try:
        spec = importlib.util.spec_from_file_location(os.path.splitext(os.path.basename(path))[0], path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


# --------------------------------------------------------------------------- harness
class UNetModel:
    """Wrapper that trains a native model exactly as the reference harness does (train_model.py:27-136)."""

    def __init__(self, exp_config, logger=None, tensorboard=False, log_root="./logs"):
        # one process per GPU: bind this rank to cuda:LOCAL_RANK *before* any buffer is allocated (the model, the
        # optimiser state and the RCCL communicator must all live on the same device)
        # Control plane (unique-id hand-off, barriers, the scheduler's mean loss) = a gloo group on the host; the ONE RCCL
        # communicator of the process is opened through the C ABI by dp.GradSync (uz_comm_init).
        self.rank, self.local_rank, self.world = dp.init_from_env(backend="gloo")
        if self.world > 1 and torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
            torch.cuda.set_stream(torch.cuda.Stream())      # overlapped gradient exchange needs a created (non-default) stream
        kwargs = dict(input_channels=exp_config.input_channels, num_classes=exp_config.n_classes,
                      num_filters=exp_config.filter_channels, latent_levels=exp_config.latent_levels,
                      no_convs_fcomb=exp_config.no_convs_fcomb, beta=exp_config.beta,
                      image_size=getattr(exp_config, "image_size", (1, 128, 128)), reversible=exp_config.use_reversible)
        try:
            self.net = exp_config.model(**kwargs)
        except TypeError:
            # the reference's own Unet rejects latent_levels / image_size (unet.py:88-90) and cannot be driven by
            # its harness; here the two unsupported keywords are dropped instead
            kwargs.pop("latent_levels"), kwargs.pop("image_size")
            self.net = exp_config.model(**kwargs)
        self.exp_config, self.batch_size = exp_config, getattr(exp_config, "batch_size", 12)
        self.logger = logger or logging.getLogger("unet_zoo_amd")
        self.device = self.net.device
        self.optimizer = FusedAdam(self.net, lr=1e-3, weight_decay=1e-5)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, "min", min_lr=1e-4, patience=50000)
        if self.world > 1:
            self.net.set_data_parallel(True, backend="rccl" if torch.cuda.is_available() and os.environ.get("UZ_DP_BACKEND", "rccl") == "rccl" else "torch")
            self.net._dp.broadcast_params()
        # static shapes: replay the tapes' dependency DAGs on the engine's lanes (NativeModel.enable_graphs; bit-identical to the
        # one-stream tape) and, once the plan exists, let the engine measure its own schedule (tune_schedule; UZ_TUNE_SCHEDULE rounds, 0 = off)
        if torch.cuda.is_available() and hasattr(self.net, "enable_graphs") and os.environ.get("UZ_TRAIN_REPLAY", "1") != "0":
            self.net.enable_graphs(True)
        self._tune_rounds = int(os.environ.get("UZ_TUNE_SCHEDULE", "8"))
        self.log_root = log_root
        self.tot_loss = self.kl_loss = self.reconstruction_loss = 0
        self.iteration = 0

    def train_step(self, x_b, s_b):
        patch = torch.as_tensor(x_b, dtype=torch.float32).to(self.device)          # no-ops for the device tensors of the native provider
        mask = torch.unsqueeze(torch.as_tensor(s_b, dtype=torch.float32).to(self.device), 1)
        self.mask, self.patch = mask, patch
        guarded = hasattr(self.net, "guard_bounds")
        if guarded:
            self.net.snapshot_step()
        self.net.forward(patch, mask, training=True)
        self.loss = self.net.loss(mask)
        self.optimizer.zero_grad()
        self.loss.backward()
        if guarded and self.net.guard_bounds():
            # a split-fp16 kernel clamped a tensor (never expected: the bounds are maintained by the producing kernels) - the
            # engine has switched to fp32 MFMA arithmetic on EVERY rank (the decision is collective); repeat this batch there
            # before the parameters are touched: same eps, BatchNorm buffers and counters rewound, so it is the same step
            self.net.restore_step()
            self.net.forward(patch, mask, training=True)
            self.loss = self.net.loss(mask)
            self.optimizer.zero_grad()
            self.loss.backward()
        self.tot_loss += self.loss.detach()
        self.reconstruction_loss += getattr(self.net, "reconstruction_loss", 0)
        self.kl_loss += getattr(self.net, "kl_divergence_loss", 0)
        self.optimizer.step()
        return self.loss

    def tune_schedule(self):
        """Profile-guided lane schedule on the batch of the last train_step (NativeModel.tune_schedule): every tuning pass starts from
        the same BatchNorm buffers, counters and noise-stream position (snapshot_step / restore_step) and leaves them as it found
        them; no optimiser step runs; gradients are overwritten by the next step anyway.  Collective under data parallelism."""
        net = self.net
        if self._tune_rounds <= 0 or not getattr(net, "_use_graphs", False) or getattr(net, "replay_mode", "") != "lanes" or not hasattr(net, "snapshot_step"):
            return None
        net.snapshot_step()

        def step():
            net.restore_step()
            net.forward(self.patch, self.mask, training=True)
            loss = net.loss(self.mask)
            self.optimizer.zero_grad()
            loss.backward()
        res = net.tune_schedule(step, rounds=self._tune_rounds)
        net.restore_step()
        if self.rank == 0:
            self.logger.info("lane schedule tuned: %s", res)
        return res

    def train(self, data, iterations=None):
        self.net.train()
        n_it = iterations if iterations is not None else self.exp_config.iterations
        log_every = getattr(self.exp_config, "logging_frequency", 1000)
        if self.world > 1:                                       # every rank draws its own shard of each global batch
            if hasattr(data.train, "reseed"):
                data.train.reseed(1000003 * self.rank + 17)
            np.random.seed(1000003 * self.rank + 17)             # the native BatchProvider samples from numpy's global RNG, like the reference
        for self.iteration in range(1, n_it):
            x_b, s_b = data.train.next_batch(self.batch_size)
            loss = self.train_step(x_b, s_b)
            if self.iteration == 3:                              # the plan exists and is warm: measure its schedule once
                self.tune_schedule()
            val_every = getattr(self.exp_config, "validation_frequency", 0)
            if val_every and self.iteration % val_every == 0:
                self.validate(data)
            if self.iteration % log_every == 0:
                self.logger.info("Iteration {} Loss {}".format(self.iteration, float(loss)))
                self.tot_loss = self.kl_loss = self.reconstruction_loss = 0
            # train_model.py:134 (compares on the host: one sync per step); under data parallelism every rank steps the
            # scheduler on the SAME number - the mean of the ranks' losses - so the learning rates cannot diverge
            self.scheduler.step(dp.mean_scalar(loss.detach()))
        self.logger.info("Finished training.")

    @torch.no_grad()
    def _evaluate_image(self, image, s_gt_arr, n_samples, rng, keep=False):
        """One image of the validation / test loops (train_model.py:166-230 and :369-425): `n_samples` copies through
        forward(training=False) -> accumulate_output(softmax) -> [loss] -> argmax -> GED / NCC / per-label Dice on the
        device.  s_gt_arr: (H, W, A) annotator label maps.  Returns a dict; keep=True also returns the drawn samples."""
        from . import metrics
        cfg, n_classes = self.exp_config, self.exp_config.n_classes
        ann_range = list(getattr(cfg, "annotator_range", range(s_gt_arr.shape[-1])))
        patch = torch.as_tensor(image, dtype=torch.float32).to(self.device)
        H, W = patch.shape[-2:]
        val_patch = patch.reshape(1, 1, H, W)
        s_b = s_gt_arr[:, :, rng.choice(ann_range)]
        val_mask = torch.as_tensor(s_b, dtype=torch.float32).to(self.device).reshape(1, 1, H, W)
        val_masks = torch.as_tensor(s_gt_arr, dtype=torch.float32).to(self.device).permute(2, 0, 1).contiguous()      # (A, H, W)
        patch_arr = val_patch.repeat(n_samples, 1, 1, 1)
        mask_arr = val_mask.repeat(n_samples, 1, 1, 1)
        out = self.net.forward(patch_arr, mask_arr, training=False)
        # PHISeg: list of level logits, summed in place + softmax.  ProbabilisticUnet: the reference feeds the forward()
        # output itself - last_conv(unet_features), probabilistic_unet.py:255 - through its pass-through
        # accumulate_output (:285-290), so its N "samples" are identical; reproduced as is (train_model.py:185-189).
        soft = self.net.accumulate_output(out, use_softmax=True)
        elbo = float(self.net.loss(mask_arr))
        pred = torch.argmax(soft, dim=1)                                        # (N, H, W)
        ged = metrics.generalised_energy_distance(pred, val_masks.long(), nlabels=n_classes - 1, label_range=range(1, n_classes))
        onehot = torch.stack([(val_masks == k) for k in range(n_classes)], dim=1).long()            # (A, K, H, W)
        ncc = metrics.variance_ncc_dist(soft, onehot)
        s_mean = torch.argmax(torch.mean(soft, dim=0), dim=0)
        dice = metrics.per_label_dice(s_mean, val_mask.reshape(H, W).long(), n_classes)
        res = dict(ged=ged, ncc=ncc, dice=dice, elbo=elbo)
        if keep:
            res.update(soft=soft, pred=pred, gts=val_masks, mask=val_mask.reshape(H, W), onehot=onehot)
        return res

    @torch.no_grad()
    def validate(self, data):
        """UNetModel.validate (train_model.py:138-275) on the device: per validation image, `validation_samples`
        prior samples in eval mode -> accumulate_output(softmax) -> loss -> argmax -> GED / NCC / Dice
        (unet_zoo_amd.metrics), then the running means and the best-model checkpoints."""
        cfg = self.exp_config
        self.net.eval()
        self.save_model("validation_ckpt")
        n_samples = getattr(cfg, "validation_samples", 16)
        n_classes = cfg.n_classes
        n_img = data.validation.images.shape[0] if getattr(cfg, "num_validation_images", "all") == "all" \
            else min(cfg.num_validation_images, data.validation.images.shape[0])
        rng = np.random.default_rng(self.iteration)
        ged_l, dice_l, ncc_l, elbo_l = [], [], [], []
        for ii in range(n_img):
            r = self._evaluate_image(data.validation.images[ii], data.validation.labels[ii], n_samples, rng)
            ged_l.append(r["ged"]); ncc_l.append(r["ncc"]); dice_l.append(r["dice"]); elbo_l.append(r["elbo"])
        dice = torch.tensor(dice_l)
        self.avg_dice = float(dice.mean())
        self.foreground_dice = float(dice.mean(dim=0)[1]) if n_classes > 1 else float("nan")
        self.val_elbo = float(np.nanmean(elbo_l)) if len(elbo_l) and not all(np.isnan(elbo_l)) else float("nan")
        self.avg_ged, self.avg_ncc = float(np.mean(ged_l)), float(np.mean(ncc_l))
        self.logger.info(" - Foreground dice: %.4f" % self.foreground_dice)
        self.logger.info(" - Mean (neg.) ELBO: %.4f" % self.val_elbo)
        self.logger.info(" - Mean GED: %.4f" % self.avg_ged)
        self.logger.info(" - Mean NCC: %.4f" % self.avg_ncc)
        best = self.__dict__.setdefault("_best", dict(dice=-1.0, loss=float("inf"), ged=float("inf"), ncc=-1.0))
        if float(dice.mean(dim=0).mean()) >= best["dice"]:
            best["dice"] = float(dice.mean(dim=0).mean()); self.save_model("best_dice")
        if self.val_elbo <= best["loss"]:
            best["loss"] = self.val_elbo; self.save_model("best_loss")
        if self.avg_ged <= best["ged"]:
            best["ged"] = self.avg_ged; self.save_model("best_ged")
        if self.avg_ncc >= best["ncc"]:
            best["ncc"] = self.avg_ncc; self.save_model("best_ncc")
        self.net.train()
        return dict(dice=self.avg_dice, foreground_dice=self.foreground_dice, elbo=self.val_elbo, ged=self.avg_ged, ncc=self.avg_ncc)

    @torch.no_grad()
    def test(self, data, sys_config=None, rounds=10, n_samples=10):
        """UNetModel.test (train_model.py:333-475): load `<experiment>_best_loss.pth`, then `rounds` passes over the
        test split with `n_samples` prior samples per image; GED / NCC arrays are dumped as
        `ged<n>_<model>_2.npz` / `ncc<n>_<model>_2.npz` next to the checkpoint (:446-447); returns the mean
        Dice / GED / NCC over the rounds (:473-475 logs them)."""
        cfg = self.exp_config
        self.net.eval()
        model_selection = cfg.experiment_name + "_best_loss.pth"
        root = getattr(sys_config, "log_root", None) or self.log_root
        model_dir = os.path.join(root, cfg.log_dir_name, cfg.experiment_name)
        model_path = os.path.join(model_dir, model_selection)
        self.logger.info("Testing {}".format(model_selection))
        if not os.path.exists(model_path):
            self.logger.info("The file {} does not exist. Aborting test function.".format(model_path))
            return None
        self.net.load_state_dict(torch.load(model_path))
        rng = np.random.default_rng(0)
        ged_l, dice_l, ncc_l = [], [], []
        end = dict(dice=0.0, ged=0.0, ncc=0.0)
        for rnd in range(rounds):
            self.logger.info("Doing iteration {}".format(rnd))
            for ii in range(data.test.images.shape[0]):
                r = self._evaluate_image(data.test.images[ii], data.test.labels[ii], n_samples, rng)
                ged_l.append(r["ged"]); ncc_l.append(r["ncc"]); dice_l.append(r["dice"])
            dice = torch.tensor(dice_l)
            self.avg_dice = float(dice.mean())
            self.foreground_dice = float(dice.mean(dim=0)[1]) if cfg.n_classes > 1 else float("nan")
            self.avg_ged, self.avg_ncc = float(np.mean(ged_l)), float(np.mean(ncc_l))       # lists grow over the rounds, as in the reference
            if self.rank == 0:
                np.savez(os.path.join(model_dir, "ged%s_%s_2.npz" % (str(n_samples), model_selection)), np.asarray(ged_l))
                np.savez(os.path.join(model_dir, "ncc%s_%s_2.npz" % (str(n_samples), model_selection)), np.asarray(ncc_l))
            self.logger.info(" - Foreground dice: %.4f" % self.foreground_dice)
            self.logger.info(" - Mean GED: %.4f" % self.avg_ged)
            self.logger.info(" - Mean NCC: %.4f" % self.avg_ncc)
            end["dice"] += self.avg_dice; end["ged"] += self.avg_ged; end["ncc"] += self.avg_ncc
        out = {k: v / rounds for k, v in end.items()}
        self.logger.info("Mean dice: {}".format(out["dice"]))
        self.logger.info("Mean ged: {}".format(out["ged"]))
        self.logger.info("Mean ncc: {}".format(out["ncc"]))
        return out

    def save_model(self, savename):
        """<log_root>/<log_dir_name>/<experiment_name>/<experiment_name>_<savename>.pth (train_model.py:558-564)."""
        name = self.exp_config.experiment_name + "_" + savename + ".pth"
        log_dir = os.path.join(self.log_root, self.exp_config.log_dir_name, self.exp_config.experiment_name)
        os.makedirs(log_dir, exist_ok=True)
        path = os.path.join(log_dir, name)
        if self.rank == 0:
            torch.save({k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()}, path)
        return path


def resolve_sys_config(local, data_root=None, preproc_folder=None, log_root=None):
    """The reference picks `config.local_config` or `config.system` from the LOCAL argument (train_model.py:577-581) and hands
    the module to the data loader and to test().  Here LOCAL may be 'local' / anything else (-> `config.local_config` /
    `config.system` when such a package is importable, as in a checkout of the reference) or the path of a Python file with
    the same attributes; --data-root / --preproc-folder / --log-root (or UZ_DATA_ROOT / UZ_PREPROC_FOLDER / UZ_LOG_ROOT)
    override single attributes.  Missing attributes default to None / './logs'."""
    cfg = types.SimpleNamespace(data_root=None, preproc_folder=None, log_root="./logs")
    mod = None
    if local and os.path.isfile(local):
        spec = importlib.util.spec_from_file_location("uz_sys_config", local)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    else:
        try:
            mod = importlib.import_module("config.local_config" if local == "local" else "config.system")
        except ImportError:
            mod = None
    if mod is not None:
        for k, v in vars(mod).items():
            if not k.startswith("_"):
                setattr(cfg, k, v)
    for attr, val, env in (("data_root", data_root, "UZ_DATA_ROOT"), ("preproc_folder", preproc_folder, "UZ_PREPROC_FOLDER"),
                           ("log_root", log_root, "UZ_LOG_ROOT")):
        val = val or os.environ.get(env)
        if val:
            setattr(cfg, attr, val)
    if cfg.data_root and not cfg.preproc_folder:
        cfg.preproc_folder = os.path.join(os.path.dirname(os.path.abspath(cfg.data_root)), "preproc")
    return cfg


def main(argv=None):
    ap = argparse.ArgumentParser(description="Script for training (native MI355X path)")
    ap.add_argument("EXP_PATH", type=str, help="Path to experiment config file")
    ap.add_argument("LOCAL", type=str, help="'local' / cluster name as in the reference (imports config.local_config / config.system when "
                                            "present), or the path of a sys_config Python file")
    ap.add_argument("dummy", type=str, nargs="?", default="dummy", help="(unused, kept for CLI compatibility)")
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--data-root", default=None, help="sys_config.data_root: the LIDC pickle (data_lidc.pickle)")
    ap.add_argument("--preproc-folder", default=None, help="sys_config.preproc_folder: where the prepared splits are cached")
    ap.add_argument("--log-root", default=None, help="sys_config.log_root: checkpoints land in <log_root>/<log_dir_name>/<experiment_name>")
    ap.add_argument("--test", action="store_true", help="run UNetModel.test on the best-loss checkpoint after training (train_model.py:333-475)")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    sys_config = resolve_sys_config(args.LOCAL, args.data_root, args.preproc_folder, args.log_root)
    exp_config = load_experiment(args.EXP_PATH)
    model = UNetModel(exp_config, log_root=sys_config.log_root)
    data = exp_config.data_loader(sys_config=sys_config, exp_config=exp_config) if hasattr(exp_config, "data_loader") else lidc_data(sys_config, exp_config)
    model.train(data, iterations=args.iterations)
    print(model.save_model("last"))
    if args.test:
        model.test(data, sys_config=sys_config)


if __name__ == "__main__":
    main()
