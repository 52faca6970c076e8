"""``lidc_data(sys_config, exp_config)`` - drop-in for the reference's ``data/lidc_data.py:9-53``: ``.train`` / ``.validation`` /
``.test`` batch providers over the prepared LIDC arrays (training split augmented, several annotators per image), with
``.validation.images/labels`` and ``.test.images/labels`` exposed for the validation / test loops (train_model.py:164-168)."""
import numpy as np

from . import lidc_data_loader
from .batch_provider import BatchProvider


class lidc_data:
    def __init__(self, sys_config, exp_config):
        data = lidc_data_loader.load_and_maybe_process_data(input_file=sys_config.data_root,
                                                            preprocessing_folder=sys_config.preproc_folder, force_overwrite=False)
        self.data = data
        indices = {tt: np.arange(data[tt]["images"].shape[0]) for tt in data}
        if not hasattr(exp_config, "annotator_range"):
            exp_config.annotator_range = range(exp_config.num_labels_per_subject)
        common = dict(add_dummy_dimension=True, num_labels_per_subject=exp_config.num_labels_per_subject,
                      annotator_range=exp_config.annotator_range)
        self.train = BatchProvider(data["train"]["images"], data["train"]["labels"], indices["train"], do_augmentations=True,
                                   augmentation_options=exp_config.augmentation_options, **common)
        self.validation = BatchProvider(data["val"]["images"], data["val"]["labels"], indices["val"], **common)
        self.test = BatchProvider(data["test"]["images"], data["test"]["labels"], indices["test"], **common)
        self.test.images, self.test.labels = data["test"]["images"], data["test"]["labels"]
        self.validation.images, self.validation.labels = data["val"]["images"], data["val"]["labels"]
