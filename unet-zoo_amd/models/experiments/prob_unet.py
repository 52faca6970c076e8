"""Probabilistic U-Net - same attribute surface as the reference's models/experiments/prob_unet.py.
(As in the reference, the harness does not forward `latent_dim`, so the model runs with its default 2.)"""
from models.probabilistic_unet import ProbabilisticUnet
from data.lidc_data import lidc_data
from utils import normalise_image

experiment_name = "ProbabilisticUnet"
log_dir_name = "lidc"
data_loader = lidc_data
filter_channels = [32, 64, 128, 192, 192, 192, 192]
latent_levels = 1
latent_dim = 6
iterations = 5000000
n_classes = 2
num_labels_per_subject = 4
no_convs_fcomb = 3
beta = 1.0
use_reversible = False
exponential_weighting = True
input_channels = 1
epochs_to_train = 20
batch_size = 12
image_size = (1, 128, 128)
augmentation_options = {"do_flip_lr": True, "do_flip_ud": True, "do_rotations": True, "do_scaleaug": True, "nlabels": n_classes}
input_normalisation = normalise_image
validation_samples = 16
num_validation_images = 100
logging_frequency = 1000
validation_frequency = 1000
weight_decay = 10e-5
pretrained_model = None
model = ProbabilisticUnet
