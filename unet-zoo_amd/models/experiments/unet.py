"""Vanilla U-Net - attribute surface of the reference's models/experiments/unet.py plus the fields its
harness needs but that file lacks (batch_size, iterations, image_size, data_loader)."""
from models.unet import Unet
from data.lidc_data import lidc_data

experiment_name = "Unet"
log_dir_name = "lidc"
data_loader = lidc_data
filter_channels = [32, 64, 128, 192]
latent_levels = len(filter_channels) - 1
n_classes = 2
no_convs_fcomb = 4
beta = 10.0
use_reversible = False
input_channels = 1
epochs_to_train = 50
batch_size = 32
iterations = 5000000
image_size = (1, 128, 128)
logging_frequency = 1000
validation_frequency = 1000
pretrained_model = None
model = Unet
