"""PHiSeg 7 resolution / 5 latent levels, batch 32 (BASELINE config 4) - same attribute surface as the reference's
models/experiments/phiseg_7_5_12.py (every name the reference harness reads is present)."""
from models.phiseg import PHISeg
from data.lidc_data import lidc_data
from utils import normalise_image

experiment_name = "PHISeg_7_5_32"
log_dir_name = "lidc"
data_loader = lidc_data

filter_channels = [32, 64, 128, 192, 192, 192, 192]
latent_levels = 5
iterations = 5000000
n_classes = 2
num_labels_per_subject = 4
no_convs_fcomb = 4          # not used by PHISeg
beta = 10.0                 # not used by PHISeg
use_reversible = False
exponential_weighting = True
input_channels = 1
epochs_to_train = 20
batch_size = 32
image_size = (1, 128, 128)
augmentation_options = {"do_flip_lr": True, "do_flip_ud": True, "do_rotations": True, "do_scaleaug": True, "nlabels": n_classes}
input_normalisation = normalise_image
validation_samples = 16
num_validation_images = 100
logging_frequency = 1000
validation_frequency = 1000
weight_decay = 10e-5
pretrained_model = None
model = PHISeg
