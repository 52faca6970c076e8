"""Input pipeline of the training harness: LIDC preparation + mini-batch provider, with the batch assembly and
augmentation running on the GPU over an HBM-resident dataset (data/batch_provider.py, data/lidc_data_loader.py, data/lidc_data.py
of the reference)."""
from .batch_provider import BatchProvider  # noqa: F401
from .lidc_data import lidc_data  # noqa: F401
