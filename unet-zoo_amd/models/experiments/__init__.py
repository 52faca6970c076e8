"""Experiment files of the native package (same attribute surface as the reference experiment modules)."""
