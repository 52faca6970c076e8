"""Test suite: CPU tier (-m "not gpu") and GPU tier (-m gpu); see tests/conftest.py."""
