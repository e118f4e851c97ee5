"""Synthetic-weight model graphs of BASELINE.json's configs, built from rten_amd.ops."""
