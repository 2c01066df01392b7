"""Hyper-parameter sets and the k-fold training helpers the CLIs use (MI355X build)."""
