"""Dataset, graph / NIfTI I/O and projections for the MI355X HIP path."""
