"""Networks, training harness and metrics on the MI355X HIP path."""
