"""CPU oracle — test infrastructure only (see retake_oracle.c). Never imported by the product."""
