"""Seeded synthetic workloads of the benchmark: inputs and weights only -- neither product code nor checker (oracle/)."""
