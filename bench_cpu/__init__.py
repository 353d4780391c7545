"""CPU port of the self-defined commit stage (bench.py side figure only; not part of the product, not part of the oracle)."""
