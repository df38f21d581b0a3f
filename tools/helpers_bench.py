"""Small shared helpers of the tools/ benchmarks."""
import numpy as np


def rand_words(seed, count, words=3):
    return np.random.default_rng(seed).integers(0, 2**63, size=(count, words), dtype=np.uint64)
