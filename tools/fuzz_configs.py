#!/usr/bin/env python3
"""Seeded random configurations against the torch port of the reference graph - the body of
tests/test_gpu_parity.py::test_random_configurations_match_the_oracle_chain over an arbitrary seed range.
usage (on a GPU box): python tools/fuzz_configs.py first_seed count"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_parity as T

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    try:
        T.test_random_configurations_match_the_oracle_chain(seed)
    except AssertionError as e:
        bad += 1
        print("seed %d FAILED: %s" % (seed, str(e)[:300]))
    except Exception as e:      # a configuration the library rejects is a finding too
        bad += 1
        print("seed %d ERROR: %r" % (seed, e))
print("%d seeds, %d failures" % (count, bad))
