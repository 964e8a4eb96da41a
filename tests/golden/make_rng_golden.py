#!/usr/bin/env python3
"""Pins the device generator's stream (csrc/md_rng.h): draws of every kind for fixed (seed, call order) from the CPU test
double, which runs the same Philox code as the device kernels; tests/test_device_rng.py replays them on both targets
(bit-exact for uniforms, integers, binomials, permutations; 1e-6 for normals, whose log / cos come from the platform's libm).
    python tests/golden/make_rng_golden.py        -> tests/golden/rng_stream.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def draw(nd):
    nd.device_rng(True, seed=20261004)
    out = {
        "uniform_f64": nd.random_uniform((257,)).get(),
        "uniform_f32": nd.random_uniform((3, 85), np.float32).get(),
        "normal_f64": nd.random_normal((129,)).get(),
        "normal_f32": nd.random_normal((130,), np.float32).get(),
        "integers_i64": nd.random_integers(-5, 1 << 40, (200,)).get(),
        "integers_i32": nd.random_integers(0, 7, (4, 50), np.int32).get(),
        "binomial_1": nd.random_binomial(1, 0.3, (300,)).get(),
        "binomial_40": nd.random_binomial(40, 0.75, (100,)).get(),
        "permutation": nd.random_permutation(1000).get(),
    }
    nd.device_rng(False)
    return out


if __name__ == "__main__":
    import conftest
    lib, gpu = conftest.bound_library()
    assert not gpu, "generate on the CPU test double"
    from minidiff_amd import ndarray as nd
    np.savez_compressed(os.path.join(HERE, "rng_stream.npz"), **draw(nd))
    print("written")
