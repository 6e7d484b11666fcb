"""Only the piece of the reference's tool/qmath.py the BPR path uses (sigmoid, :115-116).
The HIP kernel evaluates the same expression in double (train_kernels.hpp)."""
from math import exp


def sigmoid(val):
    return 1 / (1 + exp(-val))
