"""3pre_amd -- MI355X-native hot path of the 1-point-RANSAC EKF-SLAM reference ahtamjidi/3PRE.

The directory name starts with a digit, so import it with importlib:

    import importlib; pre3 = importlib.import_module("3pre_amd")

Contents: the C-ABI shared library (csrc/ -> lib/libpre3.so, declared in include/pre3.h), and thin
Python mirrors of the reference's functions for this path (ekf.py, matcher.py).  Nothing here computes on
the CPU and nothing imports oracle/.
"""
from . import _lib
from ._lib import CARTESIAN, F32, F64, INVDEPTH, LIB_PATH, X_K_K, X_K_KM1, Pre3Error, device_count
from .ekf import (CHI2INV_2_95, EkfFilter, compute_hypothesis_support_fast, generate_state_vector_pattern,
                  predict_state_and_covariance, update)
from .matcher import kNearestNeighbors, siftmatch, siftmatch_merge, siftmatch_partial

__all__ = ["EkfFilter", "update", "predict_state_and_covariance", "compute_hypothesis_support_fast", "generate_state_vector_pattern", "siftmatch", "siftmatch_partial", "siftmatch_merge",
           "kNearestNeighbors", "Pre3Error", "device_count", "LIB_PATH", "F64", "F32", "INVDEPTH", "CARTESIAN", "X_K_K", "X_K_KM1",
           "CHI2INV_2_95"]
