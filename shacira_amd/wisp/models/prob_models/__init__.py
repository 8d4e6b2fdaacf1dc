from .bit_estimator import Bitparm, BitEstimator
