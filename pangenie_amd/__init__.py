"""pangenie_amd — MI355X-native genotyping hot path of PanGenie (emissions + diploid
forward-backward HMM) behind the reference's HMM / GenotypingResult interface.

Layout: csrc/ = hand-written HIP kernels + the C ABI (include/pangenie_hmm.h);
host/ = C++ mirror of the reference's host-side interface (HMM adapter, UniqueKmers,
ProbabilityTable, GenotypingResult); the Python modules here are thin plumbing
(ctypes, synthetic panels, torch.distributed sharding) for tests and bench.py.
"""
__version__ = "0.1.0"
