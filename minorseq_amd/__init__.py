"""minorseq_amd -- MI355X-native pileup + Fisher hot path of PacBio MinorSeq's juliet.

The product is the C-ABI HIP library (include/msq_abi.h -> minorseq_amd/lib/libmsq_hip.so) and the C++ front
that mirrors the reference's MSA / AminoAcidCaller interface (minorseq_amd/cpp).  The Python modules here are
harness plumbing for tests and bench.py.
"""
__all__ = ["abi"]
