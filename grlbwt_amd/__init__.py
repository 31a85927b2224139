"""grlbwt_amd -- MI355X-native BCR-BWT construction (grlBWT parse-then-induce path).

Host-side mirror of the reference interface over the C-ABI library
(include/grlbwt_hip.h).  The HIP library is mandatory: nothing here falls back
to a CPU path.
"""
