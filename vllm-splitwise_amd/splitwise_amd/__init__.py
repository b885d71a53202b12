"""MI355X-native disaggregated prefill/decode hot path (host side).

The native operators live in ``libsplitwise_hip.so`` (built from ``../csrc``) and are
reached through :mod:`splitwise_amd._C`.
"""
__version__ = "0.1.0"
