"""Import-path shim: ``from model.unipose import unipose`` (reference unipose.py:26) resolves to the
MI355X-native implementation in ``unipose_amd``."""
