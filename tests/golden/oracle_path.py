"""Puts <repo>/oracle on sys.path (test-side helper; the product never imports the oracle)."""
import os
import sys

_ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle")
if _ORACLE not in sys.path:
    sys.path.insert(0, _ORACLE)
