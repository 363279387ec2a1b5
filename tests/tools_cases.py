"""The golden cases (single definition shared with tools/make_golden.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import CASES, GPU_CASES  # noqa: E402,F401
