"""Run pytest against an experiment build of the library: python scripts/debug/pytest_with_lib.py <lib.so> <pytest args...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from magpo_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
