"""Fixtures for the experiment tests (not part of the product's suite: `pytest` from the root runs tests/ only)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
