#!/usr/bin/env python3
"""Command-line entry of the ISA hazard lint (the implementation lives in the package: arreau_amd/_isa_lint.py, so an
installed or copied arreau_amd builds without the repository's tools/ directory).

    python tools/isa_lint.py [--all] <file.s> ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arreau_amd._isa_lint import *  # noqa: F401,F403,E402  (tests import the lint through this module, too)
from arreau_amd import _isa_lint as _impl  # noqa: E402

if __name__ == "__main__":
    sys.exit(_impl.main(sys.argv[1:]))
