#!/usr/bin/env python
"""Drop-in for the reference's `python matchemb.py ...` (see pfann_amd/matchemb.py)."""
import sys

from pfann_amd.matchemb import main

if __name__ == "__main__":
    sys.exit(main(sys.argv))
