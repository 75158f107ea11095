#!/usr/bin/env python
"""Drop-in for the reference's `python builder.py <music list> <db dir> [config]`."""
import sys

from pfann_amd.builder import main

if __name__ == "__main__":
    sys.exit(main(sys.argv))
