#!/usr/bin/env python
"""Drop-in for the reference's `python matcher.py <query list> <db dir> <result file>`."""
import sys

from pfann_amd.matcher import main

if __name__ == "__main__":
    sys.exit(main(sys.argv))
