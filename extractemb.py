#!/usr/bin/env python
"""Drop-in for the reference's `python extractemb.py ...` (see pfann_amd/extractemb.py)."""
import sys

from pfann_amd.extractemb import main

if __name__ == "__main__":
    sys.exit(main(sys.argv))
