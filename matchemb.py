#!/usr/bin/env python
"""Drop-in for the reference's `python matchemb.py ...` (see pfann_amd/matchemb.py)."""
import sys

from pfann_amd import prewarm

if __name__ == "__main__":
    prewarm.start()                     # HIP init + code-object loading on a thread under the import of torch below
    from pfann_amd.matchemb import main
    prewarm.fast_exit(main(sys.argv))
