#!/usr/bin/env python
"""Drop-in for the reference's `python builder.py <music list> <db dir> [config]`."""
import sys

from pfann_amd import prewarm

if __name__ == "__main__":
    # HIP initialisation + code-object loading run on a thread under the import of torch below
    prewarm.start()
    from pfann_amd.builder import main
    prewarm.fast_exit(main(sys.argv))
