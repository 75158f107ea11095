#!/usr/bin/env python
"""Drop-in for the reference's `python extractemb.py ...` (see pfann_amd/extractemb.py)."""
import sys

from pfann_amd import launch, prewarm

if __name__ == "__main__":
    _rc = launch.self_launch_if_asked(sys.argv)      # PFANN_GPUS=N: N ranks of this command, one per GPU (no torch import yet)
    if _rc is not None:
        sys.exit(_rc)
    prewarm.start(engine=prewarm.engine_job_for("extractemb", sys.argv))
    from pfann_amd.extractemb import main
    prewarm.fast_exit(main(sys.argv))
