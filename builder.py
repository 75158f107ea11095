#!/usr/bin/env python
"""Drop-in for the reference's `python builder.py <music list> <db dir> [config]`."""
import sys

from pfann_amd import launch, prewarm

if __name__ == "__main__":
    _rc = launch.self_launch_if_asked(sys.argv)      # PFANN_GPUS=N: N ranks of this command, one per GPU (no torch import yet)
    if _rc is not None:
        sys.exit(_rc)
    # HIP initialisation, code-object loading and the model (configs.json -> context, model.pt -> weights) on a thread under
    # the import of torch below
    prewarm.start(engine=prewarm.engine_job_for("builder", sys.argv))
    from pfann_amd.builder import main
    prewarm.fast_exit(main(sys.argv))
