"""`nerf_sampling` import names served by nerf_sampling_amd (the MI355X build) -- an ALIAS package, no code of its own.

Put `<repo>/compat` ahead of the reference checkout on PYTHONPATH and the reference's experiment scripts, yaml files
(`module: "nerf_sampling.trainers.DepthNetTrainer"`) and tests import this build's operators under the names they
already use.  Every submodule here re-exports a nerf_sampling_amd module; nothing is copied from the reference.
Kept OUT of the repo root on purpose: tools/make_golden.py imports the real reference under the same name.
"""
