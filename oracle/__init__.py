"""oracle/ -- TEST INFRASTRUCTURE ONLY (parity unpinned).

fp32 CPU restatement, in plain PyTorch ops, of the ControlNet-conditioned
latent-diffusion sampling path that RichardObi/ccnet drives
(src/python/training/run_controlnet_inference.py:211-256 and
src/python/training/util.py:934-962,1043-1092).  All arithmetic on that path
lives in third-party packages that are NOT vendored in the reference and are
not installed here (monai-generative unpinned, monai>=1.2.0, diffusers==0.32.0;
requirements.txt:15-19), so the networks and schedulers below restate the
published monai-generative algorithm.  The reference
holds no tests, golden vectors or fixtures for this path, and the third-party
packages cannot be imported here: **parity unpinned**.  What pins the oracle
instead: closed-form constants, structural invariants,
and torch's own nn.functional ops.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (ccnet_amd/) never does.
"""
