"""CPU oracle for the scLDM latent-diffusion hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference
algorithm for the path SURVEY.md section 8 names (DiT forward + CFG, fixed-step
Euler/Heun sampling, MCAB encode/decode + NB head).  Each function cites the
reference file:line it follows (paths relative to /root/reference/).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / reported baseline: the
product (`scldm_amd`) never imports `oracle` and has no CPU fallback.

Parity pinning: checked against golden vectors produced by importing the
reference's own modules in the build container (`tests/golden/make_golden.py`,
fixtures committed under `tests/golden/*.npz`).  The fixed-step Euler/Heun
stepping arithmetic lives in the un-vendored, unpinned third-party `torchdiffeq`
(reference call site src/scldm/transport/integrators.py:111): that boundary is
"parity unpinned" and is anchored on an analytic known-answer test instead.
"""
