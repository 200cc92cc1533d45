"""oracle/ -- CPU restatement of the CREStE perception->costmap hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import anything from this package, and only as the *checker*
(or the timed CPU baseline) -- never as the thing shipped.  The product path
(`creste_public_amd`) never imports `oracle` and fails loudly when the HIP
library is missing.

What it is: a plain PyTorch-CPU (fp32) re-statement of the reference's
algorithm for SURVEY.md section 8 rows A1..A11, written from the reference's
behaviour, each function citing the reference file:line it follows.  The
modules keep the reference's `state_dict` key names so the same weights load
into the oracle and into the HIP-backed host modules.

Parity pin status
-----------------
* Hand-written reference stages (conv blocks, Up, DeconvHead, splat, VIN value
  iteration, SVF, MaxEnt-IRL loss, depth/fov/utility helpers): PINNED.  The
  reference's own Python was imported in the build container
  (`tests/golden/make_golden.py`, import shims per SURVEY.md App. B) and its
  outputs are committed as `tests/golden/*.npz`; `tests/test_oracle_golden.py`
  checks this restatement against them.
* EfficientNet-B0 MBConv trunk (`efficientnet_pytorch`, un-pinned pip
  dependency, latest public release 0.7.1) and ResNet-18 BasicBlocks
  (`torchvision>=0.16`): the packages are NOT in /root/reference and not
  installed here, and the reference holds no tests/golden vectors at those
  boundaries -> "parity unpinned" for those two sub-graphs.  They restate the
  published architectures (block table, static "same" padding computed for a
  224^2 image, SE ratio 0.25 of block input filters, swish, BN eps 1e-3); shapes
  are cross-checked against the sizes the survey probed (612->306->153->76->38->19).
"""
