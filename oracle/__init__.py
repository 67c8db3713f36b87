"""CPU oracle for the runia_core OOD-scoring hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy/SciPy restatement of the
reference algorithms (CEA-LIST/runia_core v0.2.0) used as the *checker* by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``runia_core_amd/`` may import it: the product path is the HIP
library and fails loudly when that library or a GPU is missing.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_goldens.py)
against the golden vectors hard-coded in the reference's own unit tests and
against fixtures produced by importing the reference's hot-path source files
in the build container (tools/make_goldens.py, tools/make_goldens_r2.py -> tests/golden/*.npz).
The sampler (``mc_stack``) is pinned since round 2 by outputs of the reference's own
``MCSamplerModule.forward`` loaded by path (tests/golden/ref_sampler.npz; the only restated
piece that runs there is the third-party ``dropblock==0.3.0`` layer, absent from the image).
Unpinned: ``roi_align`` (torchvision is absent; restated from its published algorithm - the
reference's per-ROI glue around it is pinned with that restatement plugged in,
tests/golden/ref_roi.npz) and ``philox4x32_10`` / ``counter_draws`` (the build's own
throughput-mode draws; checked against Random123's published known-answer vectors).
"""
from .hotpath import *  # noqa: F401,F403
