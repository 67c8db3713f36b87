"""CPU oracle for the runia_core OOD-scoring hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy/SciPy restatement of the
reference algorithms (CEA-LIST/runia_core v0.2.0) used as the *checker* by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``runia_core_amd/`` may import it: the product path is the HIP
library and fails loudly when that library or a GPU is missing.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_goldens.py)
against the golden vectors hard-coded in the reference's own unit tests and
against fixtures produced by importing the reference's hot-path source files
in the build container (tools/make_goldens.py -> tests/golden/*.npz).  The one
exception is the DropBlock mask path (``mc_stack``): the third-party
``dropblock==0.3.0`` package is absent from the image and the only value-level
pin in the reference needs a dataset download, so that stage is
"parity unpinned" (restated from the published algorithm).
"""
from .hotpath import *  # noqa: F401,F403
