"""CPU oracle for the StabStitch++ inference hot path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch CPU restatement
(pure PyTorch-CPU / numpy) of the reference algorithm
(`/root/reference/Full_model_inference/Codes`).  It exists so that the HIP
path in `stabstitch2_amd/` can be checked against it.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; the product package never does (tests/test_layout.py enforces this).

Parity pinning: the reference ships no tests, fixtures or golden vectors
(SURVEY.md §4).  The oracle is pinned instead against outputs of the reference
itself, produced in the build container by `tests/golden/make_goldens.py`
(imports the reference under a small shim) and committed as `.npz` fixtures
under `tests/golden/`.  `tests/test_oracle_golden.py` checks every oracle
function against those fixtures.

Third-party pieces whose source is not under /root/reference (torchvision
0.14.1 resnet18 / GaussianBlur, scikit-image 0.15 PSNR/SSIM) are restated
from their published semantics; see DESIGN.md "parity pinning".
"""

GRID_H = 6
GRID_W = 8
