"""oracle/ -- CPU restatement of the MMNN_STS multimodal-fusion training path.  TEST INFRASTRUCTURE ONLY.

Nothing under this directory is part of the product.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker / the timed CPU baseline.  The product
package `mmnn_sts_amd` never imports `oracle` and raises if its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * model arithmetic (DenseNet-3D, MLP, fusion heads, GradientBlender, GradCAM): PINNED -- `restatement.py` is
    checked against golden vectors produced in the build container by running the reference's own classes
    (`make_golden.py`, via `ref_shim.py`), committed under `tests/golden/`.
  * Cox partial-likelihood loss: the reference delegates to `pycox.models.loss.CoxPHLoss` (requirements.txt:16,
    un-pinned, not installed, not vendored).  It is restated from pycox's published algorithm
    (`cox_ph_loss_sorted`, eps 1e-7).  The reference holds no test or golden value for it => "parity unpinned"
    for that one function; the known-answer values of SURVEY.md 8(c) are asserted in tests/test_oracle.py.
"""
