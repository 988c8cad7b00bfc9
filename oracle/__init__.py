"""CPU oracle for the scarlet proximal-gradient hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and there only as the
checker.  Nothing under ``scarlet_amd/`` imports it; the product path fails
loudly when the HIP library is missing instead of falling back to this code.

What it is: a plain NumPy (plus a few lines of C for the sequential radial
sweep) restatement of the reference algorithm for the path
``Blend.fit -> Blend.get_model -> Observation.render -> likelihood gradient ->
adaprox step -> proximal projections``.  Every function cites the reference
``file:line`` (relative to the pmelchior/scarlet checkout) it follows.

Parity pinning ("Oracle"):

* forward path (render, FFT convolution, diff kernel, log-likelihood, prox
  operators, monotonic-operator set-up tables): PINNED against the reference's
  own literal known-answer tables (tests/test_constraint.py, tests/test_fft.py,
  tests/test_observation.py) and against golden vectors produced by importing
  the reference in the build container (``oracle/refshim/make_golden.py``,
  fixtures in ``tests/golden/``).
* analytic gradient: pinned against central finite differences of the
  *reference's own* forward log-likelihood (golden fixture ``hsc_grad_fd``) and
  it follows the reference authors' analytic restatement in
  ``scarlet/lite/models.py:206-216,537-545``.
* optimizer arithmetic (``proxmin.adaprox`` with ``scheme="amsgrad"``):
  **parity unpinned**.  proxmin (>=0.6.11, setup.py:140) is a third-party
  dependency that is neither vendored in the reference checkout nor installed
  here; the update is restated from the reference's in-repo mirror
  ``scarlet/lite/parameters.py:274-305`` and from Reddi, Kale & Kumar (2018).
* ``scarlet.lite`` loop (``oracle/lite.py``): PINNED, bit for bit, against a
  25-iteration ``LiteBlend.fit`` trajectory with ``FistaParameter`` that the reference
  itself ran in the build container (every line of that loop is in-repo reference code;
  ``tests/golden/lite_fista.npz``).  The same convolution / gradient / proximal code
  serves the main path, so this also pins those pieces end to end.  The
  ``AdaproxParameter`` golden ran the reference's loop around the shim's AMSGrad
  moments (ours): it pins everything but those five lines.
"""

from . import fftconv, proxops, pgm, lite  # noqa: F401
