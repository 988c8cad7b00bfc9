"""Spectra of factorized components.

Public surface mirrors the reference (scarlet/spectrum.py:9-71): ``Spectrum(frame,
*parameters, bbox=)`` and ``TabulatedSpectrum(frame, spectrum, bbox=None, min_step=0)``
with the same argument meaning and the same failures (AssertionError on a wrong frame /
box type, on a ``Parameter`` that is not called "spectrum", on a box whose shape differs
from the spectrum's).  The device loop reads three things off a TabulatedSpectrum: the
values, the step rule and the positivity floor (``Blend._specs``).
"""

from functools import partial

from .bbox import Box
from .constraint import PositivityConstraint
from .frame import Frame
from .model import Model
from .parameter import Parameter, relative_step

#: floor of the positivity constraint and relative step of a free-form spectrum
#: (spectrum.py:54-56); the update kernel hard-codes the same floor
SPECTRUM_FLOOR = 1e-20
SPECTRUM_REL_STEP = 1e-2


def _require(value, kind, what):
    assert isinstance(value, kind), "{} must be a {}".format(what, kind.__name__)
    return value


class Spectrum(Model):
    """1-D spectral dependence of a ``FactorizedComponent``; ``bbox`` is its 1-D box in
    the channel axis of ``frame``."""

    def __init__(self, frame, *parameters, bbox=None):
        self.frame = _require(frame, Frame, "frame")
        self.bbox = _require(bbox, Box, "bbox")
        super().__init__(*parameters)


class TabulatedSpectrum(Spectrum):
    """Free-form spectrum: one amplitude per channel, kept above ``SPECTRUM_FLOOR``, with
    steps of 1 % of the mean amplitude but at least ``min_step`` (typically the noise
    rms per channel)."""

    def __init__(self, frame, spectrum, bbox=None, min_step=0):
        values = self._as_parameter(spectrum, min_step)
        # without a box the spectrum spans all channels of the frame
        expected = frame.bbox[0].shape if bbox is None else bbox.shape
        assert tuple(expected) == tuple(values.shape), "spectrum does not fill its box"
        super().__init__(frame, values, bbox=Box(values.shape) if bbox is None else bbox)

    @staticmethod
    def _as_parameter(spectrum, min_step):
        """A ready ``Parameter`` is taken as is (it must be named "spectrum"); plain
        arrays get the standard step rule and positivity constraint."""
        if isinstance(spectrum, Parameter):
            assert spectrum.name == "spectrum"
            return spectrum
        return Parameter(
            spectrum,
            name="spectrum",
            step=partial(relative_step, factor=SPECTRUM_REL_STEP, minimum=min_step),
            constraint=PositivityConstraint(zero=SPECTRUM_FLOOR),
        )

    def get_model(self, *parameters):
        return self.get_parameter(0, *parameters)
