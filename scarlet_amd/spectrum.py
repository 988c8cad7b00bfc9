"""Spectra of factorized components (reference scarlet/spectrum.py)."""

from functools import partial

from .bbox import Box
from .constraint import PositivityConstraint
from .frame import Frame
from .model import Model
from .parameter import Parameter, relative_step


class Spectrum(Model):
    def __init__(self, frame, *parameters, bbox=None):
        assert isinstance(frame, Frame)
        self.frame = frame
        assert isinstance(bbox, Box)
        self.bbox = bbox
        super().__init__(*parameters)


class TabulatedSpectrum(Spectrum):
    """Free-form spectrum: one amplitude per channel, kept slightly positive
    (``PositivityConstraint(zero=1e-20)``), steps of 1 % of the mean amplitude
    with the floor ``min_step`` (e.g. the noise rms per channel)."""

    def __init__(self, frame, spectrum, bbox=None, min_step=0):
        if isinstance(spectrum, Parameter):
            assert spectrum.name == "spectrum"
        else:
            spectrum = Parameter(
                spectrum, name="spectrum",
                step=partial(relative_step, factor=1e-2, minimum=min_step),
                constraint=PositivityConstraint(zero=1e-20),
            )
        if bbox is None:
            assert frame.bbox[0].shape == spectrum.shape
            bbox = Box(spectrum.shape)
        else:
            assert bbox.shape == spectrum.shape
        super().__init__(frame, spectrum, bbox=bbox)

    def get_model(self, *parameters):
        return self.get_parameter(0, *parameters)
