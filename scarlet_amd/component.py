"""Components: hyperspectral models inside a bounding box.

Same classes, constructor signatures and failure modes as the reference
(scarlet/component.py:14-291) so that scripts written for it keep working; the bodies are
this package's own.  ``get_model`` here is the host-side, inspection-time evaluation;
during ``Blend.fit`` the same outer products are rendered on the GPU.
"""

import numpy as np

from .bbox import Box, overlapped_slices
from .constraint import PositivityConstraint
from .fft import fast_zero_pad
from .frame import Frame
from .model import Model, UpdateException
from .morphology import Morphology
from .parameter import Parameter, relative_step
from .spectrum import Spectrum


def _update_children(component, rebuild_box):
    """Run ``update()`` on every child; when one of them reports a changed box
    (``UpdateException``), give the container its new box first and pass the exception
    on -- the remaining children are not visited (component.py:172-185, 280-290)."""
    for child in component.children:
        try:
            child.update()
        except UpdateException:
            component.bbox = rebuild_box()
            raise


class Component(Model):
    """A model confined to ``bbox`` inside ``frame`` (default: the whole frame).  Keeps
    the pair of slices that place the boxed model into the frame up to date whenever
    either of the two changes."""

    def __init__(self, frame, *parameters, children=None, bbox=None):
        assert isinstance(frame, Frame)
        assert bbox is None or isinstance(bbox, Box)
        self._frame = frame
        self._bbox = frame.bbox if bbox is None else bbox
        self._reslice()
        super().__init__(*parameters, children=children)

    def _reslice(self):
        # (worked out when somebody asks: a fit that resizes thousands of boxes on the
        # device never looks at most of them)
        self._placement = None

    def _place(self):
        if self._placement is None:
            self._placement = overlapped_slices(self._frame.bbox, self._bbox)
        return self._placement

    @property
    def _model_frame_slices(self):
        return self._place()[0]

    @property
    def _model_slices(self):
        return self._place()[1]

    @property
    def bbox(self):
        return self._bbox

    @bbox.setter
    def bbox(self, b):
        self._bbox = self._frame.bbox if b is None else b
        self._reslice()

    @property
    def frame(self):
        return self._frame

    @frame.setter
    def frame(self, f):
        self._frame = f
        self._reslice()

    def model_to_box(self, bbox=None, model=None):
        """Zero image of ``bbox`` (default: the frame) with the boxed ``model`` (default:
        the current one) written into the overlap."""
        model = self.get_model() if model is None else model
        target = self.frame.bbox if bbox is None else bbox
        if target == self.frame.bbox:
            into, out_of = self._model_frame_slices, self._model_slices  # cached pair
        else:
            into, out_of = overlapped_slices(target, self.bbox)
        placed = np.zeros(target.shape, dtype=model.dtype)
        placed[into] = model[out_of]
        return placed

    def _in_frame(self, model, frame):
        """``model`` as is, or embedded into ``frame`` when one is given."""
        return model if frame is None else self.model_to_box(frame.bbox, model)


class FactorizedComponent(Component):
    """Spectrum (C,) x morphology (h, w) or (1 | C, h, w): the component the device loop
    fits."""

    def __init__(self, frame, spectrum, morphology):
        assert isinstance(spectrum, Spectrum)
        assert isinstance(morphology, Morphology)
        super().__init__(frame, children=[spectrum, morphology],
                         bbox=self._joint_box(spectrum, morphology))

    @staticmethod
    def _joint_box(spectrum, morphology):
        return spectrum.bbox @ morphology.bbox[-2:]

    def get_model(self, *parameters, frame=None):
        spectrum, morphology = self.get_models_of_children(*parameters)
        if morphology.ndim not in (2, 3):
            raise AttributeError("morphology must be 2D or 3D")
        # a 2-D morphology is shared by all channels, a 3-D one broadcasts over them
        image = morphology if morphology.ndim == 3 else morphology[np.newaxis]
        return self._in_frame(spectrum[:, np.newaxis, np.newaxis] * image, frame)

    def update(self):
        _update_children(self, lambda: self._joint_box(*self.children))


class CubeComponent(Component):
    """Free-form (C, h, w) cube.  Supported for model evaluation; not fitted by
    the device loop."""

    def __init__(self, frame, cube, bbox=None):
        if not isinstance(cube, Parameter):
            cube = Parameter(cube, name="cube", step=relative_step,
                             constraint=PositivityConstraint())
        assert cube.name == "cube"
        super().__init__(frame, cube, bbox=bbox)

    def get_model(self, *parameters, frame=None):
        return self._in_frame(self.get_parameter(0, *parameters), frame)


class CombinedComponent(Component):
    """Sum (or product) of child components of one frame, evaluated over the box it
    was given at construction (the first child's) or after the last resize (their
    union)."""

    def __init__(self, components, operation="add"):
        assert len(components)
        frame = components[0].frame
        assert all(isinstance(c, Component) and c.frame is frame for c in components)
        assert operation in ["add", "multiply"]
        self.operation = operation
        super().__init__(frame, children=components, bbox=components[0].bbox)

    def get_model(self, *parameters, frame=None):
        box = self.bbox
        combine = np.add if self.operation == "add" else np.multiply
        total = np.zeros(box.shape)
        for child, part in zip(self.children,
                               self.get_models_of_children(*parameters, frame=None)):
            if child.bbox != box:
                margins = tuple((lo - b_lo, b_hi - hi) for lo, hi, b_lo, b_hi in
                                zip(child.bbox.start, child.bbox.stop, box.start, box.stop))
                part = fast_zero_pad(part, margins)
            combine(total, part, out=total)
        return self._in_frame(total, frame)

    def _union_box(self):
        box = self.children[0].bbox.copy()
        for c in self.children[1:]:
            box |= c.bbox
        return box

    def update(self):
        _update_children(self, self._union_box)
