"""Components: hyperspectral models inside a bounding box (reference
scarlet/component.py).  ``get_model`` here is the host-side, inspection-time
evaluation; during ``Blend.fit`` the same outer products are rendered on the GPU."""

import numpy as np

from .bbox import Box, overlapped_slices
from .constraint import PositivityConstraint
from .fft import fast_zero_pad
from .frame import Frame
from .model import Model, UpdateException
from .morphology import Morphology
from .parameter import Parameter, relative_step
from .spectrum import Spectrum


class Component(Model):
    def __init__(self, frame, *parameters, children=None, bbox=None):
        assert isinstance(frame, Frame)
        if bbox is None:
            bbox = frame.bbox
        assert isinstance(bbox, Box)
        self._bbox = bbox
        self.frame = frame
        super().__init__(*parameters, children=children)

    def _reslice(self):
        self._model_frame_slices, self._model_slices = overlapped_slices(
            self._frame.bbox, self._bbox
        )

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
        """Embed the boxed model into (the part of) ``bbox`` it overlaps."""
        if model is None:
            model = self.get_model()
        if bbox is None or bbox == self.frame.bbox:
            bbox = self.frame.bbox
            frame_sl, model_sl = self._model_frame_slices, self._model_slices
        else:
            frame_sl, model_sl = overlapped_slices(bbox, self.bbox)
        out = np.zeros(bbox.shape, dtype=model.dtype)
        out[frame_sl] = model[model_sl]
        return out


class FactorizedComponent(Component):
    """Spectrum (C,) x morphology (h, w): the component the device loop fits."""

    def __init__(self, frame, spectrum, morphology):
        assert isinstance(spectrum, Spectrum)
        assert isinstance(morphology, Morphology)
        bbox = spectrum.bbox @ morphology.bbox[-2:]
        super().__init__(frame, children=[spectrum, morphology], bbox=bbox)

    def get_model(self, *parameters, frame=None):
        spectrum, morphology = self.get_models_of_children(*parameters)
        if morphology.ndim == 2:
            model = spectrum[:, None, None] * morphology[None, :, :]
        elif morphology.ndim == 3:
            model = spectrum[:, None, None] * morphology
        else:
            raise AttributeError("morphology must be 2D or 3D")
        if frame is not None:
            model = self.model_to_box(frame.bbox, model)
        return model

    def update(self):
        for child in self.children:
            try:
                child.update()
            except UpdateException as exc:
                # follow the morphology's new box
                spectrum, morphology = self.children
                self.bbox = spectrum.bbox @ morphology.bbox[-2:]
                raise exc


class CubeComponent(Component):
    """Free-form (C, h, w) cube.  Supported for model evaluation; not fitted by
    the device loop."""

    def __init__(self, frame, cube, bbox=None):
        if isinstance(cube, Parameter):
            assert cube.name == "cube"
        else:
            cube = Parameter(cube, name="cube", step=relative_step,
                             constraint=PositivityConstraint())
        super().__init__(frame, cube, bbox=bbox)

    def get_model(self, *parameters, frame=None):
        model = self.get_parameter(0, *parameters)
        if frame is not None:
            model = self.model_to_box(frame.bbox, model)
        return model


class CombinedComponent(Component):
    """Sum (or product) of child components over the first child's box."""

    def __init__(self, components, operation="add"):
        assert len(components)
        frame = components[0].frame
        for c in components:
            assert isinstance(c, Component)
            assert c.frame is frame
        super().__init__(frame, children=components, bbox=components[0].bbox)
        assert operation in ["add", "multiply"]
        self.operation = operation

    def get_model(self, *parameters, frame=None):
        models = self.get_models_of_children(*parameters, frame=None)
        bbox = self.bbox
        model = np.zeros(bbox.shape)
        for child, m in zip(self.children, models):
            if child.bbox != bbox:
                pad = tuple(
                    (child.bbox.start[d] - bbox.start[d], bbox.stop[d] - child.bbox.stop[d])
                    for d in range(bbox.D)
                )
                m = fast_zero_pad(m, pad)
            if self.operation == "add":
                model += m
            else:
                model *= m
        if frame is not None:
            model = self.model_to_box(frame.bbox, model)
        return model

    def update(self):
        for child in self.children:
            try:
                child.update()
            except UpdateException as exc:
                box = self.children[0].bbox.copy()
                for c in self.children[1:]:
                    box |= c.bbox
                self.bbox = box
                raise exc
