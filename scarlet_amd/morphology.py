"""Image morphologies of factorized components (reference
scarlet/morphology.py:26-207, 476-513, 607-688).  Parametric profiles (Gaussian,
Spergel, starlet) are outside the scope of this package."""

import numpy as np
import numpy.ma as ma

from .bbox import Box, overlapped_slices
from .constraint import (
    CenterOnConstraint,
    ConstraintChain,
    MonotonicityConstraint,
    NormalizationConstraint,
    PositivityConstraint,
    SymmetryConstraint,
)
from .frame import Frame
from .model import Model, UpdateException
from .parameter import prepare_param, Parameter, relative_step


def get_minimal_boxsize(size, min_size=21, increment=10):
    """Smallest odd box size ``min_size + k * increment`` that holds ``size``
    pixels (reference initialization.py:173-177)."""
    boxsize = min_size
    while boxsize < size:
        boxsize += increment
    return boxsize


def _empty_margin(image, thresh):
    """Width of the frame of pixels <= ``thresh`` around ``image``: the largest ``d`` such
    that the d outermost rows and columns on every side hold nothing above ``thresh``
    (the whole image counts as margin when it is empty)."""
    mask = np.asarray(image) > thresh
    rows = mask.any(axis=1)
    if not rows.any():
        return (min(image.shape) + 1) // 2
    cols = mask.any(axis=0)
    # first and last occupied row / column, counted from the nearer edge
    return int(min(rows.argmax(), cols.argmax(), rows[::-1].argmax(), cols[::-1].argmax()))


def _edge_pull(image, m, v, step):
    """Mean pull of the next Adam step over each of the four edges of the box
    (morphology.py:166-176): ``-m / sqrt(sqrt(v)) * step * (image > 0)`` averaged over the
    edge pixels with ``v != 0``.  The reference evaluates the whole image as a masked array
    and takes the means of four slices; this is the same arithmetic on the four edges only --
    the masked mean is the sum of the zero-filled slice over the count of unmasked entries --
    bit for bit (tests/test_host_logic.py), at a tenth of the cost.  An edge without an
    unmasked pixel gives nan, as the masked constant does when it is put into an array."""
    out = np.empty(4)
    for e, sl in enumerate(((slice(None), 0), (slice(None), -1), (0, slice(None)), (-1, slice(None)))):
        vv = v[sl]
        seen = vv != 0
        count = int(seen.sum())
        if count == 0:
            out[e] = np.nan
            continue
        gu = -m[sl] / np.sqrt(np.sqrt(np.where(seen, vv, 1.0))) * step
        pull = gu * (image[sl] > 0)
        out[e] = np.where(seen, pull, 0.0).sum() * 1.0 / count
    return out


class Morphology(Model):
    """Spatial part of a factorized component inside ``bbox`` (default: the box of
    ``frame``)."""

    def __init__(self, frame, *parameters, bbox=None):
        assert isinstance(frame, Frame)
        assert bbox is None or isinstance(bbox, Box)
        self.frame = frame
        self.bbox = frame.bbox if bbox is None else bbox
        super().__init__(*parameters)

    def shrink_box(self, image, thresh=0):
        """Peel off empty borders; adopt the next smaller standard box size
        (morphology.py:50-67)."""
        size = max(image.shape)
        newsize = get_minimal_boxsize(size - 2 * _empty_margin(image, thresh))
        if newsize < size:
            inset = (size - newsize) // 2
            self.bbox.origin = tuple(o + inset for o in self.bbox.origin)
            self.bbox.shape = (newsize, newsize)


class ImageMorphology(Morphology):
    """Free-form image morphology inside ``bbox``."""

    def __init__(self, frame, image, bbox=None, shifting=False, shift=None, resizing=True):
        if isinstance(image, Parameter):
            assert image.name == "image"
        else:
            image = Parameter(image, name="image", step=relative_step,
                              constraint=PositivityConstraint())
        # without a box the image must cover the frame's spatial extent
        assert image.shape == (frame.bbox[1:].shape if bbox is None else bbox.shape)
        bbox = Box(image.shape) if bbox is None else bbox
        self.resizing = resizing
        self.shifting = shifting
        if shift is None:
            # kept for parameter-order compatibility with the reference, which creates
            # this unused 2-vector for every image morphology (morphology.py:113)
            shift = Parameter(np.zeros(2), name="shift", step=1e-2, fixed=self.shifting)
        else:
            assert shift.shape == (2,)
            if not isinstance(shift, Parameter):
                shift = Parameter(shift, name="shift", step=1e-2)
        super().__init__(frame, image, shift, bbox=bbox)

    def get_model(self, *parameters):
        image = self.get_parameter(0, *parameters)
        if self.shifting:
            # Fourier sub-pixel shift (morphology.py:124-130).  The reference creates
            # the shift with ``fixed=self.shifting`` (morphology.py:113), so with
            # shifting=True and no explicit ``shift`` it stays at zero and this is the
            # identity up to FFT round-off.
            from . import fft

            image = fft.shift(image, self.get_parameter(1, *parameters), return_Fourier=False)
        return image

    def update(self):
        """Every 10 iterations: shrink the box when all edges are empty, grow it
        when the next Adam step pulls flux over an edge; the optimizer state is
        sliced / padded along and the step halved (reference morphology.py:132-207)."""
        image = self._parameters[0]
        if not self.resizing or image.fixed:
            return
        bbox = self.bbox.copy()
        self.shrink_box(image)
        if bbox != self.bbox:
            # the new box sits `inset` pixels inside the old one on every side (shrink_box)
            inset = self.bbox.origin[-1] - bbox.origin[-1]
            sl = tuple(slice(inset, inset + n) for n in self.bbox.shape)

            def cut(a):
                return a[sl] if a is not None else None

            image = Parameter(image[sl], name=image.name, prior=image.prior,
                              constraint=image.constraint, step=image.step / 2,
                              fixed=image.fixed, m=cut(image.m), v=cut(image.v),
                              vhat=cut(image.vhat))
            self._parameters = (image,) + self._parameters[1:]
            raise UpdateException
        if image.m is not None:
            edge_pull = _edge_pull(np.asarray(image), image.m, image.v, image.step)
            if np.any(edge_pull > 0.1):
                size = max(bbox.shape)
                newsize = get_minimal_boxsize(size + 1)
                pad = (newsize - size) // 2

                def grow(a):
                    return np.pad(a, pad, mode="constant") if a is not None else None

                image = Parameter(np.pad(image, pad, mode="linear_ramp"), name=image.name,
                                  prior=image.prior, constraint=image.constraint,
                                  step=image.step / 2, fixed=image.fixed, m=grow(image.m),
                                  v=grow(image.v), vhat=grow(image.vhat))
                self._parameters = (image,) + self._parameters[1:]
                self.bbox.origin = tuple(o - pad for o in self.bbox.origin)
                self.bbox.shape = (newsize, newsize)
                raise UpdateException


class ExtendedSourceMorphology(ImageMorphology):
    """Image morphology for galaxies: monotonic from the centre (``monotonic`` in
    'flat' / 'angle' / 'nearest' or None), optionally symmetric, positive, centre
    pixel kept on, normalised to unit maximum; step 1e-2."""

    def __init__(self, frame, center, image, bbox=None, monotonic="angle", symmetric=False,
                 min_grad=0, shifting=False, resizing=True):
        # booleans are the old spelling of the monotonicity argument
        weighting = {True: "angle", False: None}.get(monotonic, monotonic) \
            if isinstance(monotonic, bool) else monotonic
        shape_terms = []
        if weighting is not None:
            shape_terms.append(MonotonicityConstraint(neighbor_weight=weighting,
                                                      min_gradient=min_grad))
        if symmetric:
            shape_terms.append(SymmetryConstraint())
        # the order of the fused device chain (constraint._DEVICE_ORDER)
        constraints = shape_terms + [PositivityConstraint(), CenterOnConstraint(),
                                     NormalizationConstraint("max")]
        image = Parameter(image, name="image", step=1e-2, constraint=ConstraintChain(*constraints))
        self.pixel_center = np.round(center).astype("int")
        # with shifting the sub-pixel offset of the centre becomes a free parameter
        # (morphology.py:673-676); Blend.fit refuses it: free Fourier shifts do not run
        # on the device yet
        self.shift = (Parameter(np.asarray(center, dtype=float) - self.pixel_center,
                                name="shift", step=1e-1) if shifting else None)
        super().__init__(frame, image, bbox=bbox, shifting=shifting, shift=self.shift,
                         resizing=resizing)

    @property
    def center(self):
        if self.shift is not None:
            return self.pixel_center + self.shift
        return self.pixel_center


class PointSourceMorphology(Morphology):
    """The model PSF (``frame.psf``) evaluated at a free sub-pixel ``center``
    (reference morphology.py:476-513).  The box is the PSF box moved to the rounded
    initial centre and never changes; the only parameter is the centre."""

    def __init__(self, frame, center):
        from .psf import PSF

        assert frame.psf is not None and isinstance(frame.psf, PSF)
        self.psf = frame.psf
        # the PSF box, centred on the origin, moves to the pixel nearest the centre
        cy, cx = (int(c) for c in np.rint(np.asarray(center, dtype=float)))
        self.center = prepare_param(center, name="center")
        super().__init__(frame, self.center, bbox=self.psf.bbox + (0, cy, cx))

    def get_model(self, *parameters):
        """Model-frame PSF image evaluated at the sub-pixel offset of the centre from
        the middle of the box."""
        (y_lo, y_hi), (x_lo, x_hi) = self.bbox.bounds[1:]
        middle = np.array([(y_lo + y_hi) / 2, (x_lo + x_hi) / 2])
        return self.psf.get_model(offset=np.asarray(self.get_parameter(0, *parameters)) - middle)

    @property
    def integral(self):
        return self.psf.get_model().sum()
