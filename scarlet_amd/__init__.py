"""scarlet_amd: the proximal-gradient fitting loop of pmelchior/scarlet on
AMD MI355X (gfx950), behind scarlet's own Python API.

Hot path (``Blend.fit`` -> render -> FFT convolution -> likelihood gradient ->
AMSGrad step -> proximal projections) = hand-written HIP kernels + rocFFT in
``libscarlet_amd.so`` (C ABI: ``include/scarlet_amd.h``); this package holds the
host-side mirror of the reference interface and the batching/sharding drivers.
There is no CPU fallback for the loop.
"""

from .bbox import Box, overlapped_slices  # noqa: F401
from .constraint import (  # noqa: F401
    Constraint,
    ConstraintChain,
    PositivityConstraint,
    NormalizationConstraint,
    L0Constraint,
    L1Constraint,
    ThresholdConstraint,
    MonotonicityConstraint,
    MonotonicMaskConstraint,
    SymmetryConstraint,
    CenterOnConstraint,
)
from .parameter import Parameter, relative_step  # noqa: F401
from .prior import Prior  # noqa: F401
from .psf import PSF, ImagePSF, FunctionPSF, GaussianPSF, MoffatPSF  # noqa: F401
from .batch import BlendBatch, ComponentSpec, PointSourceSpec  # noqa: F401
from .frame import Frame  # noqa: F401
from .observation import Observation  # noqa: F401
from .renderer import (  # noqa: F401
    Renderer,
    NullRenderer,
    ConvolutionRenderer,
    ResolutionRenderer,
)
from .wcs import LinearWCS, TanWCS  # noqa: F401
from .spectrum import Spectrum, TabulatedSpectrum  # noqa: F401
from .morphology import (  # noqa: F401
    Morphology,
    ImageMorphology,
    ExtendedSourceMorphology,
    PointSourceMorphology,
)
from .component import (  # noqa: F401
    Component,
    FactorizedComponent,
    CubeComponent,
    CombinedComponent,
)
from .blend import Blend  # noqa: F401
from .fitting import fit_blends  # noqa: F401
from .source import (  # noqa: F401
    ExtendedSource,
    SingleExtendedSource,
    MultiExtendedSource,
    CompactExtendedSource,
    PointSource,
)
from .model import Model, UpdateException  # noqa: F401
from . import fft, initialization, measure, operator, synthetic  # noqa: F401
from ._lib import configure  # noqa: F401

__version__ = "0.1.0"
