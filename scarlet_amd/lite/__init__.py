"""``scarlet.lite`` on the GPU: same names as the reference's lite package
(scarlet/lite/__init__.py), the fitting loop of ``LiteBlend.fit`` on the device."""

from .initialization import (  # noqa: F401
    get_min_psf,
    init_adaprox_component,
    init_all_sources_main,
    init_fista_component,
    init_main_parameters,
    init_monotonic_morph,
    multifit_seds,
    parameterize_sources,
)
from .measure import calculate_snr, weight_sources  # noqa: F401
from .models import (  # noqa: F401
    LiteBlend,
    LiteComponent,
    LiteFactorizedComponent,
    LiteObservation,
    LiteSource,
)
from .parameters import (  # noqa: F401
    AdaproxParameter,
    FistaParameter,
    LiteParameter,
    grow_array,
)
from .utils import (  # noqa: F401
    bounds_to_bbox,
    get_circle_mask,
    insert_image,
    integrated_circular_gaussian,
    integrated_gaussian,
    project_morph_to_center,
)
