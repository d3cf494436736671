"""Python mirror of ``smoothField`` (js/climate-util.js:5-25) over the C ABI — the Laplacian smoothing the reference's
climate stages apply to elevation, pressure, continentality, rain-shadow and precipitation fields."""
from __future__ import annotations

import numpy as np

from . import capi
from .terrain_post import Planet, _planet_for


def smooth_field(mesh, field: np.ndarray, passes: int, r_xyz=None, planet: Planet | None = None) -> None:
    """Mutates ``field`` (contiguous float32, numRegions) in place like the reference; returns None."""
    if not (isinstance(field, np.ndarray) and field.dtype == np.float32 and field.flags.c_contiguous):
        raise TypeError("field must be a contiguous float32 array (it is rewritten in place)")
    pl = planet or _planet_for(mesh, r_xyz)
    if field.size != pl.numRegions:
        raise ValueError("field length must equal mesh.numRegions")
    capi.check(capi.lib().wo_smooth_field(pl.handle, capi.ptr(field), int(passes)), "wo_smooth_field")
