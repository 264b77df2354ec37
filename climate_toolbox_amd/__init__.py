"""MI355X-native weighted grid->region aggregation engine: a drop-in for the aggregation path of
ClimateImpactLab/climate_toolbox (``climate_toolbox.aggregations``).  See DESIGN.md."""

__version__ = "0.1.0"

from .aggregations import (  # noqa: F401
    weighted_aggregate_grid_to_regions,
    prepare_spatial_weights_data,
    prepare_weights,
    PreparedWeights,
    clear_caches,
    results_on_device,
    _reindex_spatial_data_to_regions,
    _aggregate_reindexed_data_to_regions,
)
from .standardize import (  # noqa: F401  (coordinate standardisation folded into the plan)
    standardize_climate_data,
    convert_lons_split,
    convert_lons_mono,
    rename_coords_to_lon_and_lat,
)
from .transformations import (  # noqa: F401  (tas_poly fused into the aggregation)
    tas_poly,
    tas_poly_aggregate,
    snyder_edd,
    snyder_gdd,
)
