"""MI355X-native multispectral index path (white balance -> NDVI/GNDVI/NDWI -> statistics).

Drop-in for the per-pixel path of lars-uav/lars-image-processing: the functions
below keep the reference's signatures and run on hand-written HIP kernels
(gfx950) behind a C ABI (include/lars_hip.h).  No PyTorch, no NumPy fallback.
"""
from .api import (  # noqa: F401
    align_images,
    analyze_index,
    analyze_index_statistics,
    analyze_ndvi_statistics,
    calculate_index,
    calculate_index_statistics_by_timeframe,
    calculate_ndvi,
    calculate_ndvi_array,
    change_detection,
    classification_mask,
    colorize_difference,
    colorize_index,
    colormap_lut,
    correct_white_balance,
    download_processed_images,
    fix_white_balance,
    fix_white_balance_rgnir,
    generate_ndvi_report,
    index_histogram,
    preprocess_large_image,
    process_image,
    time_series_points,
    timeseries_row,
)
from .batch import TileBatch, local_fold, merge_records, shard_range, summarize, timeseries_rows  # noqa: F401
from .tiffio import read_image, read_tiff, write_tiff  # noqa: F401

__version__ = "0.1.0"
