"""Directory driver: decode -> white balance -> indices -> files.

Mirrors ``backend-process.py:49-97`` (``process_image`` / ``batch_process``): same
directory layout (``white_balanced/<name>_wb.tif``, ``<INDEX>/<name>_<index>.png``), same
extension filter, same per-file ``try``/``print`` error policy -- but each image crosses PCIe
once (``lars_h_process_image``: white balance, all requested indices and their colormaps in one
upload) and decoding / encoding overlap the GPU work on a thread pool (PIL releases the GIL;
the library gives every thread its own stream and workspace).

The index pictures are the per-pixel RGBA images of the reference's colormaps at full resolution (RdYlGn / RdYlBu,
``vmin=-1, vmax=1``), produced by the fused kernel.  The reference's matplotlib figure (imshow + colorbar,
``backend-process.py:40-47``) is figure rendering and not part of this package.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

from . import api

EXTENSIONS = {".tif", ".tiff", ".png", ".jpg", ".jpeg"}      # backend-process.py:89
# zlib level of the per-pixel colormap PNGs: encoding, not the GPU, bounds the directory driver (a 2048x2048 RGBA map takes
# 1.2 s at Pillow's default level 6 and 0.6 s at level 1, for a smaller file on colormap images); same pixels either way
LUT_PNG_LEVEL = 1


def process_image(image_path, output_dir, process_wb=False, indices=None, full_depth=False, lut_format="png"):
    """One file: same outputs as backend-process.py:49-73.  Returns the statistics dicts.
    ``full_depth=True`` reads three-sample 16-bit TIFFs at their full depth (``tiffio.read_image``; Pillow, hence the
    reference, keeps their high bytes only).  ``lut_format="tiff"`` writes the colormap images as
    uncompressed RGBA TIFFs ``<name>_<index>.tif`` instead of PNGs: PNG compression of a 4096 x 4096 map takes seconds,
    the GPU work milliseconds.  ``lut_format="png8"`` writes palette PNGs (one byte per pixel = the colormap entry, the
    colormap as the palette): ``Image.open(p).convert("RGBA")`` gives the pixels of the RGBA file, from a quarter of the
    bytes to compress."""
    if lut_format not in ("png", "png8", "tiff"):
        raise ValueError(f"lut_format must be 'png', 'png8' or 'tiff', got {lut_format!r}")
    from PIL import Image
    from .tiffio import read_image
    image_path, output_dir = Path(image_path), Path(output_dir)
    name = image_path.stem
    arr = read_image(image_path, full_depth)
    if arr.ndim != 3 or arr.shape[2] < 3:
        raise ValueError(f"{image_path.name}: expected an image with at least 3 channels, got shape {arr.shape}")
    indices = list(indices or [])
    palette = lut_format == "png8"
    # palette files: the colormap entry of every pixel comes from the device, one byte per pixel (lars_h_process_image)
    res = api.process_image(arr, indices=indices, white_balance=True, want_arrays=False, want_rgba=not palette,
                            want_entries=palette) if indices else None
    corrected = res["corrected"] if res else api.fix_white_balance(arr)
    if process_wb:
        (output_dir / "white_balanced").mkdir(parents=True, exist_ok=True)
        Image.fromarray(corrected[:, :, :3] if corrected.shape[2] > 4 else corrected).save(
            output_dir / "white_balanced" / f"{name}_wb.tif")
    stats = {}
    for t in indices:
        (output_dir / t).mkdir(parents=True, exist_ok=True)
        out = output_dir / t / f"{name}_{t.lower()}.png"
        entry = res["indices"][t]
        if lut_format == "tiff":
            from .tiffio import write_tiff
            write_tiff(out.with_suffix(".tif"), entry["rgba"])
        elif palette:
            im = Image.fromarray(entry["entry"], "P")
            im.putpalette(api.colormap_lut(api._colormap_for(t)).tobytes(), rawmode="RGBA")
            im.save(out, compress_level=LUT_PNG_LEVEL)
        else:
            Image.fromarray(entry["rgba"], "RGBA").save(out, compress_level=LUT_PNG_LEVEL)
        stats[t] = entry["stats"]
    return stats


def files_of_rank(files, rank=0, world=1):
    """The files one rank of a multi-GPU run owns: a contiguous block of the sorted list (``batch.shard_range``, the split the
    tile batches use).  Files are independent (backend-process.py:92-95 loops over them one by one), so there is no exchange."""
    from .batch import shard_range
    rank, world = int(rank), int(world)
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of {world}")
    files = list(files)
    start, stop = shard_range(len(files), rank, world)
    return files[start:stop]


def batch_process(input_dir, output_dir, process_wb=False, process_ndvi=False, process_gndvi=False,
                  process_ndwi=True, workers=4, verbose=True, full_depth=False, lut_format="png", rank=0, world=1,
                  device=None):
    """backend-process.py:75-97 with its module constants as arguments.  Returns ``{file name: stats | error}``.
    ``rank`` / ``world``: one process per GPU, each takes its block of the sorted file list (``files_of_rank``); the output
    directories are shared, the file names distinct.  ``device``: the GPU ordinal every worker thread binds (the library's
    context is per thread and defaults to device 0); None leaves the threads' binding alone."""
    input_path, output_path = Path(input_dir), Path(output_dir)
    indices = [t for t, on in (("NDVI", process_ndvi), ("GNDVI", process_gndvi), ("NDWI", process_ndwi)) if on]
    files = files_of_rank(sorted(f for f in input_path.glob("*") if f.suffix.lower() in EXTENSIONS), rank, world)
    total = len(files)
    results = {}

    def one(job):
        idx, f = job
        try:
            if verbose:
                print(f"Processing {idx}/{total}: {f.name}")
            return f.name, process_image(f, output_path, process_wb, indices or None, full_depth, lut_format)
        except Exception as e:                              # same policy as upstream :96-97
            if verbose:
                print(f"Error processing {f.name}: {str(e)}")
            return f.name, e

    def bind():
        if device is not None:
            from . import _ffi
            _ffi.call("lars_set_device", int(device))

    if workers <= 1:
        bind()
        for job in enumerate(files, 1):
            k, v = one(job)
            results[k] = v
    else:
        with ThreadPoolExecutor(max_workers=min(workers, os.cpu_count() or 1), initializer=bind) as pool:
            for k, v in pool.map(one, enumerate(files, 1)):
                results[k] = v
    return results


def export_zip(image_array, selected_indices, corrected_array=None):
    """ZIP of the processed images of one upload (``download_processed_images``,
    process-images.py:567-617): ``white_balanced.png`` + ``<INDEX>_visualization.png`` per index, the latter as
    full-resolution per-pixel colormap images (one GPU pass for white balance, every index and every colormap).
    ``corrected_array`` (the cached white-balanced image the UI keeps, process-images.py:1132) skips the
    white-balance step.
    """
    import io
    import zipfile
    from PIL import Image
    indices = list(selected_indices)
    if corrected_array is not None:
        res = api.process_image(np.asarray(corrected_array), indices=indices, white_balance=False, want_arrays=False, want_rgba=True)
        corrected = np.asarray(corrected_array)
    else:
        res = api.process_image(np.asarray(image_array), indices=indices, white_balance=True, want_arrays=False, want_rgba=True)
        corrected = res["corrected"]
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", zipfile.ZIP_DEFLATED) as zf:
        png = io.BytesIO()
        Image.fromarray(corrected).save(png, format="PNG")
        zf.writestr("white_balanced.png", png.getvalue())
        for t in indices:
            png = io.BytesIO()
            Image.fromarray(res["indices"][t]["rgba"], "RGBA").save(png, format="PNG", compress_level=LUT_PNG_LEVEL)
            zf.writestr(f"{t}_visualization.png", png.getvalue())
    return buf.getvalue()


def main(argv=None):
    """``python -m lars_image_processing_amd.driver IN OUT [--wb] [--ndvi] [--gndvi] [--no-ndwi] ...``: backend-process.py's
    ``__main__`` with its constants as flags.  Under a launcher (``python -m torch.distributed.run --nproc-per-node N -m
    lars_image_processing_amd.driver ...``) every rank binds GPU ``LOCAL_RANK`` and processes its block of the files."""
    import argparse
    import json
    ap = argparse.ArgumentParser(prog="lars_image_processing_amd.driver")
    ap.add_argument("input_dir")
    ap.add_argument("output_dir")
    ap.add_argument("--wb", action="store_true", help="also write white_balanced/<name>_wb.tif (PROCESS_WB)")
    ap.add_argument("--ndvi", action="store_true")
    ap.add_argument("--gndvi", action="store_true")
    ap.add_argument("--no-ndwi", dest="ndwi", action="store_false", help="backend-process.py:12-15 has only NDWI on")
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--full-depth", action="store_true")
    ap.add_argument("--lut-format", default="png", choices=["png", "png8", "tiff"])
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    from .dist import env_rank_world
    rank, local_rank, world = env_rank_world()
    res = batch_process(args.input_dir, args.output_dir, args.wb, args.ndvi, args.gndvi, args.ndwi, args.workers,
                        not args.quiet, args.full_depth, args.lut_format, rank, world,
                        device=local_rank if world > 1 else None)
    failed = {k: str(v) for k, v in res.items() if isinstance(v, Exception)}
    print(json.dumps({"rank": rank, "world": world, "files": len(res), "failed": failed}))
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
