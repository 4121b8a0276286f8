"""
``auromat-convert`` on the MI355X path: georeference (and optionally resample) a sequence of frames and write one
CDF or netCDF file per frame — the flag set and the flow of the reference's console script (auromat/cli/convert.py:58-219).

What differs, and why:

* ``--data`` is a directory of frames as ``<id>.wcs`` (an astrometry.net FITS header as 80-column cards, or a ``.json``
  dict of the same cards: ``CRVAL*``, ``CD*``, ``CRPIX*``, ``IMAGEW/H``, ``DATE-OBS`` [+ ``DATESHIF``], ``POS?``
  [+ ``POS?SHIF``]) next to ``<id>.npy`` (uint8 / uint16 array (h, w, 3)) or a developed image file (``<id>.jpg`` /
  ``.png`` / ``.tif`` through Pillow, as the reference's ISS provider caches them).  The reference reads provider archives
  there (ESA ISS ``api.json``, THEMIS ``thg_l1_*`` CDFs): downloads, RAW development and the CDF library are its I/O
  layer, not part of this package — ``--bps``, ``--correctgamma`` and ``--autobright`` belong to that layer and are
  accepted but have nothing to act on.
* ``--format netcdf``: netCDF-4 / HDF5 as the reference's files (:mod:`auromat_amd.export._nc4`) or, with
  ``--netcdf-container classic``, netCDF classic (:mod:`auromat_amd.export._nc3`).  ``--format cdf``: a version-3 CDF laid out
  by :mod:`auromat_amd.export._cdf3` without NASA's CDF library — read that module's header: no CDF library has opened
  these files yet.
* ``--resample`` runs the sequence through the single-pass frame pipeline (:class:`auromat_amd.pipeline.SequencePipeline`,
  grids only: no per-pixel array is ever written or copied to the host): with ``--resolution`` (arcsec / px, the
  reference's flag and default) every frame's px/deg follows from its own bounding box (``plateCarreeResolution``, the
  box-first plan); ``--px-per-deg N`` (an addition) fixes the grid instead.  With several GPUs (``torchrun --nproc-per-node N
  -m auromat_amd.cli.convert``) the frames are sharded over the ranks and every rank writes the files of its own frames.
"""
from __future__ import print_function

import argparse
import os
import sys
from datetime import datetime, timedelta

import numpy as np
import numpy.ma as ma

description = 'Georeference frames on the GPU, optionally resample them, and store them as CDF or netCDF files.'

epilog = '''
Resample on the geomagnetic grid with the reference's resolution rule:
auromat-convert --data frames/ --format netcdf --resample --resolution 80

Resample a long sequence on a fixed 0.1 degree geographic grid, as fast as the GPU goes:
auromat-convert --data frames/ --format netcdf --resample --grid geo --px-per-deg 10

Don't store pixel corner coordinates:
auromat-convert --data frames/ --format netcdf --without-bounds

CDF files (version 3, laid out without NASA's CDF library), MLat/MLT coordinates only:
auromat-convert --data frames/ --format cdf --resample --without-geo
'''


class Grid(object):
    geo = 'geo'
    mag = 'mag'


class Format(object):
    cdf = 'cdf'
    netcdf = 'netcdf'


def date(s):
    return datetime.strptime(s, '%Y-%m-%dT%H:%M:%S')


def getParser():
    parser = argparse.ArgumentParser(prog='auromat-convert', epilog=epilog, description=description,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('--data', help='Data directory, by default the current directory: <id>.wcs (or <id>.json) '
                                       'header files next to <id>.npy (or .jpg / .png / .tif) images.', default=os.getcwd())
    period = parser.add_argument_group('period', 'These arguments optionally specify which data to convert.')
    period.add_argument('--start', help='UTC start date, format 2000-01-01T12:00:00', type=date)
    period.add_argument('--end', help='UTC end date (inclusive)', type=date)
    mappingArgs = parser.add_argument_group('mapping')
    mappingArgs.add_argument('--altitude', help='Altitude in km onto which to map the images, default is 110km',
                             default=110, type=int)
    mappingArgs.add_argument('--exact-centers', dest='exactCenters', action='store_true',
                             help='Cast a ray through every pixel centre (the reference\'s getMapping default) instead of '
                                  'averaging the four corner hits.')
    mappingArgs.add_argument('--min-elevation', dest='minElevation', type=float, default=-1.0,
                             help='Mask pixels seen under a smaller elevation angle before resampling (an addition: the '
                                  'reference\'s auromat-convert resamples the provider\'s mappings unmasked, cli/convert.py:'
                                  '176-185, and so does this driver by default; the user guide recommends 10 degrees); '
                                  'negative = no mask.')
    esaIssArgs = parser.add_argument_group('ESA ISS data (accepted for compatibility; RAW development is not part of this package)')
    esaIssArgs.add_argument('--bps', help='bits per sample, default is 16', choices=[8, 16], default=16, type=int)
    esaIssArgs.add_argument('--correctgamma', action='store_true')
    esaIssArgs.add_argument('--autobright', action='store_true')
    resampleArgs = parser.add_argument_group('resampling')
    resampleArgs.add_argument('--resample', help='Whether to resample or not', action='store_true')
    resampleArgs.add_argument('--resolution', metavar='RES', help='in arcsec/px, default 100', default=100, type=float)
    resampleArgs.add_argument('--px-per-deg', dest='pxPerDeg', metavar='N', type=float,
                              help='fixed grid resolution in pixels per degree instead of --resolution: the sequence then '
                                   'runs through the single-pass frame pipeline')
    resampleArgs.add_argument('--grid', help='The grid which will be regular after resampling. Default is MLat/MLT grid. '
                                             'Use geo for geographical grid.', default=Grid.mag, choices=[Grid.geo, Grid.mag])
    outputArgs = parser.add_argument_group('output')
    outputArgs.add_argument('--out', help='Output directory, by default the "converted" subdirectory of --data')
    outputArgs.add_argument('--overwrite', help='Overwrites existing files.', action='store_true')
    outputArgs.add_argument('--skip', help='Skips already converted files.', action='store_true')
    outputArgs.add_argument('--format', help='Data format of converted files', choices=[Format.cdf, Format.netcdf],
                            required=True)
    outputArgs.add_argument('--netcdf-container', dest='netcdfContainer', choices=['netcdf4', 'classic'], default='netcdf4',
                            help='netCDF files: the netCDF-4 / HDF5 container the reference writes (zlib, one row per chunk; '
                                 'default) or netCDF classic with 64-bit offsets (no compression). Not an option of the '
                                 'reference.')
    outputArgs.add_argument('--without-bounds', dest='withoutBounds', action='store_true',
                            help='Do not include coordinates of pixel corners. If set, then only the pixel center '
                                 'coordinates are written, otherwise both.')
    outputArgs.add_argument('--without-mag', dest='withoutMag', help='Do not include MLat/MLT coordinates.',
                            action='store_true')
    outputArgs.add_argument('--without-geo', dest='withoutGeo', action='store_true',
                            help='Do not include geodetic coordinates. Only usable with CDF output.')
    parser.add_argument('--version', action='version', version='auromat_amd (MI355X)')
    return parser


def parseargs(argv=None):
    parser = getParser()
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 0:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args(argv)
    if not args.out:
        args.out = os.path.join(args.data, 'converted')
    if args.overwrite and args.skip:
        parser.error('only one of --overwrite and --skip is allowed')
    if args.withoutGeo and args.format == Format.netcdf:
        parser.error('--without-geo is only usable with --format cdf')
    if args.format == Format.cdf:
        #  the version-3 container is laid out from the format description; no CDF library has opened such a file yet
        sys.stderr.write('auromat-convert: note: CDF files are written without NASA\'s CDF library and have only been read back by this '
                         'package\'s own reader; validate one with cdflib / pycdf before relying on them\n')
    return args


# ---- input: a directory of headers + arrays ---------------------------------------------------------------------
def read_header(path):
    """A FITS header stored as 80-column ASCII cards (what astrometry.net's .wcs files are) or as JSON -> dict."""
    from ..fits import readHeader
    return readHeader(path)


IMAGE_EXTENSIONS = ('.npy', '.jpg', '.jpeg', '.png', '.tif', '.tiff')


def find_image(data_dir, base):
    """-> path of the image that belongs to header `base` (<base>.npy, or a developed image file as the reference's ISS
    provider keeps next to its .wcs files: <base>.jpg / .png / .tif), None when there is none."""
    for ext in IMAGE_EXTENSIONS:
        for e in (ext, ext.upper()):
            path = os.path.join(data_dir, base + e)
            if os.path.exists(path):
                return path
    return None


def read_image(path, mmap=False):
    """(h, w, 3) uint8 / uint16 array of an image file: .npy as stored, anything else through Pillow (8-bit JPEG / PNG,
    8- or 16-bit TIFF / PNG; grey images are repeated over three channels, an alpha channel is dropped)."""
    if mmap and path.lower().endswith('.npy'):
        return np.load(path, mmap_mode='r')
    from ..util.image import loadImage
    a = loadImage(path)
    if a.dtype not in (np.uint8, np.uint16):
        raise NotImplementedError('unsupported sample type %s in %s' % (a.dtype, path))
    return a


def list_frames(data_dir, start=None, end=None):
    """-> sorted list of (identifier, header dict, image path) within [start, end] by (shifted) photo time."""
    from ..mapping.spacecraft import getPhotoTime
    frames = []
    for name in sorted(os.listdir(data_dir)):
        base, ext = os.path.splitext(name)
        if ext not in ('.wcs', '.json'):
            continue
        img = find_image(data_dir, base)
        if img is None:
            continue
        hdr = read_header(os.path.join(data_dir, name))
        t = getPhotoTime(hdr)
        if t is None:
            continue
        if 'DATESHIF' in hdr:
            t = t + timedelta(seconds=float(hdr['DATESHIF']))
        if (start and t < start) or (end and t > end):
            continue
        frames.append((t, base, hdr, img))
    frames.sort(key=lambda f: f[0])
    return [(b, h, i) for _, b, h, i in frames]


def target_path(args, identifier):
    """-> path to write, or None when the frame is to be skipped; exits like the reference when the file exists."""
    path = os.path.join(args.out, identifier + extension(args))
    if os.path.exists(path):
        if args.skip:
            print('skipping', path)
            return None
        elif args.overwrite:
            os.remove(path)
        else:
            print('The file', path, 'already exists.\nPlease use --skip or --overwrite, or a different output folder.',
                  file=sys.stderr)
            sys.exit(1)
    return path


def extension(args):
    return '.cdf' if args.format == Format.cdf else '.nc'


# ---- the two ways through the GPU ----------------------------------------------------------------------------
def convert_with_classes(args, frames, export):
    """Frame by frame through the mapping classes: the reference's flow, its resolution rule included."""
    from ..mapping.spacecraft import getMapping
    from ..resample import resample, resampleMLatMLT
    for identifier, hdr, img_path in frames:
        path = target_path(args, identifier)
        if path is None:
            continue
        mapping = getMapping(read_image(img_path), hdr, altitude=args.altitude,
                             fastCenterCalculation=not args.exactCenters, identifier=identifier)
        if args.resample:
            if args.minElevation >= 0:
                mapping = mapping.maskedByElevation(args.minElevation)
            fn = resample if args.grid == Grid.geo else resampleMLatMLT
            mapping = fn(mapping, arcsecPerPx=args.resolution)
        print('storing', path)
        export(path, mapping)


def grid_mapping(res, cam, t, altitude, identifier, magnetic):
    """The resampled grid of one frame (host arrays of the few hundred KB the pipeline returns) as the mapping that
    ``resample`` / ``resampleMLatMLT`` would have returned: a GenericMapping in geodetic coordinates; a grid that is
    regular in (MLat, SM longitude) goes through SM -> GEO like ``convertSMMappingToGeo`` (reference
    mapping.py:1549-1559)."""
    from ..coordinates.transform import smToLatLon
    from ..mapping.mapping import GenericMapping
    lat, lon, lat_c, lon_c = res['lat'], res['lon'], res['lat_c'], res['lon_c']
    if magnetic:
        lat, lon = smToLatLon(lat, lon, t)
        lat_c, lon_c = smToLatLon(lat_c, lon_c, t)
    img = ma.masked_array(res['img'], mask=np.repeat(res['mask'][:, :, None], res['img'].shape[2], 2))
    return GenericMapping(lat, lon, lat_c, lon_c, ma.masked_invalid(res['mean'][:, :, -1]), altitude, img, cam, t, identifier)


def host_snapshot(mapping, with_mag):
    """What the exporters read of a mapping, evaluated HERE (the device work of a mapping — MLat/MLT of the grid, the traced
    outline behind the bounding box — stays on the thread that owns the GPU context) and held as plain host arrays: such a
    snapshot can be written by another thread while the next frames are processed."""
    from types import SimpleNamespace
    snap = SimpleNamespace(lats=mapping.lats, lons=mapping.lons, latsCenter=mapping.latsCenter, lonsCenter=mapping.lonsCenter,
                           elevation=mapping.elevation, img=mapping.img, boundingBox=mapping.boundingBox,
                           photoTime=mapping.photoTime, altitude=mapping.altitude, cameraPosGCRS=mapping.cameraPosGCRS,
                           metadata=mapping.metadata, identifier=mapping.identifier)
    if with_mag:
        snap.mLatMlt, snap.mLatMltCenter = mapping.mLatMlt, mapping.mLatMltCenter
    return snap


def convert_with_pipeline(args, frames, export):
    """The whole sequence through the single-pass frame pipeline — at a fixed px/deg (--px-per-deg) or, the reference's
    flags, at --resolution arcsec per pixel, where every frame's px/deg follows from its own bounding box (the box-first plan
    of SequencePipeline(arcsecPerPx=...)) —; this rank's share of the frames."""
    import torch
    import torch.distributed as dist
    from ..mapping.spacecraft import frame_inputs
    from ..pipeline import SequencePipeline
    from ..resample import grid_coordinates
    from ..sequence import shard
    from .._native import to_host
    distributed = 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1
    rank, world = 0, 1
    if distributed:
        rank, world = int(os.environ.get('RANK', '0')), int(os.environ['WORLD_SIZE'])
        # the "file exists, neither --skip nor --overwrite" exit of the reference is decided for ALL frames on EVERY rank
        # before the process group exists: every rank leaves together and nobody waits in a collective for a rank that
        # has gone (a check per rank's own share left the others in dist.barrier() until the launcher killed them)
        if not args.skip and not args.overwrite:
            for identifier, _, _ in frames:
                path = os.path.join(args.out, identifier + extension(args))
                if os.path.exists(path):
                    print('The file', path, 'already exists.\nPlease use --skip or --overwrite, or a different output folder.',
                          file=sys.stderr)
                    sys.exit(1)
        # (AMT_CONVERT_ONE_GPU=1 AMT_CONVERT_BACKEND=gloo: several ranks on one GPU, for rehearsals — the ranks only meet in the
        # closing barrier)
        torch.cuda.set_device(0 if os.environ.get('AMT_CONVERT_ONE_GPU') else int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(os.environ.get('AMT_CONVERT_BACKEND', 'nccl'))
        rank, world = dist.get_rank(), dist.get_world_size()
    todo = []
    for identifier, hdr, img_path in [frames[k] for k in shard(len(frames), rank, world)]:
        path = target_path(args, identifier)
        if path is not None:
            todo.append((identifier, hdr, img_path, path))
    if todo:
        first = read_image(todo[0][2], mmap=True)
        magnetic = args.grid == Grid.mag
        seq = SequencePipeline(first.shape[1], first.shape[0], nchan=first.shape[2], img_dtype=first.dtype,
                               altitude=args.altitude, fast=not args.exactCenters,
                               min_elevation=args.minElevation if args.minElevation >= 0 else None,
                               pxPerDeg=args.pxPerDeg or 10, magnetic=magnetic, keep_coordinates=False,
                               arcsecPerPx=None if args.pxPerDeg else args.resolution)

        def feed():
            # decoding a 12 Mpx JPEG takes ~100 ms of host time, the GPU 0.2 ms per frame: images are read ahead on a
            # few threads (Pillow and NumPy release the GIL while they decode / load)
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            ahead = max(1, int(os.environ.get('AMT_CONVERT_READ_AHEAD', '8')))
            from .._native import host_threads
            with ThreadPoolExecutor(max_workers=host_threads(min(ahead, 8))) as pool:
                pending = deque()
                it = iter(todo)
                for item in it:
                    pending.append((item, pool.submit(read_image, item[2])))
                    if len(pending) >= ahead:
                        break
                while pending:
                    (identifier, hdr, img_path, path), fut = pending.popleft()
                    nxt = next(it, None)
                    if nxt is not None:
                        pending.append((nxt, pool.submit(read_image, nxt[2])))
                    cam, t = frame_inputs(hdr)
                    yield hdr, cam, t, fut.result()

        metas = [frame_inputs(hdr) for _, hdr, _, _ in todo]
        results = seq.process(feed(), keep_on_device=True)
        # the files are written by a few threads (deflate releases the interpreter lock; AMT_CONVERT_WRITERS=0: in line): a
        # resampled grid takes 6 ms to compute and 20-odd to compress and lay out
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        from .._native import host_threads
        n_writers = max(0, int(os.environ.get('AMT_CONVERT_WRITERS', '4')))
        n_writers = host_threads(n_writers) if n_writers else 0          # (several ranks on one host share its cores)
        writers = ThreadPoolExecutor(max_workers=n_writers) if n_writers else None
        written = deque()
        try:
            for k, ((identifier, hdr, img_path, path), (cam, t), res) in enumerate(zip(todo, metas, results)):
                if res is None:
                    why = 'a pole in view: --resolution yields no longitude resolution for' if \
                        seq.plans[k] == 'pole-without-resolution' else 'no valid pixel in'
                    print(why, identifier, file=sys.stderr)
                    continue
                host = dict(res)
                host.update(grid_coordinates(res))
                host.update(mean=to_host(res['mean']), img=to_host(res['img'], dtype=first.dtype), mask=to_host(res['mask']).astype(bool))
                print('storing', path)
                mapping = grid_mapping(host, cam, t, args.altitude, identifier, magnetic)
                if writers is None:
                    export(path, mapping)
                    continue
                written.append(writers.submit(export, path, host_snapshot(mapping, not args.withoutMag)))
                while len(written) > 2 * n_writers:
                    written.popleft().result()
            while written:
                written.popleft().result()                # (an exporter's exception surfaces here)
        finally:
            if writers is not None:
                writers.shutdown(wait=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    args = parseargs(argv)
    from functools import partial
    frames = list_frames(args.data, args.start, args.end)
    if not frames:
        raise NotImplementedError('No <id>.wcs / <id>.json + <id>.npy (or .jpg / .png / .tif) frames found in ' + args.data)
    if args.format == Format.cdf:
        from ..export.cdf import write
        export = partial(write, includeBounds=not args.withoutBounds, includeMagCoords=not args.withoutMag,
                         includeGeoCoords=not args.withoutGeo)
    else:
        from ..export.netcdf import write
        export = partial(write, includeBounds=not args.withoutBounds, includeMagCoords=not args.withoutMag,
                         includeGeoCoords=not args.withoutGeo,
                         format='NETCDF4' if args.netcdfContainer == 'netcdf4' else 'NETCDF3_64BIT')
    os.makedirs(args.out, exist_ok=True)
    # --resample, with the reference's --resolution (default 100 arcsec per pixel) or with --px-per-deg: the sequence pipeline;
    # without --resample the frames are exported as they are, through the mapping classes (AMT_CONVERT_CLASSES=1 sends
    # resampling runs that way too: the reference's flow frame by frame, for A/B runs)
    if args.resample and not os.environ.get('AMT_CONVERT_CLASSES'):
        convert_with_pipeline(args, frames, export)
    else:
        convert_with_classes(args, frames, export)
    print('Done.')


if __name__ == '__main__':
    main()
