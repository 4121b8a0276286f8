"""
The convert driver (auromat_amd/cli/convert.py): the reference's flag set (cli/convert.py:58-130) and its checks on
the CPU; on the GPU a directory of frames through both routes — mapping classes with the reference's arcsec/px rule,
and the single-pass sequence pipeline at a fixed px/deg — into netCDF files that read back as the same grids.
"""
import json
import os

import numpy as np
import numpy.ma as ma
import pytest

from conftest import ROOT


def write_frames(tmp_path, n=3, w=256, h=170):
    from auromat_amd.synthetic import frame_image, sequence_frame
    d = tmp_path / 'frames'
    d.mkdir()
    for k in range(n):
        hdr, cam, t, seed = sequence_frame(k, w, h)
        hdr = dict(hdr)
        hdr.update({'DATE-OBS': t.strftime('%Y-%m-%dT%H:%M:%S.%f'), 'POSX': float(cam[0]), 'POSY': float(cam[1]),
                    'POSZ': float(cam[2])})
        if k == 0:
            # a real .wcs-style file: 80-column cards
            cards = []
            for key, v in hdr.items():
                val = ("'%s'" % v).ljust(20) if isinstance(v, str) else ('%20s' % repr(v))
                cards.append(('%-8s= %s' % (key, val)).ljust(80))
            cards.append('END'.ljust(80))
            (d / ('frame%02d.wcs' % k)).write_text(''.join(cards))
        else:
            (d / ('frame%02d.json' % k)).write_text(json.dumps(hdr))
        np.save(str(d / ('frame%02d.npy' % k)), frame_image(w, h, seed=seed))
    return str(d)


def test_flags_are_the_references():
    from auromat_amd.cli.convert import getParser, parseargs
    opts = {a.option_strings[0] for a in getParser()._actions if a.option_strings}
    for flag in ('--data', '--start', '--end', '--altitude', '--bps', '--correctgamma', '--autobright', '--resample',
                 '--resolution', '--grid', '--out', '--overwrite', '--skip', '--format', '--without-bounds', '--without-mag',
                 '--without-geo', '--version'):
        assert flag in opts, flag
    a = parseargs(['--data', '/x', '--format', 'netcdf'])
    assert a.out == '/x/converted' and a.altitude == 110 and a.resolution == 100 and a.grid == 'mag' and not a.resample
    assert parseargs(['--data', '/x', '--format', 'cdf', '--without-geo']).withoutGeo
    for bad in (['--data', '/x', '--format', 'netcdf', '--overwrite', '--skip'], ['--data', '/x', '--format', 'netcdf', '--without-geo'],
                ['--data', '/x', '--format', 'hdf'], ['--data', '/x']):
        with pytest.raises(SystemExit):
            parseargs(bad)
    with pytest.raises(SystemExit):
        parseargs([])


def test_header_cards_and_frame_listing(tmp_path):
    from datetime import datetime
    from auromat_amd.cli.convert import list_frames, read_header
    d = write_frames(tmp_path)
    hdr = read_header(os.path.join(d, 'frame00.wcs'))
    ref = json.load(open(os.path.join(d, 'frame01.json')))
    assert set(hdr) == set(ref) and hdr['CTYPE1'] == 'RA---TAN' and hdr['IMAGEW'] == 256
    assert hdr['CD1_1'] == ref['CD1_1'] and isinstance(hdr['CRPIX1'], float)
    frames = list_frames(d)
    assert [f[0] for f in frames] == ['frame00', 'frame01', 'frame02']
    assert [f[0] for f in list_frames(d, start=datetime(2012, 1, 25, 9, 26, 56))] == ['frame01', 'frame02']


@pytest.mark.gpu
def test_convert_both_routes(tmp_path, capsys):
    from auromat_amd.cli.convert import main
    from auromat_amd.export import _nc4
    from auromat_amd.mapping.netcdf import NetCDFMapping, read_arrays
    from auromat_amd.mapping.spacecraft import getMapping
    from auromat_amd.resample import resample, resampleMLatMLT
    d = write_frames(tmp_path)
    # 1. the reference's route: arcsec / px from each frame's box, MLat/MLT grid by default
    out1 = str(tmp_path / 'o1')
    main(['--data', d, '--format', 'netcdf', '--resample', '--min-elevation', '10', '--resolution', '900', '--out', out1])
    assert sorted(os.listdir(out1)) == ['frame00.nc', 'frame01.nc', 'frame02.nc']
    hdr = json.load(open(os.path.join(d, 'frame01.json')))
    m = getMapping(np.load(os.path.join(d, 'frame01.npy')), hdr, fastCenterCalculation=True,
                   identifier='frame01').maskedByElevation(10)
    want = resampleMLatMLT(m, arcsecPerPx=900)
    got = read_arrays(os.path.join(out1, 'frame01.nc'))
    assert np.array_equal(got['lats'].filled(np.nan), want.lats.filled(np.nan), equal_nan=True)
    assert np.array_equal(got['img'].filled(0), want.img.filled(0)) and np.array_equal(ma.getmaskarray(got['img']), ma.getmaskarray(want.img))
    assert np.allclose(got['elevation'].filled(-1), want.elevation.filled(-1), atol=1e-4)       # stored as float32 zenith angle
    f = _nc4.open_file(os.path.join(out1, 'frame01.nc'))
    # (the geodetic coordinates of an MLat/MLT grid are curvilinear; and the MLat/MLT the exporter recomputes from them at the
    # mapping altitude are not exactly regular either, in the reference as here: smToLatLon goes through a point 1 km
    # from the Earth's centre, transform.py:472-480 — so both systems are stored as 2-D arrays with cell bounds)
    assert f.vars['mlat'].dims == ('y', 'x') and f.vars['lat'].dims == ('y', 'x') and f.vars['mlat_bounds'].dims == ('y', 'x', 'vertex4')
    # the file reads back as a mapping that satisfies the class guarantees
    back = NetCDFMapping(os.path.join(out1, 'frame01.nc'))
    back.checkGuarantees()
    assert back.identifier == 'frame01' and back.altitude == 110 and back.photoTime == m.photoTime
    # re-running refuses, --skip skips, --overwrite overwrites
    with pytest.raises(SystemExit):
        main(['--data', d, '--format', 'netcdf', '--resample', '--min-elevation', '10', '--resolution', '900', '--out', out1])
    main(['--data', d, '--format', 'netcdf', '--resample', '--min-elevation', '10', '--resolution', '900', '--out', out1, '--skip'])
    assert 'skipping' in capsys.readouterr().out
    # 2. the fast route: fixed px/deg through the single-pass sequence pipeline, geographic grid, no bounds
    out2 = str(tmp_path / 'o2')
    main(['--data', d, '--format', 'netcdf', '--resample', '--min-elevation', '10', '--grid', 'geo', '--px-per-deg', '5', '--out', out2, '--without-bounds',
          '--without-mag'])
    want = resample(m, pxPerDeg=5)
    f = _nc4.open_file(os.path.join(out2, 'frame01.nc'))
    assert f.vars['lat'].dims == ('lats',) and 'lat_bounds' not in f.vars and 'mlat' not in f.vars
    assert np.array_equal(f.vars['lat'].data, want.latsCenter.data[:, 0]) and np.array_equal(f.vars['lon'].data, want.lonsCenter.data[0, :])
    red = ma.masked_equal(f.vars['img_red'].data, f.vars['img_red'].attrs['_FillValue'])
    assert np.array_equal(ma.getmaskarray(red), ma.getmaskarray(want.img)[:, :, 0])
    assert np.array_equal(red.filled(0), want.img[:, :, 0].filled(0).astype(np.int32))
    # 3. without --resample the unresampled mapping goes out (2D coordinates with cell bounds)
    out3 = str(tmp_path / 'o3')
    main(['--data', d, '--format', 'netcdf', '--out', out3, '--end', '2012-01-25T09:26:56'])
    assert os.listdir(out3) == ['frame00.nc']
    f = _nc4.open_file(os.path.join(out3, 'frame00.nc'))
    assert f.vars['lat'].dims == ('y', 'x') and f.vars['lat_bounds'].data.shape == (170, 256, 4)


@pytest.mark.gpu
def test_convert_to_cdf(tmp_path):
    """--format cdf (the format of every example in the reference's own help text, cli/convert.py:35-44): the same mappings as
    --format netcdf, in version-3 CDF files (container: see export/_cdf3.py's header — unpinned)."""
    from auromat_amd.cli.convert import main
    from auromat_amd.export import _cdf3
    from auromat_amd.mapping.cdf import CDFMapping
    from auromat_amd.mapping.netcdf import read_arrays
    d = write_frames(tmp_path)
    flags = ['--data', d, '--resample', '--min-elevation', '10', '--resolution', '900']
    out_nc, out_cdf, out_mag = str(tmp_path / 'nc'), str(tmp_path / 'cdf'), str(tmp_path / 'mag')
    main(flags + ['--format', 'netcdf', '--out', out_nc])
    main(flags + ['--format', 'cdf', '--out', out_cdf])
    assert sorted(os.listdir(out_cdf)) == ['frame00.cdf', 'frame01.cdf', 'frame02.cdf']
    for k in range(3):
        want = read_arrays(os.path.join(out_nc, 'frame%02d.nc' % k))
        got = CDFMapping(os.path.join(out_cdf, 'frame%02d.cdf' % k))
        got.checkGuarantees()
        for key in ('lats', 'lons', 'latsCenter', 'lonsCenter'):
            assert np.array_equal(getattr(got, key).filled(np.nan), want[key].filled(np.nan), equal_nan=True), key
        # (both store a float32 zenith angle: netCDF float32(90 - elevation), CDF 90 - float32(elevation), as the reference's two)
        assert np.allclose(got.elevation.filled(-1), want['elevation'].filled(-1), atol=1e-4)
        assert np.array_equal(got.img.filled(0), want['img'].filled(0)) and np.array_equal(ma.getmaskarray(got.img), ma.getmaskarray(want['img']))
        assert got.photoTime == want['photoTime'] and got.altitude == want['altitude'] and got.identifier == 'frame%02d' % k
    # --without-geo (CDF only): MLat/MLT coordinates without the geodetic ones; re-running refuses as for netCDF
    main(flags + ['--format', 'cdf', '--out', out_mag, '--without-geo', '--without-bounds'])
    r = _cdf3.Reader(os.path.join(out_mag, 'frame01.cdf'))
    assert 'lat' not in r and 'mlat' in r and 'mlat_bounds' not in r and r['mlat'].compressed == 5
    with pytest.raises(SystemExit):
        main(flags + ['--format', 'cdf', '--out', out_mag, '--without-geo', '--without-bounds'])
    # the unresampled mapping of a frame
    out_raw = str(tmp_path / 'raw')
    main(['--data', d, '--format', 'cdf', '--out', out_raw, '--end', '2012-01-25T09:26:56'])
    r = _cdf3.Reader(os.path.join(out_raw, 'frame00.cdf'))
    assert r['lat'].dims == (170, 256) and r['lat_bounds'].dims == (171, 257) and r['img_red'].type in (_cdf3.CDF_INT4, _cdf3.CDF_UINT2)


def test_image_files_next_to_the_header(tmp_path):
    """<id>.png / .tif / .jpg beside <id>.wcs (what the reference's ISS provider caches) are read through Pillow."""
    PIL = pytest.importorskip('PIL')
    from PIL import Image
    from auromat_amd.cli.convert import find_image, list_frames, read_image
    rs = np.random.RandomState(3)
    rgb8 = rs.randint(0, 255, (20, 30, 3)).astype(np.uint8)
    grey16 = rs.randint(0, 65535, (20, 30)).astype(np.uint16)
    d = str(tmp_path)
    Image.fromarray(rgb8).save(os.path.join(d, 'a.png'))
    Image.fromarray(rgb8).save(os.path.join(d, 'b.tif'))
    Image.fromarray(grey16).save(os.path.join(d, 'c.tif'))
    smooth = np.dstack([np.add.outer(np.arange(20) * 5, np.arange(30) * 4 + 20 * c) for c in range(3)]).astype(np.uint8)
    Image.fromarray(smooth).save(os.path.join(d, 'e.JPG'), quality=95)
    np.save(os.path.join(d, 'f.npy'), rgb8)
    for base, want in (('a', rgb8), ('b', rgb8), ('f', rgb8)):
        got = read_image(find_image(d, base))
        assert got.dtype == np.uint8 and np.array_equal(got, want), base
    got = read_image(find_image(d, 'c'))
    assert got.dtype == np.uint16 and got.shape == (20, 30, 3) and np.array_equal(got[..., 1], grey16)
    got = read_image(find_image(d, 'e'))
    assert got.shape == (20, 30, 3) and got.dtype == np.uint8 and np.abs(got.astype(int) - smooth).mean() < 3     # lossy
    assert find_image(d, 'nothing') is None
    # a header next to a .png is a frame
    from auromat_amd.synthetic import sequence_frame
    hdr, cam, t, _ = sequence_frame(0, 30, 20)
    hdr = dict(hdr)
    hdr.update({'DATE-OBS': t.strftime('%Y-%m-%dT%H:%M:%S.%f'), 'POSX': float(cam[0]), 'POSY': float(cam[1]),
                'POSZ': float(cam[2])})
    with open(os.path.join(d, 'a.json'), 'w') as fp:
        json.dump(hdr, fp)
    frames = list_frames(d)
    assert [f[0] for f in frames] == ['a'] and frames[0][2].endswith('a.png')


@pytest.mark.gpu
@pytest.mark.parametrize('resolution_flags', [['--grid', 'geo', '--px-per-deg', '8'], ['--resolution', '600'], []],
                         ids=['px-per-deg', 'resolution', 'reference-defaults'])
def test_convert_under_a_launcher_with_two_ranks(tmp_path, resolution_flags):
    """`python -m torch.distributed.run --nproc-per-node 2 -m auromat_amd.cli.convert ... --resample [--px-per-deg 8 | --resolution R |
    nothing: the reference's defaults, 100 arcsec per pixel on the MLat/MLT grid]`: every rank converts and
    writes its share of the frames (rehearsed on the one GPU: both ranks on cuda:0, the closing barrier through gloo); the files
    equal those of a single process — which, for --resolution, is the sequence pipeline's box-first plan and equals the mapping
    classes frame by frame (AMT_CONVERT_CLASSES=1) —; a second run without --skip / --overwrite leaves on every rank before the
    group exists."""
    import socket
    import subprocess
    import sys
    from auromat_amd.cli.convert import main
    from auromat_amd.export import _nc4
    d = write_frames(tmp_path)
    single, multi = str(tmp_path / 'single'), str(tmp_path / 'multi')
    flags = ['--data', d, '--format', 'netcdf', '--resample', '--min-elevation', '10'] + resolution_flags
    main(flags + ['--out', single])
    if '--px-per-deg' not in resolution_flags:
        from auromat_amd.pipeline import SequencePipeline       # the resolution per frame came from the box-first plan
        classes = str(tmp_path / 'classes')
        os.environ['AMT_CONVERT_CLASSES'] = '1'
        try:
            main(flags + ['--out', classes])
        finally:
            del os.environ['AMT_CONVERT_CLASSES']
        for name in sorted(os.listdir(classes)):
            a, b = _nc4.open_file(os.path.join(single, name)), _nc4.open_file(os.path.join(classes, name))
            assert list(a.vars) == list(b.vars)
            for k, v in a.vars.items():
                if k == 'zenith_angle':
                    assert np.allclose(v.data, b.vars[k].data, atol=1e-4, equal_nan=True), (name, k)
                else:
                    assert np.array_equal(v.data, b.vars[k].data, equal_nan=True), (name, k)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(AMT_CONVERT_ONE_GPU='1', AMT_CONVERT_BACKEND='gloo', PYTHONPATH=os.pathsep.join([ROOT, env.get('PYTHONPATH', '')]))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), '-m', 'auromat_amd.cli.convert'] + flags + ['--out', multi]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    assert sorted(os.listdir(multi)) == sorted(os.listdir(single)) == ['frame00.nc', 'frame01.nc', 'frame02.nc']
    for name in os.listdir(single):
        a, b = _nc4.open_file(os.path.join(single, name)), _nc4.open_file(os.path.join(multi, name))
        assert list(a.vars) == list(b.vars)
        for k, v in a.vars.items():
            assert np.array_equal(v.data, b.vars[k].data, equal_nan=True), (name, k)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode != 0 and 'already exists' in res.stderr
