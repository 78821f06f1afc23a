"""fplx.nifti (host-side NIfTI-1 reader / writer behind PyMIC's image_read_write names) - CPU tests.
tests/golden/nifti/vs_gk_98_t2_lab.nii.gz is a data file the reference ships (dataset/hrT2_train/lab/), written by
SimpleITK: the reader must recover its voxels and geometry, and the writer must re-create its 352 header bytes."""
import gzip
import os
import struct

import numpy as np
import pytest

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nifti", "vs_gk_98_t2_lab.nii.gz")


def test_reads_reference_shipped_label_volume():
    from fplx import nifti
    d = nifti.load_nifty_volume_as_4d_array(FIX)
    a = d["data_array"]
    assert a.shape == (1, 30, 160, 272) and a.dtype == np.int16          # [C, D, H, W]; NIfTI dim = (272, 160, 30)
    assert set(np.unique(a)) == {0, 1} and int(a.sum()) == 8262
    idx = np.argwhere(a[0] > 0)
    assert idx.min(0).tolist() == [12, 84, 20] and idx.max(0).tolist() == [24, 120, 71]
    np.testing.assert_allclose(d["spacing"], (1.5, 0.4102, 0.4102), rtol=1e-6)        # (z, y, x)
    np.testing.assert_allclose(d["origin"], (-105.736168, -122.712135, -57.302208), rtol=1e-6)   # ITK's LPS frame
    assert d["direction"] == (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
    assert nifti.load_image_as_nd_array(FIX)["data_array"].shape == a.shape


def test_writer_recreates_the_reference_files_header(tmp_path):
    from fplx import nifti
    a = nifti.load_nifty_volume_as_4d_array(FIX)["data_array"][0]
    out = str(tmp_path / "copy.nii.gz")
    nifti.save_array_as_nifty_volume(a, out, FIX)
    want, got = gzip.open(FIX).read(), gzip.open(out).read()
    assert got[:352] == want[:352]              # every header byte SimpleITK wrote
    assert got[352:] == want[352:]
    # a uint8 mask with the geometry of the reference image: what the pseudo-label writer emits
    m = (a > 0).astype(np.uint8)
    out8 = str(tmp_path / "mask.nii.gz")
    nifti.save_array_as_nifty_volume(m, out8, FIX)
    d = nifti.load_nifty_volume_as_4d_array(out8)
    assert d["data_array"].dtype == np.uint8 and np.array_equal(d["data_array"][0], m)
    assert d["spacing"] == nifti.load_nifty_volume_as_4d_array(FIX)["spacing"]


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.int32, np.float32, np.float64])
@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_round_trip_with_oblique_geometry(tmp_path, dtype, ext):
    from fplx import nifti
    rs = np.random.RandomState(5)
    a = (rs.rand(5, 7, 9) * 100).astype(dtype)
    th = 0.3
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    rot = rot @ np.array([[1, 0, 0], [0, 0, -1.0], [0, 1, 0]])
    for direction in (rot, rot * np.array([1, 1, -1.0])[None, :]):          # proper and improper (qfac = -1)
        f = str(tmp_path / ("v" + ext))
        nifti.write_nifti(f, a, (0.5, 0.75, 2.0), (10.0, -20.0, 30.0), tuple(direction.ravel()))
        b, sp, org, dr = nifti.read_nifti(f)
        assert b.dtype == a.dtype and np.array_equal(a, b)
        np.testing.assert_allclose(sp, (0.5, 0.75, 2.0), rtol=1e-6)
        np.testing.assert_allclose(org, (10.0, -20.0, 30.0), rtol=1e-6)
        np.testing.assert_allclose(np.array(dr).reshape(3, 3), direction, atol=1e-6)
        # the q-form written next to the s-form describes the same geometry
        raw = (gzip.open(f) if f.endswith(".gz") else open(f, "rb")).read()
        qb, qc, qd = struct.unpack("<3f", raw[256:268])
        pix = struct.unpack("<4f", raw[76:92])
        q = nifti._quatern_to_mat(qb, qc, qd, pix[0], pix[1:4])
        np.testing.assert_allclose(q, np.array(struct.unpack("<12f", raw[280:328])).reshape(3, 4)[:, :3], atol=1e-5)


def test_big_endian_scaled_and_broken_files(tmp_path):
    from fplx import nifti
    a = np.arange(2 * 3 * 4, dtype=np.int16).reshape(2, 3, 4)
    f = str(tmp_path / "le.nii")
    nifti.write_nifti(f, a)
    raw = bytearray(open(f, "rb").read())
    be = bytearray(raw)

    def swap(off, fmt):
        vals = struct.unpack("<" + fmt, bytes(raw[off:off + struct.calcsize("<" + fmt)]))
        struct.pack_into(">" + fmt, be, off, *vals)
    for off, fmt in ((0, "i"), (40, "8h"), (70, "2h"), (76, "8f"), (108, "f"), (112, "2f"), (252, "2h"), (256, "6f"),
                     (280, "12f")):
        swap(off, fmt)
    be[352:] = a.astype(">i2").tobytes()
    fb = str(tmp_path / "be.nii")
    open(fb, "wb").write(bytes(be))
    b, sp, _, _ = nifti.read_nifti(fb)
    assert np.array_equal(a, b) and sp == (1.0, 1.0, 1.0)
    sc = bytearray(raw)
    struct.pack_into("<2f", sc, 112, 2.0, -3.0)                         # scl_slope / scl_inter
    fs = str(tmp_path / "sc.nii")
    open(fs, "wb").write(bytes(sc))
    assert np.array_equal(nifti.read_nifti(fs)[0], a * 2.0 - 3.0)
    bad = bytearray(raw)
    bad[344:348] = b"ni1\0"
    fbad = str(tmp_path / "bad.nii")
    open(fbad, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        nifti.read_nifti(fbad)
    open(fbad, "wb").write(bytes(raw[:380]))
    with pytest.raises(ValueError):
        nifti.read_nifti(fbad)
    with pytest.raises(ValueError):
        nifti.load_image_as_nd_array("volume.mha")
    with pytest.raises(ValueError):
        nifti.write_nifti(str(tmp_path / "x.nii"), np.zeros((2, 2)))
