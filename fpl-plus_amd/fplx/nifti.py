"""SimpleITK-free NIfTI-1 (.nii / .nii.gz) reader and writer behind PyMIC's image IO functions.

Mirrors PyMIC/pymic/io/image_read_write.py:9-37 (`load_nifty_volume_as_4d_array`), :71-95 (`load_image_as_nd_array`)
and :97-113 (`save_array_as_nifty_volume`): same names, same dictionary keys, arrays as [C, D, H, W] (NIfTI stores x
fastest, so the file's (nx, ny, nz) block IS a C-ordered [D, H, W] array), spacing returned as (z, y, x), origin and
direction in ITK's LPS frame (NIfTI is RAS: the x and y rows change sign).  Host-side file parsing: no GPU work here.

Parity note: SimpleITK is not in this image, so the reference's reader could not be run; the format follows the published
NIfTI-1 header layout (348-byte header, `n+1` magic, data at vox_offset) and the tests pin it on a data file the reference
itself ships (written by SimpleITK): its voxels, spacing, origin and direction read back as expected, and the writer
re-creates that file's header fields from the values the reader returned.
"""
import gzip
import math
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _open(name, mode):
    return gzip.open(name, mode) if name.endswith(".gz") else open(name, mode)


def _quatern_to_mat(b, c, d, qfac, pix):
    a = math.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    r = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]], np.float64)
    return r * np.array([pix[0], pix[1], pix[2] * qfac], np.float64)[None, :]


def _mat_to_quatern(r):
    """rotation matrix (columns orthonormal, det may be -1) -> (b, c, d, qfac), the NIfTI-1 standard's recipe"""
    r = np.array(r, np.float64)
    qfac = 1.0
    if np.linalg.det(r) < 0:
        r[:, 2] = -r[:, 2]
        qfac = -1.0
    a = r[0, 0] + r[1, 1] + r[2, 2] + 1.0
    if a > 0.5:
        a = 0.5 * math.sqrt(a)
        b, c, d = 0.25 * (r[2, 1] - r[1, 2]) / a, 0.25 * (r[0, 2] - r[2, 0]) / a, 0.25 * (r[1, 0] - r[0, 1]) / a
    else:
        xd, yd, zd = 1.0 + r[0, 0] - (r[1, 1] + r[2, 2]), 1.0 + r[1, 1] - (r[0, 0] + r[2, 2]), 1.0 + r[2, 2] - (r[0, 0] + r[1, 1])
        if xd > 1.0:
            b = 0.5 * math.sqrt(xd)
            c, d, a = 0.25 * (r[0, 1] + r[1, 0]) / b, 0.25 * (r[0, 2] + r[2, 0]) / b, 0.25 * (r[2, 1] - r[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * math.sqrt(yd)
            b, d, a = 0.25 * (r[0, 1] + r[1, 0]) / c, 0.25 * (r[1, 2] + r[2, 1]) / c, 0.25 * (r[0, 2] - r[2, 0]) / c
        else:
            d = 0.5 * math.sqrt(zd)
            b, c, a = 0.25 * (r[0, 2] + r[2, 0]) / d, 0.25 * (r[1, 2] + r[2, 1]) / d, 0.25 * (r[1, 0] - r[0, 1]) / d
        if a < 0:
            b, c, d = -b, -c, -d
    return b, c, d, qfac


def read_nifti(filename):
    """-> (array [nz, ny, nx] (or [nt, nz, ny, nx]), spacing_xyz, origin_lps, direction_lps 3x3 row-major tuple)"""
    with _open(filename, "rb") as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError("{0:}: not a NIfTI-1 file (shorter than the header)".format(filename))
    en = "<"
    if struct.unpack("<i", raw[:4])[0] != 348:
        en = ">"
        if struct.unpack(">i", raw[:4])[0] != 348:
            raise ValueError("{0:}: not a NIfTI-1 file (sizeof_hdr != 348)".format(filename))
    if raw[344:347] != b"n+1":
        raise ValueError("{0:}: only single-file NIfTI-1 (magic 'n+1') is supported".format(filename))
    dim = struct.unpack(en + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(en + "2h", raw[70:74])
    pixdim = struct.unpack(en + "8f", raw[76:108])
    vox_offset = int(struct.unpack(en + "f", raw[108:112])[0])
    slope, inter = struct.unpack(en + "2f", raw[112:120])
    qform_code, sform_code = struct.unpack(en + "2h", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(en + "6f", raw[256:280])
    srow = np.array(struct.unpack(en + "12f", raw[280:328]), np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError("{0:}: unsupported NIfTI datatype {1:}".format(filename, datatype))
    nd = dim[0]
    if nd < 3 or nd > 4:
        raise ValueError("unsupported image dim: {0:}".format(nd))
    shape = tuple(int(v) for v in dim[1:1 + nd])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(en)
    count = int(np.prod(shape))
    if len(raw) < vox_offset + count * dt.itemsize:
        raise ValueError("{0:}: truncated voxel data".format(filename))
    data = np.frombuffer(raw, dt, count, vox_offset).reshape(shape[::-1]).astype(dt.newbyteorder("="))
    if slope != 0.0 and not (slope == 1.0 and inter == 0.0):
        data = data.astype(np.float64) * slope + inter
    if sform_code > 0:
        affine, offs = srow[:, :3], srow[:, 3]
    elif qform_code > 0:
        affine = _quatern_to_mat(qb, qc, qd, -1.0 if pixdim[0] < 0 else 1.0, pixdim[1:4])
        offs = np.array([qx, qy, qz], np.float64)
    else:
        affine, offs = np.diag(np.array(pixdim[1:4], np.float64)), np.zeros(3)
    spacing = np.sqrt((affine ** 2).sum(0))
    spacing = np.where(spacing > 0, spacing, 1.0)
    lps = np.array([-1.0, -1.0, 1.0])[:, None]
    direction = (affine / spacing[None, :]) * lps + 0.0      # + 0.0: no negative zeros
    origin = offs * lps[:, 0] + 0.0
    return data, tuple(float(v) for v in spacing), tuple(float(v) for v in origin), tuple(float(v) for v in direction.ravel())


def write_nifti(filename, data, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
                direction=(1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0)):
    """data [D, H, W] -> single-file NIfTI-1 with the ITK (LPS) geometry given; q-form and s-form both written"""
    data = np.ascontiguousarray(data)
    if data.ndim != 3:
        raise ValueError("write_nifti: data must be [D, H, W]")
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype.name not in _CODES:
        raise ValueError("write_nifti: unsupported dtype {0:}".format(data.dtype))
    lps = np.array([-1.0, -1.0, 1.0])
    rot = np.array(direction, np.float64).reshape(3, 3) * lps[:, None]
    sp = np.array(spacing, np.float64)
    off = np.array(origin, np.float64) * lps
    b, c, d, qfac = _mat_to_quatern(rot)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    hdr[38] = ord("r")                                    # `regular`, as every NIfTI writer sets it
    struct.pack_into("<8h", hdr, 40, 3, data.shape[2], data.shape[1], data.shape[0], 1, 1, 1, 1)
    struct.pack_into("<2h", hdr, 70, _CODES[data.dtype.name], data.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, qfac, sp[0], sp[1], sp[2], 0, 0, 0, 0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    hdr[123] = 2                                          # xyzt_units: millimetres
    struct.pack_into("<2h", hdr, 252, 1, 1)               # qform_code, sform_code = scanner anatomical
    struct.pack_into("<6f", hdr, 256, b + 0.0, c + 0.0, d + 0.0, off[0] + 0.0, off[1] + 0.0, off[2] + 0.0)
    srow = np.concatenate([rot * sp[None, :], off[:, None]], 1) + 0.0      # no negative zeros
    struct.pack_into("<12f", hdr, 280, *srow.ravel())
    hdr[344:348] = b"n+1\0"
    with _open(filename, "wb") as f:
        f.write(bytes(hdr))
        f.write(data.astype(data.dtype.newbyteorder("<"), copy=False).tobytes())


def load_nifty_volume_as_4d_array(filename):
    """image_read_write.py:9-37: {'data_array' [C,D,H,W], 'origin', 'spacing' (z,y,x), 'direction'}"""
    data, spacing, origin, direction = read_nifti(filename)
    if data.ndim == 4:
        assert data.shape[0] == 1                       # a 4-D file must carry a single frame
    elif data.ndim == 3:
        data = np.expand_dims(data, axis=0)
    else:
        raise ValueError("unsupported image dim: {0:}".format(data.ndim))
    return {'data_array': data, 'origin': origin, 'spacing': (spacing[2], spacing[1], spacing[0]),
            'direction': direction}


def load_image_as_nd_array(image_name):
    """image_read_write.py:71-95 for the formats FPL+ uses (.nii.gz / .nii volumes, .npy dictionaries)"""
    if image_name.endswith(".nii.gz") or image_name.endswith(".nii"):
        return load_nifty_volume_as_4d_array(image_name)
    if image_name.endswith(".npy"):
        return np.load(image_name, allow_pickle=True)
    raise ValueError("unsupported image format")


def save_array_as_nifty_volume(data, image_name, reference_name=None):
    """image_read_write.py:97-113: [D,H,W] array, geometry copied from reference_name when given"""
    if hasattr(data, "detach"):                          # device tensor -> host
        data = data.detach().cpu().numpy()
    if reference_name is not None:
        _, spacing, origin, direction = read_nifti(reference_name)
        write_nifti(image_name, data, spacing, origin, direction)
    else:
        write_nifti(image_name, data)
