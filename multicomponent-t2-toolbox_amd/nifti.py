"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer: enough for the driver's on-disk contract
(motor/motor_recon_met2_real_data.py:167-173 loads `data` and `mask`, :475-503 saves ten volumes).  The
reference uses nibabel for this; nibabel is not a dependency of this package.

Supported: little/big-endian headers, datatypes uint8/int8/int16/uint16/int32/uint32/float32/float64,
scl_slope/scl_inter scaling, up to 7 dimensions, sform/qform affine (sform preferred; a pixdim diagonal
otherwise).  Arrays are returned in the file's (x, y, z, t, ...) index order like nibabel's get_fdata()."""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


class NiftiImage:
    def __init__(self, data, affine=None, header=None):
        self._data = np.asarray(data)
        self.affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
        self.header = header or {}

    def get_fdata(self):
        return np.array(self._data, dtype=np.float64)

    @property
    def shape(self):
        return self._data.shape


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def load(path):
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError("%s: too short for a NIfTI-1 header" % path)
    end = "<"
    if struct.unpack("<i", raw[:4])[0] != 348:
        end = ">"
        if struct.unpack(">i", raw[:4])[0] != 348:
            raise ValueError("%s: not a NIfTI-1 file (sizeof_hdr != 348)" % path)
    magic = raw[344:348]
    if magic[:3] not in (b"n+1", b"ni1"):
        raise ValueError("%s: bad NIfTI magic %r" % (path, magic))
    if magic[:3] == b"ni1":
        raise ValueError("%s: two-file NIfTI (.hdr/.img) is not supported" % path)
    dim = struct.unpack(end + "8h", raw[40:56])
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError("%s: bad dim[0]=%d" % (path, ndim))
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    datatype, bitpix = struct.unpack(end + "2h", raw[70:74])
    if datatype not in _DTYPES:
        raise ValueError("%s: unsupported NIfTI datatype %d" % (path, datatype))
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F")
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0 and np.isfinite(slope):
            arr = arr.astype(np.float64) * float(slope) + float(inter)
    affine = np.eye(4)
    if sform_code > 0:
        affine[0] = struct.unpack(end + "4f", raw[280:296])
        affine[1] = struct.unpack(end + "4f", raw[296:312])
        affine[2] = struct.unpack(end + "4f", raw[312:328])
    elif qform_code > 0:
        b, c, d = struct.unpack(end + "3f", raw[256:268])
        qx, qy, qz = struct.unpack(end + "3f", raw[268:280])
        a = math_sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        affine[:3, :3] = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])
        affine[:3, 3] = (qx, qy, qz)
    else:
        affine[0, 0], affine[1, 1], affine[2, 2] = (pixdim[1] or 1.0), (pixdim[2] or 1.0), (pixdim[3] or 1.0)
    return NiftiImage(arr, affine, {"pixdim": pixdim, "datatype": datatype, "endianness": end})


def math_sqrt(x):
    return float(np.sqrt(x))


def save(img, path):
    """img: NiftiImage or (array, affine).  Written as little-endian float64/whatever dtype the array has."""
    if not isinstance(img, NiftiImage):
        img = NiftiImage(*img)
    arr = np.asarray(img._data)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype.str[1:] not in _CODES:
        arr = arr.astype(np.float64)
    arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
    if arr.ndim > 7:
        raise ValueError("NIfTI-1 stores at most 7 dimensions")
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<2h", hdr, 70, _CODES[arr.dtype.str[1:]], arr.dtype.itemsize * 8)
    aff = np.asarray(img.affine, dtype=np.float64)
    vox = [float(np.linalg.norm(aff[:3, i])) or 1.0 for i in range(3)]
    pixdim = [1.0] + vox + [1.0] * 4
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2 | (8 << 3)                                   # xyzt_units: mm, sec
    struct.pack_into("<2h", hdr, 252, 0, 2)                   # qform_code 0, sform_code 2 (aligned)
    struct.pack_into("<4f", hdr, 280, *aff[0])
    struct.pack_into("<4f", hdr, 296, *aff[1])
    struct.pack_into("<4f", hdr, 312, *aff[2])
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")
        f.write(np.asfortranarray(arr).tobytes(order="F"))
