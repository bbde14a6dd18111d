"""Minimal NIfTI-1 single-file (.nii / .nii.gz) codec -- SURVEY.md 8(f) row N3.

The reference reads and writes quantitative maps through nibabel (``dosma/core/io/nifti_io.py:21-101``:
``nib.load`` -> ``MedicalVolume.from_nib``; ``MedicalVolume.to_nib`` -> ``nib.save``).  nibabel is a third-party
dependency that is not part of this image, so the on-disk format is implemented here from the published
NIfTI-1 specification (nifti1.h, NIH/NIfTI DFWG 2005): the 348-byte header, 4 extension bytes, raw voxels in
Fortran order at ``vox_offset``; transform precedence sform > qform > pixdim (nibabel ``get_best_affine``).
PARITY UNPINNED against nibabel itself (absent here and on the GPU box); the tests check the writer and the
reader against the specification's byte offsets independently of each other.

What ``nib.Nifti1Image(data, affine)`` + ``nib.save`` produce, and this writer reproduces: sform_code = 2
("aligned") holding the affine, qform_code = 0 with the quaternion / qoffset / pixdim fields still derived from
the affine, xyzt_units = 0, scl_slope = scl_inter = NaN (no scaling), magic "n+1", vox_offset = 352.
"""
import gzip
import struct

import numpy as np

_FMT = (
    "i10s18sihcB"   # sizeof_hdr data_type db_name extents session_error regular dim_info
    "8h"            # dim
    "3f"            # intent_p1..3
    "4h"            # intent_code datatype bitpix slice_start
    "8f"            # pixdim
    "3f"            # vox_offset scl_slope scl_inter
    "hBB"           # slice_end slice_code xyzt_units
    "4f"            # cal_max cal_min slice_duration toffset
    "2i"            # glmax glmin
    "80s24s"        # descrip aux_file
    "2h"            # qform_code sform_code
    "6f"            # quatern_b c d qoffset_x y z
    "12f"           # srow_x srow_y srow_z
    "16s4s"         # intent_name magic
)
_HDR = struct.Struct("<" + _FMT)  # written little-endian; read in either byte order
assert _HDR.size == 348

# datatype code -> numpy dtype (nifti1.h DT_* / NIFTI_TYPE_*)
_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8",
           1280: "u8"}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _quatern_to_rotation(b, c, d):
    """nifti1.h: rotation matrix of the unit quaternion (a, b, c, d), a = sqrt(1 - b^2 - c^2 - d^2)."""
    a = 1.0 - (b * b + c * c + d * d)
    if a < 1e-7:  # special case: 180 degree rotation, renormalise
        a = 1.0 / np.sqrt(b * b + c * c + d * d)
        b, c, d = b * a, c * a, d * a
        a = 0.0
    else:
        a = np.sqrt(a)
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
        [2 * b * c + 2 * a * d, a * a + c * c - b * b - d * d, 2 * c * d - 2 * a * b],
        [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a + d * d - c * c - b * b]])


def _rotation_to_quatern(affine):
    """(b, c, d, qfac, spacing) of a 4x4 affine -- the nifti_mat44_to_quatern procedure of the specification:
    normalise the columns, make the matrix proper (qfac = -1 flips the third column), orthogonalise by polar
    decomposition, then the numerically stable branch of the rotation -> quaternion formulas."""
    R = np.array(affine[:3, :3], dtype=np.float64)
    spacing = np.sqrt((R * R).sum(axis=0))
    spacing[spacing == 0] = 1.0
    R = R / spacing
    u, _, vt = np.linalg.svd(R)
    R = u @ vt  # closest orthogonal matrix
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R[:, 2] = -R[:, 2]
        qfac = -1.0
    r11, r12, r13 = R[0]
    r21, r22, r23 = R[1]
    r31, r32, r33 = R[2]
    a = r11 + r22 + r33 + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b = 0.25 * (r32 - r23) / a
        c = 0.25 * (r13 - r31) / a
        d = 0.25 * (r21 - r12) / a
    else:
        xd, yd, zd = 1.0 + r11 - (r22 + r33), 1.0 + r22 - (r11 + r33), 1.0 + r33 - (r11 + r22)
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c = 0.25 * (r12 + r21) / b
            d = 0.25 * (r13 + r31) / b
            a = 0.25 * (r32 - r23) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b = 0.25 * (r12 + r21) / c
            d = 0.25 * (r23 + r32) / c
            a = 0.25 * (r13 - r31) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b = 0.25 * (r13 + r31) / d
            c = 0.25 * (r23 + r32) / d
            a = 0.25 * (r21 - r12) / d
        if a < 0.0:
            b, c, d = -b, -c, -d
    return float(b), float(c), float(d), qfac, spacing


def write(path, data, affine):
    """Write ``data`` (any of the NIfTI dtypes; bool is stored as uint8) with a 4x4 RAS+ ``affine``."""
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype == np.float16:
        data = data.astype(np.float32)
    dt = data.dtype.newbyteorder("<") if data.dtype.itemsize > 1 else data.dtype
    if np.dtype(dt.str.lstrip("<>=|")) not in _CODES:
        raise ValueError(f"data type {data.dtype} cannot be stored in a NIfTI-1 file")
    if not 1 <= data.ndim <= 7:
        raise ValueError("NIfTI-1 stores 1 to 7 dimensions")
    affine = np.asarray(affine, dtype=np.float64)
    if affine.shape != (4, 4):
        raise ValueError("`affine` must be a 4x4 matrix")
    b, c, d, qfac, spacing = _rotation_to_quatern(affine)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    pixdim = [qfac] + [float(s) for s in spacing] + [1.0] * 4
    if data.ndim < 3:
        pixdim[1 + data.ndim:4] = [1.0] * (3 - data.ndim)
    nan = float("nan")
    hdr = _HDR.pack(
        348, b"", b"", 0, 0, b"r", 0,
        *dim,
        0.0, 0.0, 0.0,
        0, _CODES[np.dtype(dt.str.lstrip("<>=|"))], dt.itemsize * 8, 0,
        *pixdim,
        352.0, nan, nan,
        0, 0, 0,
        0.0, 0.0, 0.0, 0.0,
        0, 0,
        b"", b"",
        0, 2,
        b, c, d, float(affine[0, 3]), float(affine[1, 3]), float(affine[2, 3]),
        *[float(v) for v in affine[:3].reshape(-1)],
        b"", b"n+1\x00")
    payload = np.asarray(data, dtype=dt).tobytes(order="F")
    opener = gzip.open if str(path).lower().endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(hdr)
        f.write(b"\x00\x00\x00\x00")
        f.write(payload)


def read_header(raw):
    """Parse the first 348 bytes -> (dict of fields, byte order character)."""
    if len(raw) < 348:
        raise ValueError("file is too short to be NIfTI-1")
    for bo in ("<", ">"):
        if struct.unpack(bo + "i", raw[:4])[0] == 348:
            break
    else:
        raise ValueError("not a NIfTI-1 file (sizeof_hdr != 348)")
    s = struct.Struct(bo + _FMT)
    v = s.unpack(raw[:348])
    h = dict(dim=v[7:15], intent_code=v[18], datatype=v[19], bitpix=v[20], pixdim=v[22:30], vox_offset=v[30],
             scl_slope=v[31], scl_inter=v[32], xyzt_units=v[35], descrip=v[42], qform_code=v[44],
             sform_code=v[45], quatern=v[46:49], qoffset=v[49:52], srow=np.array(v[52:64], dtype=np.float64).reshape(3, 4),
             magic=v[65])
    if h["magic"][:3] not in (b"n+1", b"ni1"):
        raise ValueError("not a NIfTI-1 file (bad magic)")
    if h["magic"][:3] == b"ni1":
        raise ValueError("two-file NIfTI (.hdr/.img) is not supported")
    return h, bo


def best_affine(h):
    """sform if sform_code > 0, else qform if qform_code > 0, else the pixdim scaling (nibabel get_best_affine)."""
    aff = np.eye(4)
    if h["sform_code"] > 0:
        aff[:3] = h["srow"]
        return aff
    pixdim = h["pixdim"]
    if h["qform_code"] > 0:
        R = _quatern_to_rotation(*[float(q) for q in h["quatern"]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        S = np.diag([pixdim[1], pixdim[2], pixdim[3] * qfac]).astype(np.float64)
        aff[:3, :3] = R @ S
        aff[:3, 3] = h["qoffset"]
        return aff
    ndim = h["dim"][0]
    shape = np.array(h["dim"][1:4], dtype=np.float64)
    zooms = np.array([abs(p) if (i < ndim and p != 0) else 1.0 for i, p in enumerate(pixdim[1:4])])
    aff[:3, :3] = np.diag(zooms)
    aff[0, 0] = -zooms[0]  # nibabel's default: radiological, origin at the array centre
    aff[:3, 3] = -aff[:3, :3] @ ((shape - 1) / 2.0)
    return aff


def read(path, mmap=False):
    """-> (float64 array in the file's shape with scl_slope/scl_inter applied [``get_fdata``], 4x4 affine).
    ``mmap=True`` returns the stored dtype as a read-only ``np.memmap`` (uncompressed files without scaling only)."""
    gz = str(path).lower().endswith(".gz")
    opener = gzip.open if gz else open
    with opener(path, "rb") as f:
        raw = f.read(352)
        h, bo = read_header(raw)
        ndim = h["dim"][0]
        if not 1 <= ndim <= 7:
            raise ValueError(f"bad dim[0]={ndim}")
        shape = tuple(int(s) for s in h["dim"][1:1 + ndim])
        if h["datatype"] not in _DTYPES:
            raise ValueError(f"NIfTI datatype {h['datatype']} is not supported")
        dt = np.dtype(_DTYPES[h["datatype"]]).newbyteorder(bo)
        off = int(h["vox_offset"]) or 352
        slope, inter = float(h["scl_slope"]), float(h["scl_inter"])
        scaled = np.isfinite(slope) and slope != 0.0 and not (slope == 1.0 and (inter == 0.0 or not np.isfinite(inter)))
        if mmap:
            if gz or scaled:
                raise ValueError("Underlying array in the NIfTI file cannot be mem-mapped. Please set mmap=False.")
            data = np.memmap(path, dtype=dt, mode="r", offset=off, shape=shape, order="F")
            return data, best_affine(h)
        f.seek(off)
        count = int(np.prod(shape))
        buf = f.read(count * dt.itemsize)
    if len(buf) != count * dt.itemsize:
        raise ValueError("NIfTI file is truncated")
    data = np.frombuffer(buf, dtype=dt).reshape(shape, order="F").astype(np.float64)
    if scaled:
        data = data * slope + (inter if np.isfinite(inter) else 0.0)
    return data, best_affine(h)
