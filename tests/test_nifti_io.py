"""NIfTI-1 map I/O: the writer and the reader are each checked against the byte layout of
the published specification (nifti1.h) *independently* -- a hand-packed file for the reader, raw struct offsets
for the writer -- then together (round trips), then through QuantitativeValue.save_data / load_data like the
reference's tests (tests/core/io/test_nifti_io.py:52-70, tests/core/test_quant_vals.py:33-50).
nibabel itself is absent: parity with it is unpinned (see dosma_amd/io/_nifti1.py)."""
import gzip
import os
import struct

import numpy as np
import pytest

from dosma_amd import ImageDataFormat, MedicalVolume, NiftiReader, NiftiWriter
from dosma_amd.io import _nifti1, generic_load
from dosma_amd.quant_vals import T2

AFF = np.array([[0.0, 0.0, 1.5, -40.25], [-0.3125, 0.0, 0.0, 60.5], [0.0, -0.3125, 0.0, 70.0], [0, 0, 0, 1.0]])


def hand_packed_nifti(data, srow=None, quatern=None, qoffset=(0, 0, 0), pixdim=(1, 1, 1, 1), scl=(0.0, 0.0),
                      code=16, bo="<"):
    """A NIfTI-1 file built field by field at the specification's byte offsets (no use of the codec)."""
    h = bytearray(352)
    struct.pack_into(bo + "i", h, 0, 348)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into(bo + "8h", h, 40, *dim)
    struct.pack_into(bo + "hh", h, 70, code, data.dtype.itemsize * 8)
    struct.pack_into(bo + "8f", h, 76, *(list(pixdim) + [1.0] * (8 - len(pixdim))))
    struct.pack_into(bo + "f", h, 108, 352.0)
    struct.pack_into(bo + "ff", h, 112, *scl)
    struct.pack_into(bo + "hh", h, 252, 1 if quatern is not None else 0, 2 if srow is not None else 0)
    if quatern is not None:
        struct.pack_into(bo + "6f", h, 256, *quatern, *qoffset)
    if srow is not None:
        struct.pack_into(bo + "12f", h, 280, *np.asarray(srow, dtype=np.float64).reshape(-1))
    h[344:348] = b"n+1\x00"
    return bytes(h) + data.astype(data.dtype.newbyteorder(bo)).tobytes(order="F")


def test_reader_against_the_specification(tmp_path):
    rng = np.random.default_rng(0)
    data = rng.normal(size=(5, 4, 3)).astype(np.float32)
    # sform present: used as is
    p = tmp_path / "a.nii"
    p.write_bytes(hand_packed_nifti(data, srow=AFF[:3]))
    mv = NiftiReader().load(str(p))
    assert isinstance(mv, MedicalVolume) and mv.dtype == np.float64 and mv.shape == data.shape
    assert np.array_equal(mv.volume, data.astype(np.float64)) and np.allclose(mv.affine, AFF)
    assert mv.orientation == MedicalVolume(data, AFF).orientation
    # gzip + big-endian + int16 with scl_slope / scl_inter
    raw = rng.integers(-300, 300, size=(4, 3, 2)).astype(np.int16)
    p = tmp_path / "b.nii.gz"
    with gzip.open(p, "wb") as f:
        f.write(hand_packed_nifti(raw, srow=np.eye(4)[:3], scl=(0.5, 10.0), code=4, bo=">"))
    mv = NiftiReader()(str(p))
    assert np.array_equal(mv.volume, raw * 0.5 + 10.0)
    # qform only: 90-degree rotation about z (quaternion b=c=0, d=sin 45), spacing from pixdim, qfac=-1 flips z
    d = np.sqrt(0.5)
    p = tmp_path / "c.nii"
    p.write_bytes(hand_packed_nifti(data, quatern=(0.0, 0.0, d), qoffset=(1.0, 2.0, 3.0), pixdim=(-1.0, 2.0, 3.0, 4.0)))
    mv = NiftiReader().load(str(p))
    want = np.array([[0, -3.0, 0, 1.0], [2.0, 0, 0, 2.0], [0, 0, -4.0, 3.0], [0, 0, 0, 1.0]])
    assert np.allclose(mv.affine, want, atol=1e-4)
    # errors of the reference reader (nifti_io.py:46-53)
    with pytest.raises(FileNotFoundError):
        NiftiReader().load(str(tmp_path / "bleh.nii"))
    with pytest.raises(FileNotFoundError):
        NiftiReader().load(str(tmp_path))
    other = tmp_path / "I0002.dcm"
    other.write_bytes(b"x")
    with pytest.raises(ValueError):
        NiftiReader().load(str(other))
    bad = tmp_path / "bad.nii"
    bad.write_bytes(b"\0" * 400)
    with pytest.raises(ValueError):
        NiftiReader().load(str(bad))


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int16, np.uint16, np.uint8, np.int32])
def test_writer_against_the_specification(tmp_path, dtype):
    rng = np.random.default_rng(1)
    data = (rng.normal(size=(6, 5, 4)) * 50).astype(dtype)
    p = tmp_path / "sub" / "w.nii.gz"
    NiftiWriter().save(MedicalVolume(data, AFF), str(p))  # creates the directory (nifti_io.py:92)
    raw = gzip.open(p, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:348] == b"n+1\x00"
    assert struct.unpack_from("<8h", raw, 40) == (3, 6, 5, 4, 1, 1, 1, 1)
    code, bitpix = struct.unpack_from("<hh", raw, 70)
    assert (code, bitpix) == ({"f8": 64, "f4": 16, "i2": 4, "u2": 512, "u1": 2, "i4": 8}[np.dtype(dtype).str[1:]],
                              np.dtype(dtype).itemsize * 8)
    pixdim = struct.unpack_from("<8f", raw, 76)
    assert np.allclose(pixdim[1:4], [0.3125, 0.3125, 1.5]) and abs(pixdim[0]) == 1.0
    assert struct.unpack_from("<f", raw, 108)[0] == 352.0
    assert all(np.isnan(v) for v in struct.unpack_from("<ff", raw, 112))      # no scaling, like nibabel
    assert struct.unpack_from("<hh", raw, 252) == (0, 2)                      # qform unknown, sform aligned
    assert np.allclose(np.array(struct.unpack_from("<12f", raw, 280)).reshape(3, 4), AFF[:3])
    # the quaternion fields still describe the same rotation (as nibabel fills them)
    b, c, dq = struct.unpack_from("<3f", raw, 256)
    R = _nifti1._quatern_to_rotation(b, c, dq) @ np.diag([pixdim[1], pixdim[2], pixdim[3] * pixdim[0]])
    assert np.allclose(R, AFF[:3, :3], atol=1e-5)
    assert np.allclose(struct.unpack_from("<3f", raw, 268), AFF[:3, 3])
    vox = np.frombuffer(raw[352:], dtype=np.dtype(dtype).newbyteorder("<")).reshape(data.shape, order="F")
    assert np.array_equal(vox, data)
    with pytest.raises(ValueError):
        NiftiWriter().save(MedicalVolume(data, AFF), str(tmp_path / "eg.dcm"))


def test_round_trips_and_orientations(tmp_path):
    rng = np.random.default_rng(2)
    mv = MedicalVolume(rng.normal(size=(7, 6, 5, 2)), AFF)  # 4D (popt-like)
    for o in [("SI", "AP", "LR"), ("LR", "PA", "IS"), ("AP", "IS", "RL")]:
        v = mv.reformat(o)
        path = str(tmp_path / ("_".join(o) + ".nii.gz"))
        v.save_volume(path)
        back = NiftiReader().load(path)
        assert back.orientation == o and np.array_equal(back.volume, v.volume)
        assert np.allclose(back.affine, v.affine, atol=1e-4)
    m = MedicalVolume(rng.uniform(size=(4, 4, 4)) > 0.5, np.eye(4))  # bool mask -> uint8 on disk
    m.save_volume(str(tmp_path / "m.nii"), data_format=ImageDataFormat.nifti)
    assert np.array_equal(generic_load(str(tmp_path / "m.nii")).volume, m.volume.astype(np.float64))
    mm = _nifti1.read(str(tmp_path / "m.nii"), mmap=True)[0]
    assert isinstance(mm, np.memmap) and mm.dtype == np.uint8
    with pytest.raises(ValueError):
        _nifti1.read(str(tmp_path / "SI_AP_LR.nii.gz"), mmap=True)
    assert ImageDataFormat.get_image_data_format("a/b.nii.gz") == ImageDataFormat.nifti
    assert ImageDataFormat.get_image_data_format("a/b") == ImageDataFormat.dicom
    r = NiftiReader()
    r.load_state_dict({k: "foo" for k in r.state_dict()})
    # dm.read / dm.write (format from the extension or by name)
    import dosma_amd as dm

    dm.write(m, str(tmp_path / "w.nii.gz"))
    assert np.array_equal(dm.read(str(tmp_path / "w.nii.gz"), data_format="nifti").volume, m.volume.astype(np.float64))
    with pytest.raises(NotImplementedError):
        dm.read(str(tmp_path / "a_dicom_directory"))
    assert dm.to_affine(("SI", "AP", "LR"), (0.4, 0.4, 1.5)).shape == (4, 4)


def test_quantitative_value_save_load(tmp_path):
    """reference tests/core/test_quant_vals.py:33-50."""
    rng = np.random.default_rng(3)
    t2 = T2(MedicalVolume(np.around(rng.uniform(0, 80, (8, 8, 4)), 3), AFF))
    t2.add_additional_volume("r2", MedicalVolume(rng.uniform(size=(8, 8, 4)), AFF))
    t2.save_data(str(tmp_path))
    assert os.path.isfile(tmp_path / "t2" / "t2.nii.gz") and os.path.isfile(tmp_path / "t2" / "t2-r2.nii.gz")
    t2b = T2()
    t2b.load_data(str(tmp_path))
    assert t2b.volumetric_map.is_identical(t2.volumetric_map)
    with pytest.warns(UserWarning):
        t2.save_data(str(tmp_path / "again"), data_format=ImageDataFormat.dicom)
    with pytest.raises(FileNotFoundError):
        T2().load_data(str(tmp_path / "nowhere"))
