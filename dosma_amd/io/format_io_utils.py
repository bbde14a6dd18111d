"""Format dispatch helpers (reference ``dosma/core/io/format_io_utils.py``: ``get_reader`` / ``get_writer``
:24-58, ``convert_image_data_format`` :61-85, ``generic_load`` :103-155).  DICOM is out of scope."""
import os

from dosma_amd.io.format_io import ImageDataFormat
from dosma_amd.io.nifti_io import NiftiReader, NiftiWriter

__all__ = ["get_reader", "get_writer", "convert_image_data_format", "generic_load", "read", "write"]


def get_reader(data_format: ImageDataFormat):
    if data_format == ImageDataFormat.nifti:
        return NiftiReader()
    raise NotImplementedError(f"{data_format}: only NIfTI I/O is built (DICOM is out of scope, SURVEY.md 2)")


def get_writer(data_format: ImageDataFormat):
    if data_format == ImageDataFormat.nifti:
        return NiftiWriter()
    raise NotImplementedError(f"{data_format}: only NIfTI I/O is built (DICOM is out of scope, SURVEY.md 2)")


def convert_image_data_format(file_or_dir_path, new_data_format: ImageDataFormat):
    """Path of the same volume under another format's naming convention: ``x.nii.gz`` <-> directory ``x``."""
    file_or_dir_path = str(file_or_dir_path)
    current = ImageDataFormat.get_image_data_format(file_or_dir_path)
    if current == new_data_format:
        return file_or_dir_path
    if current == ImageDataFormat.dicom and new_data_format == ImageDataFormat.nifti:
        return file_or_dir_path.rstrip(os.sep) + ".nii.gz"
    if current == ImageDataFormat.nifti and new_data_format == ImageDataFormat.dicom:
        base = file_or_dir_path
        for ext in (".nii.gz", ".nii"):
            if base.lower().endswith(ext):
                return base[: -len(ext)]
    raise NotImplementedError(f"{current} -> {new_data_format}")


def generic_load(file_or_dir_path, expected_num_volumes: int = None):
    variations = [convert_image_data_format(file_or_dir_path, fmt) for fmt in ImageDataFormat]
    exist_path = None
    for fp in variations:
        if os.path.exists(fp):
            if exist_path is not None and fp != exist_path:
                raise ValueError("Ambiguous loading state - multiple possible files to load from %s" % str(variations))
            exist_path = fp
    if exist_path is None:
        raise FileNotFoundError("No file associated with basename %s found" % os.path.basename(str(file_or_dir_path)))
    vols = get_reader(ImageDataFormat.get_image_data_format(exist_path)).load(exist_path)
    if expected_num_volumes is None:
        return vols
    if type(vols) is not list:
        vols = [vols]
    assert len(vols) == expected_num_volumes, "Expected %d volumes, got %d" % (expected_num_volumes, len(vols))
    return vols[0] if len(vols) == 1 else vols


def read(path, data_format: ImageDataFormat = None, unpack: bool = False, **kwargs):
    """``dm.read`` (reference :158-193): the format comes from the extension unless given (enum or its name)."""
    if data_format is None:
        data_format = ImageDataFormat.get_image_data_format(path)
    elif isinstance(data_format, str):
        data_format = ImageDataFormat[data_format]
    out = get_reader(data_format).load(path, **kwargs)
    if unpack and isinstance(out, (tuple, list)) and len(out) == 1:
        out = out[0]
    return out


def write(vol, path, data_format: ImageDataFormat = None, **kwargs) -> None:
    """``dm.write`` (reference :196-222)."""
    if data_format is None:
        data_format = ImageDataFormat.get_image_data_format(path)
    elif isinstance(data_format, str):
        data_format = ImageDataFormat[data_format]
    get_writer(data_format).save(vol, path, **kwargs)


load = read
save = write
