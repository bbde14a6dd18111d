"""On-disk formats either side of the hot path: NIfTI-1 maps."""
from dosma_amd.io.format_io import ImageDataFormat  # noqa: F401
from dosma_amd.io.format_io_utils import generic_load, get_reader, get_writer, read, write  # noqa: F401
from dosma_amd.io.nifti_io import NiftiReader, NiftiWriter  # noqa: F401
