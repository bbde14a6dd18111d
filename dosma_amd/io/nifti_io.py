"""``NiftiReader`` / ``NiftiWriter`` -- mirror of the reference's ``dosma/core/io/nifti_io.py:21-101`` on the
built-in NIfTI-1 codec (``_nifti1.py``; the reference goes through nibabel, which this image does not have)."""
import os

import numpy as np

from dosma_amd.defaults import AFFINE_DECIMAL_PRECISION, SCANNER_ORIGIN_DECIMAL_PRECISION
from dosma_amd.io import _nifti1
from dosma_amd.io.format_io import ImageDataFormat
from dosma_amd.med_volume import MedicalVolume

__all__ = ["NiftiReader", "NiftiWriter"]


class NiftiReader:
    data_format_code = ImageDataFormat.nifti

    def load(self, file_path, mmap: bool = False) -> MedicalVolume:
        """One NIfTI file -> one MedicalVolume: float64 data (nibabel ``get_fdata``), the affine's direction
        vectors and origin rounded to 4 decimals (reference :46-61, ``MedicalVolume.from_nib`` med_volume.py:902-943)."""
        file_path = str(file_path)
        if not os.path.isfile(file_path):
            raise FileNotFoundError("{} not found".format(file_path))
        if not self.data_format_code.is_filetype(file_path):
            raise ValueError("{} must be a file with extension '.nii' or '.nii.gz'".format(file_path))
        data, affine = _nifti1.read(file_path, mmap=mmap)
        affine = np.array(affine)
        affine[:3, :3] = np.round(affine[:3, :3], AFFINE_DECIMAL_PRECISION)
        affine[:3, 3] = np.round(affine[:3, 3], SCANNER_ORIGIN_DECIMAL_PRECISION)
        return MedicalVolume(data, affine)

    read = load
    __call__ = load

    def state_dict(self):
        return dict(self.__dict__)

    def load_state_dict(self, state_dict):
        for k, v in state_dict.items():
            setattr(self, k, v)
        return self


class NiftiWriter:
    data_format_code = ImageDataFormat.nifti

    def save(self, volume: MedicalVolume, file_path: str):
        """``nib.save(volume.to_nib(), file_path)`` of the reference (:79-96): the array in its own dtype,
        the affine in the sform."""
        file_path = str(file_path)
        if not self.data_format_code.is_filetype(file_path):
            raise ValueError("{} must be a file with extension '.nii' or '.nii.gz'".format(file_path))
        dirname = os.path.dirname(file_path)
        if dirname:
            os.makedirs(dirname, exist_ok=True)
        _nifti1.write(file_path, volume.volume, volume.affine)

    write = save
    __call__ = save

    def state_dict(self):
        return dict(self.__dict__)

    def load_state_dict(self, state_dict):
        for k, v in state_dict.items():
            setattr(self, k, v)
        return self
