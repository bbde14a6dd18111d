"""``ImageDataFormat`` (reference ``dosma/core/io/format_io.py:32-89``): extension <-> format dispatch."""
import enum
import os

__all__ = ["ImageDataFormat"]


class ImageDataFormat(enum.Enum):
    nifti = 1, ("nii", "nii.gz")
    dicom = 2, ("dcm", "ima")

    def __new__(cls, key_code, extensions):
        obj = object.__new__(cls)
        obj._value_ = key_code
        obj.extensions = extensions
        return obj

    def is_filetype(self, file_path) -> bool:
        file_path = str(file_path)
        return any(file_path.lower().endswith(".%s" % ext.lower()) for ext in self.extensions)

    @classmethod
    def get_image_data_format(cls, file_or_dir_path):
        for fmt in cls:
            if fmt.is_filetype(file_or_dir_path):
                return fmt
        file_or_dir_path = str(file_or_dir_path)
        base, _ = os.path.splitext(file_or_dir_path)
        if base == file_or_dir_path:  # no extension: a directory -> dicom
            return ImageDataFormat.dicom
        raise ValueError(f"Unknown data format for {file_or_dir_path}")
