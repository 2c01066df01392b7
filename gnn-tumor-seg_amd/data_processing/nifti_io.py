"""Minimal NIfTI-1 (.nii / .nii.gz) reader and writer.

The reference goes through nibabel (data_processing/nifti_io.py:42-57), which is not
available here; this module writes/reads single-file NIfTI-1 volumes with the fixed BraTS
affine the reference uses and returns arrays the same way (`np.array(dataobj, dtype)`).
Header layout: the NIfTI-1 specification (nifti1.h), field by field; pinned byte for byte by
tests/test_nifti_bytes.py against a header assembled by hand from that specification.
PARITY UNPINNED against nibabel itself (absent; the reference holds no .nii fixture): fields
the specification leaves to the writer (scl_slope 1.0 here, NaN in nibabel — both mean "no
scaling"; descrip) may differ; every reader-relevant field is the spec's.
"""
import gzip
import struct

import numpy as np

BRATS_AFFINE = np.array([
    [-1.0, -0.0, -0.0, -0.0],
    [-0.0, -1.0, -0.0, 239.0],
    [0.0, 0.0, 1.0, 0.0],
    [0.0, 0.0, 0.0, 1.0],
])

# NIfTI datatype code <-> numpy dtype, bits per voxel
_CODES = {
    2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64,
    256: np.int8, 512: np.uint16, 768: np.uint32, 1024: np.int64, 1280: np.uint64,
}
_BY_DTYPE = {np.dtype(v): k for k, v in _CODES.items()}


def _quaternion_bcd(affine):
    """(b, c, d) of the unit quaternion of the affine's rotation part (NIfTI-1 spec, nifti1.h
    "METHOD 2"; a = sqrt(1 - b^2 - c^2 - d^2) >= 0).  Columns are normalised; an improper
    rotation has its third column flipped (that sign is qfac = pixdim[0])."""
    r = np.array(affine, dtype=np.float64)[:3, :3]
    r = r / np.sqrt((r * r).sum(axis=0))
    if np.linalg.det(r) < 0:
        r[:, 2] = -r[:, 2]
    a = r[0, 0] + r[1, 1] + r[2, 2] + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b, c, d = 0.25 * (r[2, 1] - r[1, 2]) / a, 0.25 * (r[0, 2] - r[2, 0]) / a, 0.25 * (r[1, 0] - r[0, 1]) / a
    else:
        xd, yd, zd = 1.0 + r[0, 0] - (r[1, 1] + r[2, 2]), 1.0 + r[1, 1] - (r[0, 0] + r[2, 2]), \
            1.0 + r[2, 2] - (r[0, 0] + r[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c, d, a = 0.25 * (r[0, 1] + r[1, 0]) / b, 0.25 * (r[0, 2] + r[2, 0]) / b, 0.25 * (r[2, 1] - r[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b, d, a = 0.25 * (r[0, 1] + r[1, 0]) / c, 0.25 * (r[1, 2] + r[2, 1]) / c, 0.25 * (r[0, 2] - r[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b, c, a = 0.25 * (r[0, 2] + r[2, 0]) / d, 0.25 * (r[1, 2] + r[2, 1]) / d, 0.25 * (r[1, 0] - r[0, 1]) / d
        if a < 0.0:
            b, c, d = -b, -c, -d
    return float(b), float(c), float(d)


def _open(fp, mode):
    return gzip.open(fp, mode) if str(fp).endswith(".gz") else open(fp, mode)


def save_as_nifti(img, fp, affine=BRATS_AFFINE):
    img = np.asarray(img)
    if img.dtype == np.bool_:
        img = img.astype(np.uint8)
    if img.dtype not in _BY_DTYPE:
        raise ValueError(f"unsupported dtype for NIfTI: {img.dtype}")
    if not 1 <= img.ndim <= 7:
        raise ValueError("NIfTI supports 1 to 7 dimensions")
    dim = [img.ndim] + list(img.shape) + [1] * (7 - img.ndim)
    pixdim = [1.0] * 8
    pixdim[0] = -1.0 if np.linalg.det(affine[:3, :3]) < 0 else 1.0
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    hdr[38:39] = b"r"                            # `regular`: unused by NIfTI-1, set by Analyze-lineage writers
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<hh", hdr, 70, _BY_DTYPE[img.dtype], img.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<f", hdr, 108, 352.0)      # vox_offset
    struct.pack_into("<ff", hdr, 112, 1.0, 0.0)  # scl_slope, scl_inter
    struct.pack_into("<hh", hdr, 252, 0, 2)      # qform_code 0 (unknown), sform_code 2 (aligned)
    # the quaternion / offset fields carry the same transform (ignored while qform_code is 0)
    struct.pack_into("<3f", hdr, 256, *_quaternion_bcd(affine))
    struct.pack_into("<3f", hdr, 268, *[float(affine[r][3]) for r in range(3)])
    for r in range(3):
        struct.pack_into("<4f", hdr, 280 + 16 * r, *[float(x) for x in affine[r]])
    hdr[344:348] = b"n+1\0"
    with _open(fp, "wb") as f:
        f.write(bytes(hdr) + b"\0\0\0\0")
        f.write(np.asfortranarray(img).tobytes(order="F"))


def read_nifti(fp, data_type):
    with _open(fp, "rb") as f:
        raw = f.read()
    endian = "<" if struct.unpack_from("<i", raw, 0)[0] == 348 else ">"
    dim = struct.unpack_from(endian + "8h", raw, 40)
    code, _bits = struct.unpack_from(endian + "hh", raw, 70)
    if code not in _CODES:
        raise ValueError(f"unsupported NIfTI datatype code {code}")
    vox_offset = int(struct.unpack_from(endian + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(endian + "ff", raw, 112)
    shape = tuple(dim[1:1 + dim[0]])
    dt = np.dtype(_CODES[code]).newbyteorder(endian)
    data = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=max(vox_offset, 352))
    data = data.reshape(shape, order="F")
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0 and not np.isnan(slope):
            data = data * slope + inter
    return np.array(data, dtype=data_type)
