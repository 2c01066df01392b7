"""Byte-level pin of the NIfTI-1 writer (data_processing/nifti_io.py) without nibabel: the
348-byte header is assembled here BY HAND from the NIfTI-1 specification (nifti1.h field
offsets), for what /root/reference/data_processing/nifti_io.py:42-54 asks nibabel to write —
`Nifti1Image(img, BraTS affine)` saved as a single file — and compared byte for byte, for int16
and float64 volumes, plain and gzip-compressed.  Every byte the table does not name must be 0."""
import gzip
import struct

import numpy as np
import pytest

from data_processing import nifti_io

# (offset, struct format, field) of nifti1.h — the fields a single-file volume writer sets
SPEC = {
    "sizeof_hdr": (0, "<i"), "regular": (38, "<c"), "dim_info": (39, "<b"), "dim": (40, "<8h"),
    "intent_code": (68, "<h"), "datatype": (70, "<h"), "bitpix": (72, "<h"), "slice_start": (74, "<h"),
    "pixdim": (76, "<8f"), "vox_offset": (108, "<f"), "scl_slope": (112, "<f"), "scl_inter": (116, "<f"),
    "xyzt_units": (123, "<b"), "qform_code": (252, "<h"), "sform_code": (254, "<h"),
    "quatern_bcd": (256, "<3f"), "qoffset_xyz": (268, "<3f"),
    "srow_x": (280, "<4f"), "srow_y": (296, "<4f"), "srow_z": (312, "<4f"), "magic": (344, "<4s"),
}
DT = {np.dtype(np.int16): (4, 16), np.dtype(np.float64): (64, 64), np.dtype(np.float32): (16, 32),
      np.dtype(np.uint8): (2, 8)}


def expected_header(shape, dtype):
    code, bits = DT[np.dtype(dtype)]
    fields = {
        "sizeof_hdr": (348,), "regular": (b"r",), "dim_info": (0,),
        "dim": (len(shape), *shape, *([1] * (7 - len(shape)))),
        "intent_code": (0,), "datatype": (code,), "bitpix": (bits,), "slice_start": (0,),
        # qfac = +1: the BraTS rotation diag(-1, -1, 1) is proper; unit voxels
        "pixdim": (1.0,) * 8, "vox_offset": (352.0,), "scl_slope": (1.0,), "scl_inter": (0.0,),
        "xyzt_units": (0,), "qform_code": (0,), "sform_code": (2,),
        # 180 degrees about z: quaternion (a, b, c, d) = (0, 0, 0, 1); translation (0, 239, 0)
        "quatern_bcd": (0.0, 0.0, 1.0), "qoffset_xyz": (-0.0, 239.0, 0.0),   # the affine's last column, signs of zero kept
        "srow_x": (-1.0, -0.0, -0.0, -0.0), "srow_y": (-0.0, -1.0, -0.0, 239.0), "srow_z": (0.0, 0.0, 1.0, 0.0),
        "magic": (b"n+1\0",),
    }
    hdr = bytearray(348)
    for name, values in fields.items():
        off, fmt = SPEC[name]
        struct.pack_into(fmt, hdr, off, *values)
    return bytes(hdr)


@pytest.mark.parametrize("suffix", [".nii", ".nii.gz"])
@pytest.mark.parametrize("shape,dtype", [((5, 4, 3), np.int16), ((5, 4, 3, 4), np.float64),
                                         ((240, 240, 155), np.int16), ((7,), np.uint8)])
def test_written_file_is_the_spec_header_plus_fortran_order_voxels(tmp_path, suffix, shape, dtype):
    rng = np.random.default_rng(len(shape))
    img = (rng.standard_normal(shape) * 50).astype(dtype)
    path = str(tmp_path / ("vol" + suffix))
    nifti_io.save_as_nifti(img, path)
    raw = open(path, "rb").read()
    if suffix.endswith(".gz"):
        assert raw[:2] == b"\x1f\x8b"                       # a gzip member, as nibabel writes for .nii.gz
        raw = gzip.decompress(raw)
    want = expected_header(shape, dtype)
    got = raw[:348]
    assert got == want, [name for name, (off, fmt) in SPEC.items()
                         if got[off:off + struct.calcsize(fmt)] != want[off:off + struct.calcsize(fmt)]]
    assert raw[348:352] == b"\0\0\0\0"                      # no header extension
    assert raw[352:] == img.tobytes(order="F")              # first index fastest, little-endian
    assert len(raw) == 352 + img.size * img.itemsize
    back = nifti_io.read_nifti(path, dtype)
    assert back.dtype == dtype and np.array_equal(back, img)


def test_reader_accepts_what_other_writers_produce(tmp_path):
    """nibabel-style header (scl_slope / scl_inter = NaN, an extension block before the voxels, so
    vox_offset > 352) and a big-endian file; scaled data (slope 2, intercept 1) is applied."""
    img = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    hdr = bytearray(expected_header(img.shape, np.int16))
    struct.pack_into("<ff", hdr, 112, float("nan"), float("nan"))
    struct.pack_into("<f", hdr, 108, 368.0)
    p = tmp_path / "nan.nii"
    p.write_bytes(bytes(hdr) + b"\1\0\0\0" + b"\x10\0\0\0" + b"\4\0\0\0" + b"note\0\0\0\0" + img.tobytes(order="F"))
    assert np.array_equal(nifti_io.read_nifti(str(p), np.int16), img)
    big = bytearray(348)
    for name, (off, fmt) in SPEC.items():
        values = struct.unpack_from(fmt, expected_header(img.shape, np.int16), off)
        struct.pack_into(">" + fmt[1:], big, off, *values)
    q = tmp_path / "big.nii"
    q.write_bytes(bytes(big) + b"\0\0\0\0" + img.astype(">i2").tobytes(order="F"))
    assert np.array_equal(nifti_io.read_nifti(str(q), np.int16), img)
    struct.pack_into("<ff", hdr, 112, 2.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)
    r = tmp_path / "scaled.nii.gz"
    r.write_bytes(gzip.compress(bytes(hdr) + b"\0\0\0\0" + img.tobytes(order="F")))
    assert np.array_equal(nifti_io.read_nifti(str(r), np.float64), img * 2.0 + 1.0)


def test_quaternion_of_other_rotations_follows_the_spec():
    """nifti1.h METHOD 2 round trip: rebuild R from (b, c, d) and compare, incl. an improper one."""
    rng = np.random.default_rng(0)
    for k in range(20):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if k % 2:
            q[:, 2] = -q[:, 2] if np.linalg.det(q) > 0 else q[:, 2]      # improper: qfac = -1
        b, c, d = nifti_io._quaternion_bcd(np.vstack([np.hstack([q, np.zeros((3, 1))]), [0, 0, 0, 1]]))
        a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        r = np.array([[a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
                      [2 * b * c + 2 * a * d, a * a + c * c - b * b - d * d, 2 * c * d - 2 * a * b],
                      [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a + d * d - c * c - b * b]])
        proper = q.copy()
        if np.linalg.det(q) < 0:
            proper[:, 2] = -proper[:, 2]
        assert np.allclose(r, proper, atol=1e-6)
