"""Minimal NIfTI-1 single-file writer / header reader (nibabel is not in the image).

The reference writes every reconstruction through nibabel with the affine and header of a reference
volume (`vae_reg_GP.py:618-620`, `build_model_recons.py:104-116`).  Here the 348-byte header of the
reference file is copied (so qform/sform, pixdim and units survive), with dim / datatype / bitpix /
vox_offset / scaling rewritten for the float32 map that follows.  Without a reference file a plain
header with an identity sform is written.
"""
import gzip
import struct

import numpy as np


def read_header(path):
    """The raw 348-byte NIfTI-1 header of `path` and its endianness ('<' or '>')."""
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as f:
        raw = f.read(348)
    if len(raw) < 348:
        raise ValueError('%s: not a NIfTI-1 file' % path)
    if struct.unpack('<i', raw[:4])[0] == 348:
        return raw, '<'
    if struct.unpack('>i', raw[:4])[0] == 348:
        return raw, '>'
    raise ValueError('%s: not a NIfTI-1 file' % path)


def _plain_header():
    h = bytearray(348)
    struct.pack_into('<i', h, 0, 348)
    struct.pack_into('<8f', h, 76, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)       # pixdim
    struct.pack_into('<h', h, 254, 1)                                             # sform_code: scanner
    struct.pack_into('<4f', h, 280, 1.0, 0.0, 0.0, 0.0)                           # srow_x/y/z = identity
    struct.pack_into('<4f', h, 296, 0.0, 1.0, 0.0, 0.0)
    struct.pack_into('<4f', h, 312, 0.0, 0.0, 1.0, 0.0)
    h[344:348] = b'n+1\x00'
    return h


def write_nifti1(path, array, reference=None):
    """Write `array` (3-D or 4-D, any real dtype) as float32 NIfTI-1 (.nii or .nii.gz), Fortran order on disk
    as the format requires.  `reference`: path of a NIfTI-1 file whose geometry (affine, pixdim, units) is kept."""
    a = np.asarray(array, dtype=np.float32)
    if a.ndim < 1 or a.ndim > 7:
        raise ValueError('NIfTI-1 holds 1..7 dimensions')
    if reference is not None:
        raw, en = read_header(reference)
        h = bytearray(raw)
    else:
        h, en = _plain_header(), '<'
    dim = [a.ndim] + list(a.shape) + [1] * (7 - a.ndim)
    struct.pack_into(en + '8h', h, 40, *dim)
    struct.pack_into(en + '2h', h, 70, 16, 32)                                    # datatype float32, bitpix
    struct.pack_into(en + 'f', h, 108, 352.0)                                     # vox_offset
    struct.pack_into(en + '2f', h, 112, 1.0, 0.0)                                 # scl_slope, scl_inter
    struct.pack_into(en + '2f', h, 124, 0.0, 0.0)                                 # cal_max, cal_min
    h[344:348] = b'n+1\x00'
    data = np.asfortranarray(a).astype(np.dtype(en + 'f4'), copy=False).tobytes(order='F')
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'wb') as f:
        f.write(bytes(h)); f.write(b'\x00\x00\x00\x00'); f.write(data)
