#!/usr/bin/env python3
"""Byte-level NIfTI-1 fixtures written from the format specification (nifti1.h field table), NOT with vae_gam_amd.nifti.

TEST INFRASTRUCTURE.  One `struct` format string spells out all 43 fields of the 348-byte header in their specified order and
types; the file is header + 4 zero bytes (no extensions) + the voxels with the FIRST index varying fastest.  The volume is
3 x 4 x 5 float32 with a[i, j, k] = 100 i + 10 j + k, identity sform (code 1), unit pixdim.  Written little- and big-endian:
  tests/golden/nifti1_3x4x5_f32_le.nii   tests/golden/nifti1_3x4x5_f32_be.nii
tests/test_host_logic.py checks the package's reader against both and its writer byte-for-byte against the little-endian one.
"""
import os
import struct

FIELDS = ('i'      # sizeof_hdr = 348
          '10s'    # data_type (unused)
          '18s'    # db_name (unused)
          'i'      # extents (unused)
          'h'      # session_error (unused)
          'c'      # regular (unused)
          'B'      # dim_info
          '8h'     # dim[8]: rank, nx, ny, nz, nt, nu, nv, nw
          '3f'     # intent_p1..p3
          'h'      # intent_code
          'h'      # datatype: 16 = float32
          'h'      # bitpix: 32
          'h'      # slice_start
          '8f'     # pixdim[8]
          'f'      # vox_offset: 352 for a single .nii file without extensions
          'f'      # scl_slope
          'f'      # scl_inter
          'h'      # slice_end
          'B'      # slice_code
          'B'      # xyzt_units
          'f'      # cal_max
          'f'      # cal_min
          'f'      # slice_duration
          'f'      # toffset
          'i'      # glmax (unused)
          'i'      # glmin (unused)
          '80s'    # descrip
          '24s'    # aux_file
          'h'      # qform_code
          'h'      # sform_code: 1 = scanner-anatomical
          '3f'     # quatern_b, c, d
          '3f'     # qoffset_x, y, z
          '4f'     # srow_x
          '4f'     # srow_y
          '4f'     # srow_z
          '16s'    # intent_name
          '4s')    # magic: "n+1\0" = single file


def build(endian):
    shape = (3, 4, 5)
    values = [348, b'', b'', 0, 0, b'\x00', 0,
              3, shape[0], shape[1], shape[2], 1, 1, 1, 1,
              0.0, 0.0, 0.0, 0, 16, 32, 0,
              1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0,
              352.0, 1.0, 0.0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0, 0, b'', b'', 0, 1,
              0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
              1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0,
              b'', b'n+1\x00']
    hdr = struct.pack(endian + FIELDS, *values)
    assert len(hdr) == 348
    vox = [100.0 * i + 10.0 * j + k for k in range(shape[2]) for j in range(shape[1]) for i in range(shape[0])]   # i fastest on disk
    return hdr + b'\x00\x00\x00\x00' + struct.pack(endian + '%df' % len(vox), *vox)


if __name__ == '__main__':
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
    for en, tag in (('<', 'le'), ('>', 'be')):
        with open(os.path.join(out, 'nifti1_3x4x5_f32_%s.nii' % tag), 'wb') as f:
            f.write(build(en))
    print('wrote 2 x', len(build('<')), 'bytes')
