"""Stream header of the .ckl format: Python mirror of the reference's
crackle/headers.py (CrackleHeader, FormatError, LabelFormat, CrackFormat), backed by
the C-ABI's ckl_header_info_from_bytes (src/header.hpp:98-150 semantics)."""
from enum import IntEnum

import numpy as np

from . import _lib


class FormatError(Exception):
  pass


class LabelFormat(IntEnum):
  FLAT = 0
  PINS_FIXED_WIDTH = 1
  PINS_VARIABLE_WIDTH = 2


class CrackFormat(IntEnum):
  IMPERMISSIBLE = 0
  PERMISSIBLE = 1


width2dtype = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


class CrackleHeader:
  MAGIC = b"crkl"
  FORMAT_VERSION = 1
  HEADER_BYTES = 29

  def __init__(self, info: "_lib.HeaderInfo"):
    self.format_version = int(info.format_version)
    self.label_format = int(info.label_format)
    self.crack_format = int(info.crack_format)
    self.signed = bool(info.is_signed)
    self.data_width = int(info.data_width)
    self.stored_data_width = int(info.stored_data_width)
    self.sx, self.sy, self.sz = int(info.sx), int(info.sy), int(info.sz)
    self.fortran_order = bool(info.fortran_order)
    self.markov_model_order = int(info.markov_model_order)
    self.is_sorted = bool(info.is_sorted)
    self.num_label_bytes = int(info.num_label_bytes)
    self.header_bytes = int(info.header_bytes)
    self.grid_size = 2 ** 31

  @classmethod
  def frombytes(kls, buffer: bytes) -> "CrackleHeader":
    buffer = bytes(buffer[:64])
    info = _lib.HeaderInfo()
    rc = _lib.lib().ckl_header_info_from_bytes(buffer, len(buffer), info)
    if rc != _lib.CKL_OK:
      raise FormatError(_lib.last_error())
    return kls(info)

  @property
  def dtype(self):
    return width2dtype[self.data_width]

  @property
  def stored_dtype(self):
    return width2dtype[self.stored_data_width]

  @property
  def grid_index_bytes(self):
    return 4 * self.sz if self.format_version == 0 else 4 * (self.sz + 1)

  @property
  def markov_model_bytes(self):
    """Stored model size (header.hpp:284-297): 5 bits per context row."""
    if self.markov_model_order == 0:
      return 0
    return ((1 << (2 * self.markov_model_order)) * 5 + 4) // 8

  def voxels(self) -> int:
    return self.sx * self.sy * self.sz

  @property
  def nbytes(self) -> int:
    return self.voxels() * self.data_width

  def __repr__(self):
    return str(self.__dict__)
