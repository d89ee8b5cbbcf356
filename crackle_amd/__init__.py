"""crackle_amd — MI355X-native encode/decode path of the crackle segmentation codec.

Drop-in for the reference's ``crackle.compress`` / ``crackle.decompress`` /
``crackle.decompress_range`` / ``crackle.header`` (same .ckl bitstream, bit-exact),
computed by hand-written HIP kernels behind the C-ABI in include/crackle_amd.h.
"""
from .headers import CrackleHeader, FormatError, LabelFormat, CrackFormat
from .codec import (
  compress, decompress, decompress_range, header, labels, num_labels, contains,
  voxel_counts, centroids, bounding_boxes, reencode, voxel_connectivity_graph,
  crack_crcs, structure_equal, labels_crc, check, ok,
)
from .operations import zstack, zsplit, zshatter, array_equal, mode_pooling_2x2x1, point_cloud

__all__ = [
  "CrackleHeader", "FormatError", "LabelFormat", "CrackFormat",
  "compress", "decompress", "decompress_range", "header", "labels", "num_labels", "contains",
  "voxel_counts", "centroids", "bounding_boxes", "reencode", "voxel_connectivity_graph", "crack_crcs", "structure_equal", "labels_crc", "check", "ok",
  "zstack", "zsplit", "zshatter", "array_equal", "mode_pooling_2x2x1", "point_cloud",
]
