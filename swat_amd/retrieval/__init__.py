from .index import SegmentedIndex, group_by_label  # noqa: F401
from .ops import similarity_segmented, segment_topk  # noqa: F401
