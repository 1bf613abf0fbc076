"""Minimal stand-in for maskrcnn_benchmark.structures.bounding_box.BoxList: the on-line path
only touches .bbox / .size / .mode and add_field / get_field
(OnlineRegionClassifier.py:213-215, predict_regions.py:21-23,78).  Used when
maskrcnn_benchmark is not installed."""
import torch


class BoxList:
    def __init__(self, bbox, image_size, mode="xyxy"):
        self.bbox = torch.as_tensor(bbox)
        self.size = image_size
        self.mode = mode
        self.extra_fields = {}

    def add_field(self, field, data):
        self.extra_fields[field] = data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def __len__(self):
        return self.bbox.shape[0]


def get_boxlist_class():
    try:
        from maskrcnn_benchmark.structures.bounding_box import BoxList as _BL  # noqa: F401
        return _BL
    except Exception:
        return BoxList
