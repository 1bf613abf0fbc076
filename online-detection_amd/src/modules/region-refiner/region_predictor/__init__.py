from .predict_regions import RegionPredictor  # noqa: F401
