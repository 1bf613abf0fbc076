from .train_region_refiner import RegionRefinerTrainer  # noqa: F401
