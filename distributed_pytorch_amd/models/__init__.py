from .dummy import DummyDataset, DummyModel  # noqa: F401
