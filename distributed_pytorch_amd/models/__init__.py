from .dummy import DummyDataset, DummyModel  # noqa: F401
from .gpt2 import GPT2, GPT2Config  # noqa: F401
from .resnet import ResNet, resnet50  # noqa: F401
from .wide_mlp import WideMLP  # noqa: F401
