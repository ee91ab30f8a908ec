from .adamw import FusedAdamW  # noqa: F401
from .attention import causal_attention_qkvpacked  # noqa: F401
from .batchnorm import BatchNorm2d, batch_norm_act  # noqa: F401
from .conv import PointwiseConv2d  # noqa: F401
from .embedding import token_position_embedding  # noqa: F401
from .cross_entropy import SoftmaxCrossEntropy, softmax_cross_entropy  # noqa: F401
from .layernorm import LayerNorm  # noqa: F401
from .linear import Linear, linear_bias_act  # noqa: F401
from .rank1_mlp import rank1_mlp  # noqa: F401
