from . import launcher, process_group, sampler, store  # noqa: F401
from .ddp import DistributedDataParallel  # noqa: F401
from .launcher import ProcessExitedException, ProcessRaisedException, launch, spawn  # noqa: F401
from .sampler import ShardedSampler  # noqa: F401
from .store import Store  # noqa: F401
from .zero import ShardedOptimizer  # noqa: F401
