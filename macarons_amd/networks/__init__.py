from .Attention import attention, Embedding, MultiHeadSelfAttention, FeedForward, Encoder   # noqa: F401
from .SconeVis import SconeVis                                                              # noqa: F401
from .SconeOcc import SconeOcc, PCTransformer, XEmbedding                                   # noqa: F401
from .Macarons import Macarons                                                              # noqa: F401
