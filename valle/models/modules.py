from valle2_amd.modules import (AdaptiveLayerNorm, EncoderLayer, FeedForward,  # noqa: F401
                                MultiHeadAttention, PositionalEncoding, TokenEmbedding, Transformer)
