from valle2_amd.utils import build_attn_mask, build_pad_mask, get_best_beam, topk_sampling  # noqa: F401
