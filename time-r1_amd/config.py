"""Architecture constants for the Qwen2-VL family (the model the reference trains).

Values for the released checkpoints are the public HF `config.json` numbers (not verifiable offline).
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class VisionConfig:
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_dim: int = 5120            # Qwen2-VL: mlp_ratio 4 (quick_gelu, biased fc1/fc2)
    out_hidden: int = 3584         # = text hidden
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    in_channels: int = 3
    variant: str = "qwen2_vl"      # "qwen2_vl" | "qwen2_5_vl" (RMSNorm + SwiGLU + window attention)
    window_size: int = 112
    fullatt_block_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    ln_eps: float = 1e-6

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def patch_dim(self):
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size

    @property
    def patch_dim_padded(self):   # GEMM K must be a multiple of 64
        return (self.patch_dim + 63) // 64 * 64

    @property
    def merge_unit(self):
        return self.spatial_merge_size ** 2


@dataclass
class TextConfig:
    vocab_size: int = 152064
    hidden: int = 3584
    intermediate: int = 18944
    n_layers: int = 28
    n_heads: int = 28
    n_kv_heads: int = 4
    head_dim: int = 128
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    tie_word_embeddings: bool = False

    @property
    def q_dim(self):
        return self.n_heads * self.head_dim

    @property
    def kv_dim(self):
        return self.n_kv_heads * self.head_dim

    @property
    def qkv_dim(self):
        return self.q_dim + 2 * self.kv_dim


@dataclass
class ModelConfig:
    text: TextConfig = field(default_factory=TextConfig)
    vision: VisionConfig = field(default_factory=VisionConfig)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_id: int = 151645
    pad_token_id: int = 151643
    tokens_per_second: float = 2.0     # Qwen2.5-VL only
    name: str = "qwen2-vl-7b"


def qwen2_vl_7b():
    return ModelConfig(name="Qwen2-VL-7B")


def qwen2_vl_2b():
    return ModelConfig(
        text=TextConfig(vocab_size=151936, hidden=1536, intermediate=8960, n_layers=28, n_heads=12, n_kv_heads=2, head_dim=128,
                        tie_word_embeddings=True),
        vision=VisionConfig(out_hidden=1536), name="Qwen2-VL-2B")


def tiny_test(vocab=512, n_layers=2, vision_depth=2, tie=False):
    """Small shapes that still satisfy the kernels' alignment rules (K % 64, head_dim % 32); used by tests and smoke()."""
    return ModelConfig(
        text=TextConfig(vocab_size=vocab, hidden=128, intermediate=256, n_layers=n_layers, n_heads=4, n_kv_heads=2, head_dim=32,
                        mrope_section=(4, 6, 6), tie_word_embeddings=tie),
        vision=VisionConfig(depth=vision_depth, embed_dim=64, num_heads=2, mlp_dim=128, out_hidden=128),
        image_token_id=500, video_token_id=501, vision_start_token_id=502, vision_end_token_id=503, eos_token_id=1, pad_token_id=0,
        name="tiny")


PRESETS = {"qwen2-vl-7b": qwen2_vl_7b, "qwen2-vl-2b": qwen2_vl_2b, "tiny": tiny_test}
