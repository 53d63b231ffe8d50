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

    @property
    def mlp_dim_padded(self):     # Qwen2.5-VL: intermediate 3420 -> 3456 (zero rows/cols, exact) so GEMM K/N stay multiples of 64
        return (self.mlp_dim + 63) // 64 * 64


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


def qwen2_5_vl_7b():
    """Qwen2.5-VL-7B (the model family the reference's trainer hard-codes, timer1_trainer.py:250-262; BASELINE configs[3])."""
    return ModelConfig(vision=VisionConfig(mlp_dim=3420, variant="qwen2_5_vl"), name="Qwen2.5-VL-7B")


def qwen2_5_vl_3b():
    return ModelConfig(
        text=TextConfig(vocab_size=151936, hidden=2048, intermediate=11008, n_layers=36, n_heads=16, n_kv_heads=2, head_dim=128,
                        tie_word_embeddings=True),
        vision=VisionConfig(mlp_dim=3420, out_hidden=2048, variant="qwen2_5_vl"), name="Qwen2.5-VL-3B")


def tiny_test(vocab=512, n_layers=2, vision_depth=2, tie=False):
    """Small shapes that still satisfy the kernels' alignment rules (K % 64, head_dim % 32); used by tests and smoke()."""
    return ModelConfig(
        text=TextConfig(vocab_size=vocab, hidden=128, intermediate=256, n_layers=n_layers, n_heads=4, n_kv_heads=2, head_dim=32,
                        mrope_section=(4, 6, 6), tie_word_embeddings=tie),
        vision=VisionConfig(depth=vision_depth, embed_dim=64, num_heads=2, mlp_dim=128, out_hidden=128),
        image_token_id=500, video_token_id=501, vision_start_token_id=502, vision_end_token_id=503, eos_token_id=1, pad_token_id=0,
        name="tiny")


def tiny_test_25(vocab=512, n_layers=2, vision_depth=3, tie=False):
    """tiny_test with the Qwen2.5-VL vision tower: RMSNorm, biased SwiGLU MLP of a non-aligned width, windows of 2x2 merged tokens,
    full attention only in the last block."""
    cfg = tiny_test(vocab, n_layers, vision_depth, tie)
    cfg.vision = VisionConfig(depth=vision_depth, embed_dim=64, num_heads=2, mlp_dim=100, out_hidden=128, variant="qwen2_5_vl",
                              window_size=56, fullatt_block_indexes=(vision_depth - 1,))
    cfg.name = "tiny25"
    return cfg


def to_hf_config(cfg: ModelConfig):
    """config.json in the transformers layout (configuration_qwen2_vl.py / configuration_qwen2_5_vl.py), so a directory written by
    save_model loads with `from_pretrained` (the reference's eval and vLLM paths) and with trainer.load_model_dir."""
    t, v = cfg.text, cfg.vision
    q25 = v.variant == "qwen2_5_vl"
    text = dict(vocab_size=t.vocab_size, hidden_size=t.hidden, intermediate_size=t.intermediate, num_hidden_layers=t.n_layers,
                num_attention_heads=t.n_heads, num_key_value_heads=t.n_kv_heads, rms_norm_eps=t.rms_eps, hidden_act="silu",
                rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta, "mrope_section": list(t.mrope_section)},
                rope_theta=t.rope_theta, rope_scaling={"type": "mrope", "mrope_section": list(t.mrope_section)},
                tie_word_embeddings=t.tie_word_embeddings, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id,
                max_position_embeddings=128000, model_type="qwen2_5_vl_text" if q25 else "qwen2_vl_text")
    if q25:
        vision = dict(depth=v.depth, hidden_size=v.embed_dim, hidden_act="silu", intermediate_size=v.mlp_dim, num_heads=v.num_heads,
                      in_channels=v.in_channels, patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
                      temporal_patch_size=v.temporal_patch_size, tokens_per_second=int(cfg.tokens_per_second), window_size=v.window_size,
                      out_hidden_size=v.out_hidden, fullatt_block_indexes=list(v.fullatt_block_indexes), model_type="qwen2_5_vl")
    else:
        vision = dict(depth=v.depth, embed_dim=v.embed_dim, num_heads=v.num_heads, hidden_size=v.out_hidden, mlp_ratio=v.mlp_dim // v.embed_dim,
                      hidden_act="quick_gelu", in_channels=v.in_channels, patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
                      temporal_patch_size=v.temporal_patch_size, model_type="qwen2_vl")
    return dict(architectures=["Qwen2_5_VLForConditionalGeneration" if q25 else "Qwen2VLForConditionalGeneration"],
                model_type="qwen2_5_vl" if q25 else "qwen2_vl", text_config=text, vision_config=vision, image_token_id=cfg.image_token_id,
                video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
                tie_word_embeddings=t.tie_word_embeddings, torch_dtype="bfloat16")


PRESETS = {"qwen2-vl-7b": qwen2_vl_7b, "qwen2-vl-2b": qwen2_vl_2b, "qwen2.5-vl-7b": qwen2_5_vl_7b, "qwen2.5-vl-3b": qwen2_5_vl_3b,
           "tiny": tiny_test, "tiny25": tiny_test_25}
