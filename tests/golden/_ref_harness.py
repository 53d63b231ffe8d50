"""Harness that imports the UNMODIFIED reference (/root/reference, read-only) in the build container so that golden vectors can be
captured from it. Runs only here (the reference cannot travel to the GPU box); its outputs are the small fixtures in this directory.

Stubs stand in for third-party wheels that are absent offline (trl, deepspeed, torchvision, rouge_score); they carry no algorithm -
the reference's own reward / sizing / compute_loss code runs as written (recipe: SURVEY.md appendix F).
Run with PYTHONDONTWRITEBYTECODE=1 so that importing does not write __pycache__ into /root/reference.
"""
import contextlib
import copy
import dataclasses
import importlib.machinery
import importlib.util
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch  # noqa: F401
    import transformers  # noqa: F401
    import datasets  # noqa: F401
    from transformers import TrainingArguments

    @dataclasses.dataclass
    class GRPOConfig(TrainingArguments):
        model_init_kwargs: dict = None
        max_prompt_length: int = 512
        max_completion_length: int = 256
        num_generations: int = 8
        temperature: float = 0.9
        beta: float = 0.04

    @contextlib.contextmanager
    def unwrap_model_for_generation(model, accelerator, **kw):
        yield model

    def create_reference_model(m):
        r = copy.deepcopy(m)
        for p in r.parameters():
            p.requires_grad_(False)
        return r.eval()

    @dataclasses.dataclass
    class ScriptArguments:
        pass

    _stub("trl", GRPOConfig=GRPOConfig, ModelConfig=object, ScriptArguments=ScriptArguments, TrlParser=object, get_peft_config=lambda *a: None)
    _stub("trl.data_utils", apply_chat_template=None, is_conversational=lambda ex: False)
    _stub("trl.models", create_reference_model=create_reference_model, prepare_deepspeed=None, unwrap_model_for_generation=unwrap_model_for_generation)
    _stub("trl.trainer")
    _stub("trl.trainer.grpo_config", GRPOConfig=GRPOConfig)
    _stub("trl.trainer.utils", generate_model_card=None, get_comet_experiment_url=None)
    tv = _stub("torchvision", __version__="0.21.0")
    tv.io = _stub("torchvision.io")
    tv.transforms = _stub("torchvision.transforms", InterpolationMode=types.SimpleNamespace(BICUBIC="bicubic"), functional=None)
    for n in ("deepspeed", "deepspeed.runtime", "deepspeed.runtime.fp16", "deepspeed.runtime.zero"):
        _stub(n)
    _stub("deepspeed.runtime.fp16.loss_scaler", LossScaler=type("LossScaler", (), {}))
    _stub("deepspeed.runtime.zero.config", ZeroStageEnum=type("ZeroStageEnum", (), {}))
    _stub("rouge_score", rouge_scorer=types.SimpleNamespace())
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_ref_main():
    install_stubs()
    spec = importlib.util.spec_from_file_location("ref_main", REF + "/main.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_finetune():
    """finetune.py (config 4's entry point): its own copies of the reward / metric registries (finetune.py:716-727)."""
    install_stubs()
    spec = importlib.util.spec_from_file_location("ref_finetune", REF + "/finetune.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_vision_process():
    install_stubs()
    spec = importlib.util.spec_from_file_location("ref_vision_process", REF + "/src/utils/vision_process.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_trainers():
    install_stubs()
    import src.time_r1.rl.timer1_trainer as t1
    import src.time_r1.rl.timer1_trainer_ft as t2
    return t1, t2
