"""Test infrastructure only: makes /root/reference's *model* modules importable in the build container.

Used by oracle/make_golden.py and the optional cross-check tests to run the real reference
(VikParuchuri/surya @ v0.14.6) on CPU with seeded synthetic weights.  Nothing here is product code and
nothing here exists on the GPU box (tests that need it skip when /root/reference is absent).

Recipe follows SURVEY.md Appendix A:
  * stub modules for imports missing from this image (dotenv, pydantic_settings, cv2, filetype, pypdfium2);
  * transformers 5.x removed ROPE_INIT_FUNCTIONS["default"] (used at surya/common/surya/decoder/__init__.py:333);
  * SuryaDecoderConfig needs pad_token_id passed explicitly (read at decoder/__init__.py:401);
  * lm_head/token_embed tying is done by hand by the caller when wanted (common/surya/__init__.py:111-116).
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("SURYA_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "surya"))


def install():
    """Put stubs + the reference on sys.path and patch transformers for the reference's needs."""
    if not available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    for p in (REFERENCE_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    from transformers import modeling_rope_utils as mru

    def _default_rope(config, device=None, **kw):
        d = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        inv = 1.0 / (config.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
        return inv, 1.0

    mru.ROPE_INIT_FUNCTIONS.setdefault("default", _default_rope)
    return True


def purge_bare_namespaces():
    """Forget `surya.*` packages that tests registered as bare namespaces (to import one submodule without running the package's
    __init__), so that the real package can be imported afterwards."""
    for name, m in list(sys.modules.items()):
        if name.startswith("surya.") and hasattr(m, "__path__") and getattr(m, "__spec__", None) is None:
            del sys.modules[name]


def import_recognition():
    """The reference's `surya.recognition` package itself (its __init__ imports `QuantizedCacheConfig` / `HQQQuantizedCache`, which
    transformers 5.x no longer has, and transformers' lazy module forgets attributes set on it by hand): placeholder classes are
    injected at the moment `from transformers import ...` asks for them. Only host logic of that package is usable this way
    (`RecognitionPredictor.get_bboxes_text` and friends as plain functions); the quantised-cache code paths are not."""
    import builtins
    install()

    class _Missing:
        def __init__(self, *a, **k):
            pass

    names = {"QuantizedCacheConfig", "HQQQuantizedCache"}
    orig = builtins.__import__

    def hook(name, globals=None, locals=None, fromlist=(), level=0):
        m = orig(name, globals, locals, fromlist, level)
        if name == "transformers" and fromlist:
            for n in fromlist:
                if n in names and n not in m.__dict__:
                    object.__setattr__(m, n, _Missing)
        return m

    purge_bare_namespaces()
    builtins.__import__ = hook
    try:
        import surya.recognition as sr
    finally:
        builtins.__import__ = orig
    return sr


def import_submodule(modname: str):
    """Import ONE reference submodule without running its package's __init__ (several of those import names that transformers 5.x
    dropped, or cv2 / pypdfium2 users): the parent packages are registered as bare namespaces pointing at the reference tree."""
    import importlib
    import types
    install()
    importlib.import_module("surya")
    parts = modname.split(".")
    for i in range(2, len(parts)):
        pkg = ".".join(parts[:i])
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *pkg.split("."))]
            sys.modules[pkg] = m
    return importlib.import_module(modname)


def install_layout():
    """What the layout / table-rec model family needs on top of install(): names transformers 5.x removed from pytorch_utils (only
    used by head pruning, never at inference)."""
    install()
    import transformers.pytorch_utils as pu
    if not hasattr(pu, "find_pruneable_heads_and_indices"):
        def _no_pruning(*a, **k):
            raise NotImplementedError("head pruning is not available under the shim")
        pu.find_pruneable_heads_and_indices = _no_pruning
    from transformers.modeling_utils import PreTrainedModel
    if not hasattr(PreTrainedModel, "get_head_mask"):                 # removed in 5.x; the encoder calls it with head_mask=None
        PreTrainedModel.get_head_mask = lambda self, head_mask, num_hidden_layers, *a, **k: [None] * num_hidden_layers
    return True


def import_layout_modules():
    """(config, encoder, decoder) modules of the reference's layout model with the transformers-5.x incompatibilities patched:
    SuryaADETRDecoderPreTrainedModel.tie_weights takes no arguments there, transformers now passes one."""
    install_layout()
    cfgm = import_submodule("surya.layout.model.config")
    encm = import_submodule("surya.layout.model.encoder")
    adetr = import_submodule("surya.common.adetr.decoder")
    adetr.SuryaADETRDecoderPreTrainedModel.tie_weights = lambda self, *a, **k: None
    decm = import_submodule("surya.layout.model.decoder")
    return cfgm, encm, decm


def import_table_modules():
    """(config, encoder, decoder, shaper, processor-less helpers) modules of the reference's table-recognition model, patched like
    import_layout_modules (tie_weights signature)."""
    install_layout()
    cfgm = import_submodule("surya.table_rec.model.config")
    encm = import_submodule("surya.table_rec.model.encoder")
    adetr = import_submodule("surya.common.adetr.decoder")
    adetr.SuryaADETRDecoderPreTrainedModel.tie_weights = lambda self, *a, **k: None
    decm = import_submodule("surya.table_rec.model.decoder")
    shaper = import_submodule("surya.table_rec.shaper")
    return cfgm, encm, decm, shaper
