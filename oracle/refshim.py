"""Import shim for the *real* reference implementation (container only).

TEST INFRASTRUCTURE - not product code.  Nothing under ``followyourclick_amd/``
imports this.  It makes ``/root/reference`` importable in the build container
so that (a) ``oracle/functional.py`` (the CPU restatement) can be validated
against the reference's own Python and (b) ``oracle/make_golden.py`` can emit
the golden vectors committed under ``tests/golden/``.  ``/root/reference`` does
not exist on the GPU box, so nothing executed there may call ``install()``.

The reference vendors diffusers 0.11.1 whose ``__init__`` imports APIs removed
from today's huggingface_hub / transformers; we register a stub *package*
object for ``diffusers`` so its sub-modules import without running that
``__init__`` (reference: diffusers/__init__.py:46-47,
diffusers/dynamic_modules_utils.py:29).
"""
import importlib.machinery as _M
import os
import sys
import types

_STAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")      # byte copies written by oracle/stage_ref_scripts.py (git-ignored; travels to the GPU box)
REFERENCE_ROOT = os.environ.get("FYC_REFERENCE_ROOT") or ("/root/reference" if os.path.isdir("/root/reference/animatediff") else _STAGED)


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "animatediff"))


def install() -> None:
    """Make ``animatediff`` / ``diffusers`` / ``ip_adapter`` resolve to the reference tree."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if getattr(sys.modules.get("diffusers"), "_fyc_refshim", False):
        return
    sys.dont_write_bytecode = True
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed)
    import huggingface_hub as hh

    def _offline(*a, **k):
        raise RuntimeError("offline")

    for n in ("HfFolder", "cached_download", "model_info", "hf_hub_download"):
        if not hasattr(hh, n):
            setattr(hh, n, _offline)
    for m in ("torchvision", "loguru", "imageio", "xformers"):
        if m in sys.modules:
            continue
        try:
            __import__(m)
        except Exception:
            if m == "xformers":
                continue
            mod = types.ModuleType(m)
            mod.__spec__ = _M.ModuleSpec(m, None)
            if m == "loguru":
                mod.logger = types.SimpleNamespace(
                    info=lambda *a, **k: None, warning=lambda *a, **k: None, debug=lambda *a, **k: None)
            sys.modules[m] = mod
    for name in [k for k in sys.modules if k == "diffusers" or k.startswith("diffusers.")
                 or k == "animatediff" or k.startswith("animatediff.")
                 or k == "ip_adapter" or k.startswith("ip_adapter.")]:
        del sys.modules[name]
    pkg = types.ModuleType("diffusers")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "diffusers")]
    pkg.__version__ = "0.11.1"
    pkg.__spec__ = _M.ModuleSpec("diffusers", None, is_package=True)
    pkg.__spec__.submodule_search_locations = pkg.__path__
    pkg._fyc_refshim = True
    sys.modules["diffusers"] = pkg
    pm = types.ModuleType("diffusers.pipelines")
    pm.__spec__ = _M.ModuleSpec("diffusers.pipelines", None)
    sys.modules["diffusers.pipelines"] = pm
    pkg.pipelines = pm
    pkg.StableDiffusionPipeline = object  # ip_adapter/ip_adapter.py:5 imports the name
    ip = types.ModuleType("ip_adapter")
    ip.__path__ = [os.path.join(REFERENCE_ROOT, "ip_adapter")]
    ip.__spec__ = _M.ModuleSpec("ip_adapter", None, is_package=True)
    ip.__spec__.submodule_search_locations = ip.__path__
    sys.modules["ip_adapter"] = ip
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def uninstall() -> None:
    for name in [k for k in sys.modules if k.split(".")[0] in ("diffusers", "animatediff", "ip_adapter")]:
        del sys.modules[name]
    if REFERENCE_ROOT in sys.path:
        sys.path.remove(REFERENCE_ROOT)
