"""Import shim: the product package lives in ``python-audio-separator_amd/`` (a
directory name Python cannot import directly); this module makes it importable as
``audio_separator_amd`` by extending ``__path__``."""
import os as _os

_pkg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "python-audio-separator_amd")
__path__.append(_pkg)

from .engine import Engine, AsxError, HDConfig, HTConfig, MDXConfig, NetConfig, RofConfig, V3Config, lib_path  # noqa: E402,F401
from .weights import fold_convtdf_state  # noqa: E402,F401
from .mdx import STFT, MDXDemixer  # noqa: E402,F401
from .onnx_reader import convtdf_from_onnx, OnnxFormatError  # noqa: E402,F401
from .mdxc import MDXCDemixer  # noqa: E402,F401
from .demucs import DemucsDemixer, hdconfig_from_kwargs, htconfig_from_kwargs  # noqa: E402,F401
from .vr import VRDemixer, load_model_params, model_capacity  # noqa: E402,F401
from .plugin import install, uninstall  # noqa: E402,F401
from .common_separator import CommonSeparator  # noqa: E402,F401
