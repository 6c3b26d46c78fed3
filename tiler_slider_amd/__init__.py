"""tiler_slider_amd — MI355X-native vectorised Tiler-Slider environment.

Export names follow the reference package (ref: explainrl/environment/__init__.py:13-25):
GameState, TilerSliderEnv, TilerSliderEnvFactory, TextRender — plus the batched
VecTilerSliderEnv that is the point of this build.  Importing the package loads nothing
native; constructing an environment loads lib/libtiler_slider_hip.so and fails loudly
if it is missing (no CPU fallback).
"""
from ._cabi import TilerSliderLibraryError, build_library
from .env import GameState, TilerSliderEnv
from .factory import TilerSliderEnvFactory, simple_level
from .gym_wrapper import GymVecTilerSlider
from .levels import ImageLoader, Level, pack_levels, parse_board_string
from .moves import Move
from .pipelined import PipelinedTilerSliderEnv
from .render import TextRender
from .vec_env import StepInfo, VecTilerSliderEnv

__version__ = "0.1.0"
__all__ = ["GameState", "Move", "TilerSliderEnv", "TilerSliderEnvFactory", "ImageLoader", "TextRender",
           "VecTilerSliderEnv", "PipelinedTilerSliderEnv",
           "StepInfo", "GymVecTilerSlider", "Level", "pack_levels", "parse_board_string", "simple_level", "build_library",
           "TilerSliderLibraryError"]
