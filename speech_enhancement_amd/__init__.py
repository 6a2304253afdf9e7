"""Import alias: the product lives in ``speech-enhancement_amd/`` (a directory name Python cannot
import); this stub makes it importable as ``speech_enhancement_amd``."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..', 'speech-enhancement_amd'))
from ._pkg import *  # noqa: F401,F403,E402
