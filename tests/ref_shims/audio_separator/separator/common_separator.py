"""`audio_separator.separator.common_separator` as the reference's unit tests import it, resolved to this repo's class.
A bare CommonSeparator has no engine (the architecture subclasses bind libasx.so); the reference's writer tests construct
exactly that, so here -- test infrastructure, CPU -- it gets the oracle-backed engine double of tests/fake_engine.py."""
from audio_separator_amd.common_separator import *  # noqa: F401,F403
from audio_separator_amd.common_separator import CommonSeparator as _Base
from audio_separator_amd.engine import MDXConfig
from tests.fake_engine import OracleEngine


class CommonSeparator(_Base):
    def __init__(self, config):
        super().__init__(config)
        if self.engine is None:
            self.engine = OracleEngine(MDXConfig())
