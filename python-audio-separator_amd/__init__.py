"""MI355X-native demix path for python-audio-separator's MDX plugin.

Imported through the ``audio_separator_amd`` shim at the repo root (this
directory's name is not a valid Python identifier)."""

